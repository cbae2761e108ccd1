// mtg_prepare.h -- theta -> prior verdict + celerite coefficients for ONE evaluation, as a
// device function: the body of mtg_prepare_kernel (mtg_kernels.hip), also inlined into the
// sampler's proposal kernel (mtg_sampler.hip) so that a half-step needs no separate launch.
//
// Every lane of a wave must call it (``live`` = this lane holds evaluation e): the evaluation
// indices are appended to their structure's list with wave-aggregated atomics.
#pragma once
#include "mtg_device.h"

#include <math.h>

// th_row: where this evaluation's theta row is to be READ from when it is not a.theta + e * P -- the sampler's
// speculative kernel keeps the proposals it has just made in LDS as well (a global round trip per parameter, one after
// the other through the term loop: 1.0-1.2 us of the 6-9 us the expansion takes in that kernel; the rest is the
// arithmetic of ~8 exp, a sqrt and three divisions per row on the six waves that hold the rows)
// (Not kept: the model's description read from a copy in LDS instead of the kernel arguments -- the expansion took as
// long, and copying the description out of the argument segment with per-thread indices cost 27 us.)
// th: the row's free parameters (valid for a dead thread too)
__device__ __forceinline__ void mtg_prepare_from(const MtgPrepArgs &a, int64_t e, bool live, const double *th)
{
    const MtgModel &m = a.model;
    auto par = [th, &m](int k) -> double {   // (th by value: a reference would take the address of a local, which the back end mishandles when it survives inlining)
        const int s = m.src[k];
        return s >= 0 ? th[s] : m.defaults[k];
    };

    const bool mine = e >= a.row_lo && e < a.row_hi;
    bool ok = live && mine;
    if (ok && a.add_prior) {
        // celerite Model.log_prior: every parameter, frozen ones included
        for (int k = 0; k < m.PF; ++k) {
            const double v = par(k);
            ok = ok && (v >= m.lo[k]) && (v <= m.hi[k]);
        }
        // BendingPowerlaw.log_prior, celerite_models.py:85-90
        // celerite ComplexTerm.log_prior (4-parameter form): log_a + log_c >= log_b + log_d
        for (int i = 0; i < m.nterms; ++i) {
            const int o = m.poff[i];
            if (m.kinds[i] == MTG_TERM_BPL) ok = ok && !(par(o) < par(o + 1));
            if (m.kinds[i] == MTG_TERM_COMPLEX4)
                ok = ok && !(par(o) + par(o + 2) < par(o + 1) + par(o + 3));
        }
    }
    if (live) {
        a.status[e] = ok ? MTG_ST_OK : mine ? MTG_ST_PRIOR : MTG_ST_REMOTE;
        if (!ok && mine) a.out[e] = -INFINITY;
    }

    int nover = 0;
    if (ok) {
        MtgCoefLayout lay{m.nr_max, m.nc_max};
        double *c = a.coef + e;
        const int64_t cs = a.cstride;
        int ir = 0, ic = 0;
        double asum = 0.0, jit = 0.0;
        for (int i = 0; i < m.nterms; ++i) {
            const int o = m.poff[i];
            switch (m.kinds[i]) {
            case MTG_TERM_REAL: {
                const double av = exp(par(o));
                c[lay.ar(ir) * cs] = av; c[lay.cr(ir) * cs] = exp(par(o + 1)); ++ir; asum += av;
                break;
            }
            case MTG_TERM_COMPLEX3: {
                const double av = exp(par(o));
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = 0.0;
                c[lay.cc(ic) * cs] = exp(par(o + 1)); c[lay.dc(ic) * cs] = exp(par(o + 2)); ++ic;
                asum += av;
                break;
            }
            case MTG_TERM_COMPLEX4: {
                const double av = exp(par(o));
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = exp(par(o + 1));
                c[lay.cc(ic) * cs] = exp(par(o + 2)); c[lay.dc(ic) * cs] = exp(par(o + 3)); ++ic;
                asum += av;
                break;
            }
            case MTG_TERM_SHO: {
                const double S0 = exp(par(o)), Q = exp(par(o + 1)), w0 = exp(par(o + 2));
                if (Q < 0.5) {  // over-damped: two real terms
                    const double f = sqrt(1.0 - 4.0 * Q * Q);
                    const double a1 = 0.5 * S0 * w0 * Q * (1.0 + 1.0 / f);
                    const double a2 = 0.5 * S0 * w0 * Q * (1.0 - 1.0 / f);
                    c[lay.ar(ir) * cs] = a1; c[lay.cr(ir) * cs] = 0.5 * w0 / Q * (1.0 - f); ++ir;
                    c[lay.ar(ir) * cs] = a2; c[lay.cr(ir) * cs] = 0.5 * w0 / Q * (1.0 + f); ++ir;
                    asum += a1; asum += a2;
                    ++nover;
                } else {
                    const double f = sqrt(4.0 * Q * Q - 1.0);
                    const double av = S0 * w0 * Q;
                    c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = av / f;
                    c[lay.cc(ic) * cs] = 0.5 * w0 / Q; c[lay.dc(ic) * cs] = 0.5 * w0 / Q * f; ++ic;
                    asum += av;
                }
                break;
            }
            case MTG_TERM_MATERN32: {
                const double eps = m.extra[i];
                const double w0 = sqrt(3.0) * exp(-par(o + 1));
                const double S0 = exp(2.0 * par(o)) / w0;
                const double av = w0 * S0;
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = w0 * w0 * S0 / eps;
                c[lay.cc(ic) * cs] = w0; c[lay.dc(ic) * cs] = eps; ++ic;
                asum += av;
                break;
            }
            case MTG_TERM_JITTER: {
                const double jv = exp(2.0 * par(o));
                asum += jv; jit += jv;
                break;
            }
            case MTG_TERM_DRW: {  // celerite_models.py:58-66, Q = 1/2
                const double av = exp(par(o));
                c[lay.ar(ir) * cs] = av; c[lay.cr(ir) * cs] = 0.5 * exp(par(o + 1)) / 0.5; ++ir;
                asum += av;
                break;
            }
            case MTG_TERM_LORENTZIAN: {
                // celerite_models.py:9-31.  The (a=0, c=0) real term the reference
                // returns has U = 0, so it never enters D_n or z_n: it is not
                // expanded (identical lnL, one rank less work).
                const double av = exp(par(o));
                const double w0 = exp(par(o + 2));
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = 0.0;
                c[lay.cc(ic) * cs] = 0.5 * w0 / exp(par(o + 1)); c[lay.dc(ic) * cs] = w0; ++ic;
                asum += av;
                break;
            }
            case MTG_TERM_COSINUS: {  // celerite_models.py:39-52
                const double av = exp(par(o));
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = 0.0;
                c[lay.cc(ic) * cs] = 0.0; c[lay.dc(ic) * cs] = exp(par(o + 1)); ++ic;
                asum += av;
                break;
            }
            case MTG_TERM_BPL: {  // celerite_models.py:77-83
                const double av = exp(par(o));
                const double w0 = exp(par(o + 2));
                c[lay.ac(ic) * cs] = av; c[lay.bc(ic) * cs] = exp(par(o + 1));
                c[lay.cc(ic) * cs] = w0; c[lay.dc(ic) * cs] = w0; ++ic;
                asum += av;
                break;
            }
            default:
                break;
            }
        }
        c[lay.asum() * cs] = asum;
        c[lay.jit() * cs] = jit;
        // mean(t) = slope * t + intercept; a constant mean is slope 0 (exactly the value)
        c[lay.mean(0) * cs] = m.mean_kind == MTG_MEAN_LINEAR ? par(m.nk) : 0.0;
        c[lay.mean(1) * cs] = m.mean_kind == MTG_MEAN_LINEAR ? par(m.nk + 1) : par(m.nk);
    }

    if (a.sig && live) a.sig[e] = nover;
    if (a.nsig > 1) {
        // wave-aggregated append of the evaluation index to its signature list
        const int lane = threadIdx.x & 63;
        for (int k = 0; k < a.nsig; ++k) {
            const bool mine = ok && nover == k;
            const unsigned long long mask = __ballot(mine);
            if (mask == 0ull) continue;
            const int leader = __ffsll((long long)mask) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(a.counts + k, __popcll(mask));
            base = __shfl(base, leader);
            if (mine)
                a.lists[(int64_t)k * a.cstride + base + __popcll(mask & ((1ull << lane) - 1ull))] =
                    (int)e;
        }
    }
}

// row e of the batch a.theta, or the copy of it the caller holds (th_row)
__device__ __forceinline__ void mtg_prepare_one(const MtgPrepArgs &a, int64_t e, bool live, const double *th_row = nullptr)
{
    mtg_prepare_from(a, e, live, th_row ? th_row : a.theta + (live ? e : 0) * a.model.P);
}
