#!/usr/bin/env python3
"""Generates tests/golden/loglike_golden.npz + .json (committed fixtures).

The reference holds NO log-likelihood value anywhere in its tests or docs
 and celerite / the reference package cannot be imported
in this image, so these vectors are produced by the definition-level oracle
(oracle/dense.py): dense N x N covariance, LAPACK float64 Cholesky, and for
N <= 64 also mpmath at 50 digits.  The J = 1 cases additionally carry the
closed-form Ornstein-Uhlenbeck likelihood (product of Gaussian conditionals),
an algorithm independent of both.

Run from the repo root:  python tests/golden/make_golden.py
(about two minutes; the N = 10 000 dense cases need ~2 GB of RAM).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import dense  # noqa: E402
from mind_the_gaps_amd import synthetic as synth  # noqa: E402

K = synth

KERNELS = {
    "drw": [K.K_DRW],
    "drw+sho": [K.K_DRW, K.K_SHO],
    "drw+sho+lor": [K.K_DRW, K.K_SHO, K.K_LORENTZIAN],
    "sho_overdamped": [K.K_SHO],
    "cosinus+drw": [K.K_COSINUS, K.K_DRW],
    "bpl": [K.K_BPL],
    "matern32": [K.K_MATERN32],
    "real+complex4+jitter": [K.K_REAL, K.K_COMPLEX4, K.K_JITTER],
    "complex3": [K.K_COMPLEX3],
    "5sho": [K.K_SHO] * 5,
}


def truth_for(name, kinds):
    th = synth.truth(kinds).copy()
    if name == "sho_overdamped":
        th[1] = np.log(0.1)
    if name == "5sho":  # five distinct oscillators
        for i in range(5):
            th[3 * i + 0] = np.log(20.0 + 10 * i)
            th[3 * i + 1] = np.log([3.0, 0.3, 10.0, 1.0, 0.7071067811865476][i])
            th[3 * i + 2] = np.log(2 * np.pi / (5.0 + 6 * i))
    return th


def ou_closed_form(t, y, dy, a, c, mu):
    """lnL of a DRW (a e^{-c tau}) + white noise by the scalar Kalman recursion
    (AR(1) conditionals) -- independent of dense Cholesky and of celerite."""
    r = y - mu
    var = (dy + 1e-12) ** 2
    m, P = 0.0, a
    ll = 0.0
    for n in range(len(t)):
        if n > 0:
            phi = np.exp(-c * (t[n] - t[n - 1]))
            m, P = phi * m, phi * phi * P + a * (1 - phi * phi)
        S = P + var[n]
        v = r[n] - m
        ll += -0.5 * (np.log(2 * np.pi * S) + v * v / S)
        Kg = P / S
        m, P = m + Kg * v, (1 - Kg) * P
    return float(ll)


def main():
    arrays, manifest = {}, []
    rng = np.random.default_rng(20250704)
    cid = 0

    def add_case(name, kinds, N, offset, theta, mean_kind, mean_params, with_mp, t=None, y=None, dy=None):
        nonlocal cid
        if t is None:
            t = synth.make_times(N, rng, offset)
            dy = rng.uniform(0.5, 2.0, N)
            y = 100.0 + 10.0 * rng.standard_normal(N)
            if mean_kind == 1:
                y = y + 0.01 * (t - t[0])
        if mean_params is None:
            mean_params = [float(np.mean(y))]
        co = dense.build_coeffs(kinds, theta)
        ll = dense.dense_loglike(t, y, dy, co, mean_kind, mean_params)
        ll_mp = dense.dense_loglike_mp(t, y, dy, co, mean_kind, mean_params) if with_mp else float("nan")
        ll_ou = float("nan")
        if kinds == [K.K_DRW] and mean_kind == 0:
            ll_ou = ou_closed_form(t, y, dy, co[0][0], co[1][0], mean_params[0])
        key = "c%03d" % cid
        arrays[key + "_t"] = t
        arrays[key + "_y"] = y
        arrays[key + "_dy"] = dy
        manifest.append({
            "id": key, "name": name, "kinds": [int(k) for k in kinds], "N": int(N), "t_offset": float(offset),
            "theta": [float(v) for v in theta], "mean_kind": int(mean_kind),
            "mean_params": [float(v) for v in mean_params],
            "lnL_dense_f64": ll, "lnL_mpmath50": ll_mp, "lnL_ou_closed_form": ll_ou,
        })
        print(key, name, N, offset, ll, ll_mp, ll_ou, flush=True)
        cid += 1
        return t, y, dy

    for name, kinds in KERNELS.items():
        th0 = truth_for(name, kinds)
        for N in (8, 64, 256, 1000):
            if name == "5sho" and N > 256:
                continue
            for offset in (0.0, 5e8):
                if offset != 0.0 and N not in (64, 1000):
                    continue
                t = y = dy = None
                for rep in range(3):
                    th = th0 if rep == 0 else th0 + 0.1 * np.abs(th0) * rng.standard_normal(len(th0))
                    if name == "sho_overdamped":
                        th[1] = min(th[1], np.log(0.45))
                    if name == "bpl":
                        th[1] = min(th[1], th[0] - 0.1)  # keep log_S0 >= log_Q
                    t, y, dy = add_case(name, kinds, N, offset, th, 0, None,
                                        with_mp=(N <= 64 and rep == 0 and offset == 0.0), t=t, y=y, dy=dy)
    # fitted linear mean (gpmodelling.py:99-111 / mean_models.py:24-31)
    for N in (64, 1000):
        kinds = KERNELS["drw+sho"]
        add_case("drw+sho_linear_mean", kinds, N, 0.0, truth_for("drw+sho", kinds), 1, [0.0123, 98.7],
                 with_mp=(N <= 64))
    # zero error bars: sigma = 1e-12 exactly (gpmodelling.py:54)
    t = synth.make_times(256, rng)
    add_case("drw_zero_dy", [K.K_DRW], 256, 0.0, truth_for("drw", [K.K_DRW]), 0, None, False,
             t=t, y=100 + 10 * rng.standard_normal(256), dy=np.zeros(256))
    # BASELINE size
    for name in ("drw", "drw+sho", "drw+sho+lor"):
        add_case(name, KERNELS[name], 10000, 0.0, truth_for(name, KERNELS[name]), 0, None, False)

    np.savez_compressed(os.path.join(HERE, "loglike_golden.npz"), **arrays)
    with open(os.path.join(HERE, "loglike_golden.json"), "w") as fh:
        json.dump({"generator": "tests/golden/make_golden.py", "oracle": "oracle/dense.py",
                   "cases": manifest}, fh, indent=1)
    print("wrote %d cases" % len(manifest))


if __name__ == "__main__":
    main()
