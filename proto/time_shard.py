"""Prototype (numpy) of TIME sharding for the rank-10 path (BASELINE configs[4]: N = 2e5 samples, five SHO terms, 512
walkers on 8 GPUs).  Walker sharding splits the 256 rows of a half-step over the GPUs and is limited by what 32 rows
cost on one GPU (0.53 ms against 2.5 ms for 256 rows: 4.8 x before the exchange).  The composition
pass, 85-95 % of that time, is parallel in TIME as well: every GPU can take one contiguous eighth of the light curve
for ALL rows, reduce it to ONE filtering element per row -- (A, b, C, eta, J, ell): 231 doubles at rank 10 -- and an
all-gather of those elements (8 x 256 x 231 doubles = 3.8 MB per half-step, one collective) lets every rank finish the
likelihood of every row with seven combinations.  Every GPU then composes 256 rows x N/8 samples: the same work as 32
rows x N samples, but with no serial floor that grows with N / chunks -- an eighth of the one-GPU time to first order.

The element algebra is proto/kalman_scan.py's (`reduced`): the first shard absorbs the prior at sample 0; a later shard
starts from (A, b, C, eta, J, ell) = (I, 0, 0, 0, 0, 0), the identity of the combination.

    python proto/time_shard.py          five SHO terms, N = 6000: shards 1, 2, 4, 8 against the dense likelihood
tests/test_distributed.py::test_time_sharded_likelihood_two_ranks runs two gloo ranks through `shard_element` + an
all-gather + `combine_shards` against the C oracle."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from proto.kalman_scan import build, combine_ll, model_matrices


def shard_element(t, r, var, blocks, jitter, lo, hi, nchunks=8):
    """The filtering element of samples [lo, hi) -- what ONE rank computes: `nchunks` chunk elements by the Kalman
    recursion (on the GPU: the composition kernel, one workgroup quartet per 64 chunks), reduced pairwise."""
    _, Pinf, h = build(blocks, 0.0)
    J = len(h)
    bounds = np.linspace(lo, hi, nchunks + 1).astype(int)
    elems = []
    for c in range(nchunks):
        if bounds[c] == 0:   # the prior at sample 0: A = 0, (b, C) = filtered state, ell = ln p(y_0)
            A = np.zeros((J, J)); C = Pinf.copy()
            S = h @ C @ h + var[0] + jitter
            ell = -0.5 * (np.log(2 * np.pi * S) + r[0] ** 2 / S)
            K = C @ h / S
            b = K * r[0]; C = C - np.outer(K, K) * S
        else:
            A = np.eye(J); b = np.zeros(J); C = np.zeros((J, J)); ell = 0.0
        eta = np.zeros(J); Jm = np.zeros((J, J))
        for n in range(max(bounds[c], 1), bounds[c + 1]):
            F, _, _ = build(blocks, t[n] - t[n - 1])
            FA = F @ A
            g = h @ FA
            b = F @ b; C = F @ C @ F.T + (Pinf - F @ Pinf @ F.T)
            D = h @ C @ h + var[n] + jitter
            z = r[n] - h @ b
            K = C @ h / D
            b = b + K * z; C = C - np.outer(K, K) * D
            A = FA - np.outer(K, g)
            eta = eta + g * z / D
            Jm = Jm + np.outer(g, g) / D
            ell += -0.5 * (np.log(2 * np.pi * D) + z * z / D)
        elems.append((A, b, C, eta, Jm, ell))
    while len(elems) > 1:
        nxt = [combine_ll(elems[i], elems[i + 1]) for i in range(0, len(elems) - 1, 2)]
        if len(elems) % 2:
            nxt.append(elems[-1])
        elems = nxt
    return elems[0]


def pack(e):
    """Element -> flat float64 vector (what crosses the GPUs): A | b | C | eta | J | ell."""
    A, b, C, eta, Jm, ell = e
    return np.concatenate([A.ravel(), b, C.ravel(), eta, Jm.ravel(), [ell]])


def unpack(v, J):
    o = 0
    A = v[o:o + J * J].reshape(J, J); o += J * J
    b = v[o:o + J]; o += J
    C = v[o:o + J * J].reshape(J, J); o += J * J
    eta = v[o:o + J]; o += J
    Jm = v[o:o + J * J].reshape(J, J); o += J * J
    return A, b, C, eta, Jm, float(v[o])


def combine_shards(elements):
    """The shards' elements in time order -> lnL (what EVERY rank does after the all-gather)."""
    e = elements[0]
    for nxt in elements[1:]:
        e = combine_ll(e, nxt)
    return e[5]


def time_bounds(N, shards):
    return np.linspace(0, N, shards + 1).astype(int)


if __name__ == "__main__":
    from mind_the_gaps_amd import synthetic as synth
    from oracle import dense
    kinds = [synth.K_SHO] * 5
    th = np.concatenate([[np.log(20.0 + 10 * i), np.log([3.0, 8.0, 10.0, 1.0, 0.8][i]), np.log(2 * np.pi / (5.0 + 6 * i))]
                         for i in range(5)])
    N = 6000
    t, y, dy = synth.make_lightcurves(N, 1, seed=3)
    co = dense.build_coeffs(kinds, th)
    blocks, jitter = model_matrices(co)
    r = y[0] - y[0].mean(); var = (dy[0] + 1e-12) ** 2
    want = dense.dense_loglike(t, y[0], dy[0], co, 0, [y[0].mean()])
    for G in (1, 2, 4, 8):
        b = time_bounds(N, G)
        got = combine_shards([shard_element(t, r, var, blocks, jitter, b[g], b[g + 1]) for g in range(G)])
        print("J = 10, N = %d, %d time shard(s): lnL %.10f  dense %.10f  rel. diff %.2e  (%d doubles per shard and row)"
              % (N, G, got, want, abs(got - want) / abs(want), len(pack(shard_element(t, r, var, blocks, jitter, 0, 50)))))
