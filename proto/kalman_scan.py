"""Prototype (numpy) of the time-parallel formulation of the celerite likelihood:
Kalman filter in the SDE basis + Sarkka & Garcia-Fernandez (2021) associative elements.
Checks that chunked composition reproduces the sequential lnL."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import dense

def model_matrices(coeffs):
    """Per-term (F(dt), P_inf, h): real term 1x1, complex term 2x2."""
    ar, cr, ac, bc, cc, dc, jitter = coeffs
    blocks = []
    for a, c in zip(ar, cr):
        blocks.append(("r", a, c))
    for a, b, c, d in zip(ac, bc, cc, dc):
        # P_inf = [[a, -b], [-b, p]], noise LL^T = 2cP - d [[2b, a-p],[a-p,-2b]] >= 0 ; pick p maximizing det
        if d != 0:
            p = (2 * d * (2 * c * b + d * a) + 4 * c * (c * a - d * b)) / (2 * d * d)
        else:
            p = a
        blocks.append(("c", a, b, c, d, p))
    return blocks, jitter

def build(blocks, dx):
    J = sum(1 if b[0] == "r" else 2 for b in blocks)
    F = np.zeros((J, J)); P = np.zeros((J, J)); h = np.zeros(J)
    i = 0
    for b in blocks:
        if b[0] == "r":
            _, a, c = b
            F[i, i] = np.exp(-c * dx); P[i, i] = a; h[i] = 1; i += 1
        else:
            _, a, bb, c, d, p = b
            e = np.exp(-c * dx); cs, sn = np.cos(d * dx), np.sin(d * dx)
            F[i:i+2, i:i+2] = e * np.array([[cs, -sn], [sn, cs]])
            P[i:i+2, i:i+2] = [[a, -bb], [-bb, p]]
            h[i] = 1; i += 2
    return F, P, h

def sequential(t, r, var, blocks, jitter):
    N = len(t)
    F0, Pinf, h = build(blocks, 0.0)
    m = np.zeros(len(h)); C = Pinf.copy()
    ll = 0.0
    for n in range(N):
        if n > 0:
            F, _, _ = build(blocks, t[n] - t[n-1])
            m = F @ m; C = F @ C @ F.T + (Pinf - F @ Pinf @ F.T)
        S = h @ C @ h + var[n] + jitter
        z = r[n] - h @ m
        ll += -0.5 * (np.log(2 * np.pi * S) + z * z / S)
        K = C @ h / S
        m = m + K * z; C = C - np.outer(K, K) * S
    return ll

def element(F, Q, h, R, y):
    """Filtering element of one step (x_k = F x_{k-1} + q, y = h x + r)."""
    S = h @ Q @ h + R
    K = Q @ h / S
    A = (np.eye(len(h)) - np.outer(K, h)) @ F
    b = K * y
    C = Q - np.outer(K, K) * S
    Fh = F.T @ h
    eta = Fh * y / S
    Jm = np.outer(Fh, Fh) / S
    return A, b, C, eta, Jm

def combine(e1, e2):
    A1, b1, C1, eta1, J1 = e1; A2, b2, C2, eta2, J2 = e2
    n = len(b1); I = np.eye(n)
    X = np.linalg.solve(I + C1 @ J2, np.column_stack([A1, (b1 + C1 @ eta2)[:, None], C1]))
    XA, Xb, XC = X[:, :n], X[:, n], X[:, n+1:]
    A = A2 @ XA
    b = A2 @ Xb + b2
    C = A2 @ XC @ A2.T + C2
    Y = np.linalg.solve(I + J2 @ C1, np.column_stack([(eta2 - J2 @ b1)[:, None], J2]))
    eta = A1.T @ Y[:, 0] + eta1
    Jm = A1.T @ Y[:, 1:] @ A1 + J1
    return A, b, 0.5 * (C + C.T), eta, 0.5 * (Jm + Jm.T)

def chunked(t, r, var, blocks, jitter, nchunks):
    """pass 1: compose per-chunk elements; pass 2: propagate (m, C) over chunk boundaries;
    pass 3: sequential filter inside each chunk from its start state."""
    N = len(t)
    F0, Pinf, h = build(blocks, 0.0)
    J = len(h)
    bounds = np.linspace(0, N, nchunks + 1).astype(int)
    elems = []
    for c in range(nchunks):
        e = None
        for n in range(bounds[c], bounds[c+1]):
            if n == 0:
                continue              # the first sample is handled from the prior in pass 3
            F, _, _ = build(blocks, t[n] - t[n-1])
            Q = Pinf - F @ Pinf @ F.T
            # element maps the FILTERED state at n-1 to the filtered state at n
            en = element(F, Q, h, var[n] + jitter, r[n])
            e = en if e is None else combine(e, en)
        elems.append(e)
    # filtered state after sample 0
    m = np.zeros(J); C = Pinf.copy()
    S = h @ C @ h + var[0] + jitter; z0 = r[0]
    ll0 = -0.5 * (np.log(2 * np.pi * S) + z0 * z0 / S)
    K = C @ h / S; m = m + K * z0; C = C - np.outer(K, K) * S
    starts = []
    for c in range(nchunks):
        starts.append((m.copy(), C.copy()))
        if elems[c] is None:
            continue
        A, b, Cc, eta, Jm = elems[c]
        I = np.eye(J)
        m = A @ np.linalg.solve(I + C @ Jm, m + C @ eta) + b
        C = A @ np.linalg.solve(I + C @ Jm, C) @ A.T + Cc
        C = 0.5 * (C + C.T)
    ll = ll0
    for c in range(nchunks):
        m, C = starts[c]
        for n in range(max(bounds[c], 1), bounds[c+1]):
            F, _, _ = build(blocks, t[n] - t[n-1])
            m = F @ m; C = F @ C @ F.T + (Pinf - F @ Pinf @ F.T)
            S = h @ C @ h + var[n] + jitter
            z = r[n] - h @ m
            ll += -0.5 * (np.log(2 * np.pi * S) + z * z / S)
            K = C @ h / S
            m = m + K * z; C = C - np.outer(K, K) * S
    return ll

if __name__ == "__main__":
    from mind_the_gaps_amd import synthetic as synth
    rng = np.random.default_rng(0)
    for kinds in (synth.ALT_MODEL, [synth.K_SHO], [synth.K_DRW, synth.K_BPL], [synth.K_COSINUS, synth.K_DRW], [synth.K_MATERN32]):
        N = 3000
        t, y, dy = synth.make_lightcurves(N, 1, seed=3)
        th = synth.draw_thetas(kinds, 1, seed=2)[0]
        co = dense.build_coeffs(kinds, th)
        blocks, jitter = model_matrices(co)
        r = y[0] - y[0].mean(); var = (dy[0] + 1e-12) ** 2
        want = dense.dense_loglike(t, y[0], dy[0], co, 0, [y[0].mean()])
        seq = sequential(t, r, var, blocks, jitter)
        out = [chunked(t, r, var, blocks, jitter, c) for c in (7, 64, 500)]
        print(kinds, "dense", want, "seq err", abs(seq - want) / abs(want), "chunk errs", [abs(o - want) / abs(want) for o in out])


# ---- likelihood as a REDUCTION: elements carry ell = ln p(y_chunk | x_start = 0) -------------
def combine_ll(e1, e2):
    """(A, b, C, eta, J, ell) o (A, b, C, eta, J, ell): the five-component rule plus
    ell = ell1 + ell2 - 1/2 ln det G + 1/2 [eta2' G^-1 C1 eta2 + 2 eta2' G^-1 b1 - b1' J2 G^-1 b1],
    G = I + C1 J2 (Gaussian integral of e2's information form against N(A1 x + b1, C1))."""
    A, b, C, eta, Jm = combine(e1[:5], e2[:5])
    b1, C1, eta2, J2 = e1[1], e1[2], e2[3], e2[4]
    G = np.eye(len(b1)) + C1 @ J2
    Gi_b1 = np.linalg.solve(G, b1)
    Gi_C1_eta2 = np.linalg.solve(G, C1 @ eta2)
    sign, logdet = np.linalg.slogdet(G)
    ell = e1[5] + e2[5] - 0.5 * logdet + 0.5 * (eta2 @ Gi_C1_eta2 + 2.0 * eta2 @ Gi_b1 - b1 @ J2 @ Gi_b1)
    return A, b, C, eta, Jm, ell


def reduced(t, r, var, blocks, jitter, nchunks):
    """pass 1: per-chunk element by the Kalman recursion (deviation-free form for clarity), the
    first chunk absorbing the prior; then a tree reduction of the chunk elements.  No pass 3."""
    N = len(t)
    F0, Pinf, h = build(blocks, 0.0)
    J = len(h)
    bounds = np.linspace(0, N, nchunks + 1).astype(int)
    elems = []
    for c in range(nchunks):
        if c == 0:   # prior update at sample 0: A = 0, (b, C) = filtered state, ell = ln p(y_0)
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
    while len(elems) > 1:   # pairwise tree
        nxt = [combine_ll(elems[i], elems[i + 1]) for i in range(0, len(elems) - 1, 2)]
        if len(elems) % 2:
            nxt.append(elems[-1])
        elems = nxt
    return elems[0][5]


if __name__ == "__main__":
    print("-- reduction form")
    for kinds in (synth.ALT_MODEL, [synth.K_SHO], [synth.K_DRW, synth.K_BPL], [synth.K_COSINUS, synth.K_DRW], [synth.K_MATERN32]):
        N = 3000
        t, y, dy = synth.make_lightcurves(N, 1, seed=3)
        th = synth.draw_thetas(kinds, 1, seed=2)[0]
        co = dense.build_coeffs(kinds, th)
        blocks, jitter = model_matrices(co)
        r = y[0] - y[0].mean(); var = (dy[0] + 1e-12) ** 2
        want = dense.dense_loglike(t, y[0], dy[0], co, 0, [y[0].mean()])
        out = [reduced(t, r, var, blocks, jitter, c) for c in (1, 7, 64, 256, 1000)]
        print(kinds, "reduction errs", [abs(o - want) / abs(want) for o in out])
