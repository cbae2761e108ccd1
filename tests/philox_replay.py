"""Host replay of the device sampler's random stream and moves (csrc/mtg_sampler.hip):
Philox4x32-10 keyed by the seed, counters (iteration, purpose, ensemble, index)."""
import numpy as np

M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
SPLIT, PROPOSE, ACCEPT = 1, 2, 3


def philox(c0, c1, c2, c3, seed):
    c = [np.asarray(v, dtype=np.uint64) & 0xFFFFFFFF for v in np.broadcast_arrays(c0, c1, c2, c3)]
    k0, k1 = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(M0) * c[0]
        p1 = np.uint64(M1) * c[2]
        hi0, lo0, hi1, lo1 = p0 >> np.uint64(32), p0 & mask, p1 >> np.uint64(32), p1 & mask
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + np.uint64(W0)) & mask
        k1 = (k1 + np.uint64(W1)) & mask
    return c


def u01(hi, lo):
    return ((hi << np.uint64(32) | lo) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53


def split(E, W, iteration, seed):
    """Random permutation per ensemble: walkers ranked by their 64-bit Philox key."""
    r = philox(iteration, SPLIT, np.arange(E)[:, None], np.arange(W)[None, :], seed)
    keys = (r[0] << np.uint64(32)) | r[1]
    return np.argsort(keys, axis=1, kind="stable").astype(np.int64)


def run(coords, lnp, log_prob_fn, steps, seed, a=2.0, start_iteration=0):
    """coords [E][W][P], lnp [E][W]; log_prob_fn(q[E*H, P], ensemble_index[E*H]) -> lnP."""
    coords, lnp = coords.copy(), lnp.copy()
    E, W, P = coords.shape
    H = W // 2
    e, k = np.arange(E)[:, None], np.arange(H)[None, :]
    chain, lnps = [], []
    naccept = np.zeros((E, W), dtype=np.int64)
    for it in range(start_iteration, start_iteration + steps):
        perm = split(E, W, it, seed)
        for half in (0, 1):
            r = philox(it, PROPOSE + 16 * half, e, k, seed)
            z = ((a - 1.0) * u01(r[0], r[1]) + 1.0) ** 2 / a
            w = perm[e, half * H + k]
            partner = perm[e, (1 - half) * H + (u01(r[2], r[3]) * H).astype(np.int64)]
            s, c = coords[e, w], coords[e, partner]
            q = c - (c - s) * z[:, :, None]
            new = log_prob_fn(q.reshape(E * H, P), np.repeat(np.arange(E), H)).reshape(E, H)
            ra = philox(it, ACCEPT + 16 * half, e, k, seed)
            with np.errstate(divide="ignore"):
                lu = np.log(u01(ra[0], ra[1]))
            diff = (P - 1) * np.log(z) + new - lnp[e, w]
            acc = diff > lu
            ei, ki = np.nonzero(acc)
            coords[ei, w[ei, ki]] = q[ei, ki]
            lnp[ei, w[ei, ki]] = new[ei, ki]
            naccept[ei, w[ei, ki]] += 1
        chain.append(coords.copy())
        lnps.append(lnp.copy())
    return np.array(chain), np.array(lnps), naccept
