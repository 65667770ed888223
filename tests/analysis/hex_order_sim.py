"""Which visiting order lets the HexPlane backward's LDS windows catch the taps?  2 M uniform points, runs of 256 consecutive points
along (a) the Z-order (Morton) curve, (b) the Hilbert curve; per scale the fraction of spatial-plane footprints that lie inside a
12 x 12 window anchored at the run's smallest tap.  CPU only (numpy)."""
import sys
import numpy as np

sys.path.insert(0, ".")


def morton_key(q):
    def spread(v):
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        return (v | (v << 2)) & 0x09249249
    return spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)


def hilbert_key(q, bits=10):
    """3-D Hilbert index (Skilling's transpose algorithm, vectorised)."""
    X = [q[:, 0].copy(), q[:, 1].copy(), q[:, 2].copy()]
    n = 3
    M = 1 << (bits - 1)
    Q = M
    while Q > 1:
        P = Q - 1
        for i in range(n):
            m = (X[i] & Q) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = (X[0] ^ X[i]) & P
            t = np.where(m, 0, t)
            X[0] ^= t
            X[i] ^= t
        Q >>= 1
    for i in range(1, n):
        X[i] ^= X[i - 1]
    t = np.zeros_like(X[0])
    Q = M
    while Q > 1:
        t = np.where((X[n - 1] & Q) != 0, t ^ (Q - 1), t)
        Q >>= 1
    for i in range(n):
        X[i] ^= t
    key = np.zeros_like(X[0])
    for b in range(bits - 1, -1, -1):
        for i in range(n):
            key = (key << 1) | ((X[i] >> b) & 1)
    return key


def coverage(pts01, order, run=256, sw=12):
    out = {}
    p = pts01[order]
    nrun = len(p) // run
    p = p[:nrun * run].reshape(nrun, run, 3)
    for res in (64, 128, 256, 512):
        ix = np.floor(p * (res - 1)).astype(np.int64)          # tap cell x0 (x1 = x0 + 1)
        anc = ix.min(axis=1, keepdims=True)
        rel = ix - anc
        ext = rel.max(axis=1) + 2
        inside = []
        for (a, b) in ((0, 1), (0, 2), (1, 2)):
            inside.append(((rel[:, :, a] + 1 < sw) & (rel[:, :, b] + 1 < sw)).mean())
        out[res] = (round(float(np.mean(inside)), 3), np.round(ext.mean(axis=0), 1).tolist(), np.round(np.median(ext, axis=0), 1).tolist())
    return out


if __name__ == "__main__":
    N = 2_000_000
    rng = np.random.default_rng(0)
    pts = rng.random((N, 3))
    q = np.minimum((pts * 1023).astype(np.int64), 1023)
    for name, key in (("morton", morton_key(q)), ("hilbert", hilbert_key(q))):
        order = np.argsort(key, kind="stable")
        print(name)
        for res, v in coverage(pts, order).items():
            print(f"  res {res}: inside {v[0]}  mean extent {v[1]}  median extent {v[2]}")
