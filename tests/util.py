"""Shared helpers for the test-suite: synthetic kNN graphs (numpy brute force, CPU only)."""
import numpy as np


def gaussian_mixture(n, dim, ncomp=5, seed=0, spread=6.0):
    rng = np.random.default_rng(seed)
    centers = rng.normal(0, spread, size=(ncomp, dim))
    lab = rng.integers(0, ncomp, size=n)
    x = centers[lab] + rng.normal(0, 1.0, size=(n, dim))
    return x.astype(np.float32), lab


def knn_graph(x, k):
    """Exact kNN (L2), self excluded; rows sorted by increasing distance. Returns CSR (u64, u32, f32)."""
    n = x.shape[0]
    x64 = x.astype(np.float64)
    sq = (x64 * x64).sum(1)
    nbr = np.zeros((n, k), np.uint32)
    dist = np.zeros((n, k), np.float32)
    bs = 2048
    for b in range(0, n, bs):
        d2 = sq[b:b + bs, None] + sq[None, :] - 2.0 * x64[b:b + bs] @ x64.T
        d2[np.arange(min(bs, n - b)), np.arange(b, min(b + bs, n))] = np.inf
        idx = np.argpartition(d2, k, axis=1)[:, :k]
        dd = np.take_along_axis(d2, idx, 1)
        o = np.argsort(dd, axis=1, kind="stable")
        idx = np.take_along_axis(idx, o, 1)
        dd = np.take_along_axis(dd, o, 1)
        nbr[b:b + bs] = idx
        dist[b:b + bs] = np.sqrt(np.maximum(dd, 0)).astype(np.float32)
    indptr = (np.arange(n + 1, dtype=np.uint64) * np.uint64(k))
    return indptr, nbr.reshape(-1).copy(), dist.reshape(-1).copy()


def synthetic_graph(n=2000, dim=10, k=8, seed=0, ncomp=5):
    x, lab = gaussian_mixture(n, dim, ncomp, seed)
    return (*knn_graph(x, k), x, lab)
