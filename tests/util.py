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


def assert_means_close(rows, ref_rows, names, floors, what="", k_se=2.0):
    """A statistical mode against the exact mode, both run with SEVERAL seeds: rows / ref_rows = [runs][metrics] (final cross entropy,
    edge-length quantiles ...).  For every metric |mean / mean_ref - 1| < 2 SE + floor, SE = the standard error of that ratio from the
    scatter of both sides.  One run against one run with a wide bar (and a second chance) says little about a mode whose runs scatter by
    a few per cent; the mean of a few seeds with its own error bar does.  floors: the systematic distance allowed per metric (1 % for the
    cross entropy and the median edge, 3 % for the other quantiles unless a test says why not).  k_se: 2 for the fidelity tests proper
    (a handful of checks whose floors are several SE wide); the parametrised coverage tests -- a hundred checks on tiny graphs with three
    seeds a side, whose SE is itself an estimate from three numbers -- use 4 so that the SUITE does not fail by chance."""
    a, b = np.asarray(rows, np.float64), np.asarray(ref_rows, np.float64)
    assert a.ndim == 2 and b.ndim == 2 and a.shape[1] == b.shape[1] == len(names) == len(floors) and a.shape[0] >= 2 and b.shape[0] >= 2
    ma, mb = a.mean(0), b.mean(0)
    se = np.sqrt(a.var(0, ddof=1) / a.shape[0] + b.var(0, ddof=1) / b.shape[0]) / np.abs(mb)
    ratio = ma / mb
    print("%s: mean ratios %s (2 SE %s) over %d vs %d runs" % (what, dict(zip(names, np.round(ratio, 4))), np.round(2 * se, 4), a.shape[0], b.shape[0]))
    for q, name in enumerate(names):
        assert abs(ratio[q] - 1.0) < k_se * se[q] + floors[q], "%s: %s mean ratio %.4f, allowed 1 +- (%g SE %.4f + %.3f)" % (what, name, ratio[q], k_se, k_se * se[q], floors[q])
