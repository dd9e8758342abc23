import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import annembed_amd as A
rng = np.random.default_rng(3)
for n in (1650000, 11000000):
    k = 6
    y = (rng.normal(size=(n, 2)) * (1 + 3 * rng.random((n, 1)))).astype(np.float32)
    nbr = rng.integers(0, n, size=(n, k)).astype(np.uint32)
    g = A.KGraph(np.arange(n + 1, dtype=np.uint64) * np.uint64(k), nbr.reshape(-1), np.sort(rng.random((n, k)).astype(np.float32), axis=1).reshape(-1), k)
    for nbng in (6, 100):
        t0 = time.perf_counter()
        r = A.quality_estimate_from_edge_length(g, y, nbng)
        print("n %d nbng %d: quality estimate %.3f s  median radius %.5f" % (n, nbng, time.perf_counter() - t0, r.radii_quantiles[2]), flush=True)
