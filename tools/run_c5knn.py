"""configs[4]'s per-GPU shard size on a graph with the in-degree skew of HIGH-dimensional data: the exact kNN graph (k = 10) of 6.25 M points
of a 128-D Gaussian mixture (isotropic components: strong hubness), asked_dim 16 -- the time-sliced mode (what AE_CE_AUTO picks at 625 M
samples per batch) and the rounds mode, uniform and hubness-weighted negatives.  usage: python tools/run_c5knn.py [n] [dim] [k]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6_250_000
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
k = int(sys.argv[3]) if len(sys.argv) > 3 else 10
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

rng = np.random.default_rng(3)
ncomp = 32
means = rng.normal(size=(ncomp, dim)).astype(np.float32) * 3.0
lab = rng.integers(0, ncomp, n)
x = means[lab]
for b in range(0, n, 1 << 20):
    e = min(n, b + (1 << 20))
    x[b:e] += rng.normal(size=(e - b, dim)).astype(np.float32)
t0 = time.perf_counter()
kg = A.KGraph.bruteforce_l2(x, k)
print("exact kNN graph of %d x %d, k %d in %.1f s" % (n, dim, k, time.perf_counter() - t0), flush=True)
del x
hub = kg.hubness()
print("in-degree: max %d, 99.9 %% quantile %d, nodes nobody points at %d" % (int(hub.max()), int(np.quantile(hub, 0.999)), int((hub == 0).sum())), flush=True)
d = 16
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
npar = A.to_proba_edges(kg, 1.0, 1.0)
for hw in (False, True):
    h = hub if hw else None
    auto = A.EntropyOptim(kg, npar, A.EmbedderParams(asked_dim=d, hubness_weighting=hw), y0, hub_counts=h)
    print("hubness weighting %s: AE_CE_AUTO resolves to mode %d, slice info %s" % (hw, auto.get_ce_mode(), auto.slice_info()), flush=True)
    del auto
    for name, mode in (("sliced", A.AE_CE_SLICED), ("rounds", A.AE_CE_HOGWILD)):
        if name not in os.environ.get("MODES", "sliced,rounds").split(","):
            continue
        r = bench.time_mode(A, L, kg, npar, y0, d, mode, 2, 1, hub=h)
        print("  %s ms/step %.1f ce_after %.0f" % (name, r["ms_per_step"], r["ce_after"]), flush=True)
        del r
