"""the two launch forms of the time-sliced class path (merged slices / one launch per class) against the exact mode on the stiff
k = 6 blobs graph of tests/test_gpu_configs.py::test_k6_blobs_without_hubness_40_batches, many seeds a side.
usage: python tools/run_blobs_forms.py [n_seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import annembed_amd as A  # noqa: E402
import test_gpu_configs as T  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = 60000
g = A.KGraph.bruteforce_l2(T._blobs(n), 6)
indptr, nbr, _ = g.get_neighbours()
npar = A.to_proba_edges(g, 0.75, 1.0)
y0 = (np.random.default_rng(5).random(size=(n, 2)).astype(np.float32) - 0.5)
seeds = [1000 + 7919 * s for s in range(n_seeds)]


def rows(mode, knobs):
    os.environ.update(knobs)
    try:
        out = []
        for sd in seeds:
            y, ce, _ = T._run_ce(A, g, npar, y0, 40, mode, seed=sd)
            out.append(T._metrics(indptr, nbr, y, ce))
        return np.array(out)
    finally:
        for k in knobs:
            os.environ.pop(k, None)


res = {"exact": rows(A.AE_CE_SEQUENTIAL, {})}
res["per class"] = rows(A.AE_CE_SLICED, {"AE_DEBUG_KNOBS": "1", "AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_MERGE": "1"})
res["merged"] = rows(A.AE_CE_SLICED, {"AE_DEBUG_KNOBS": "1", "AE_SL_FORCE_CLASSES": "1", "AE_SL_MERGE": "1"})
res["optimistic"] = rows(A.AE_CE_SLICED, {})
b = res["exact"]
for name, a in res.items():
    se = np.sqrt(a.var(0, ddof=1) / len(a) + b.var(0, ddof=1) / len(b)) / b.mean(0)
    print("%-12s / exact: ce, q25, q50, q75 = %s  2 SE %s  (%d seeds a side)" % (name, np.round(a.mean(0) / b.mean(0), 4), np.round(2 * se, 4), len(a)), flush=True)
a, b = res["merged"], res["per class"]
se = np.sqrt(a.var(0, ddof=1) / len(a) + b.var(0, ddof=1) / len(b)) / b.mean(0)
print("merged / per class: %s  2 SE %s" % (np.round(a.mean(0) / b.mean(0), 4), np.round(2 * se, 4)))
