"""The launch forms of the time-sliced class path (merged slices / one launch per class / the optimistic passes) against the exact mode on the
stiff k = 6 blobs graph of tests/test_gpu_configs.py::test_k6_blobs_without_hubness_40_batches, many seeds a side -- and, round 6, the A/B
variants that look for the merged form's bias (+0.7 % CE, -1.5 % median edge at 48 seeds, round 5):
  late      merged, the negatives' rows read after the lane's dependencies are met, past the caches (AE_SL_DBG bit 64)
  ov_every  merged, the thin overflow class in EVERY slice (not every 8th)
  base11    merged with the base palette (11 classes instead of 15)
  pc15      one launch per class with the wide palette (15 classes)
usage: python tools/run_blobs_forms.py [n_seeds] [variants: comma list of exact,per_class,merged,optimistic,late,ov_every,base11,pc15]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import annembed_amd as A  # noqa: E402
import test_gpu_configs as T  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
which = sys.argv[2].split(",") if len(sys.argv) > 2 else ["exact", "per_class", "merged", "optimistic"]
n = 60000
g = A.KGraph.bruteforce_l2(T._blobs(n), 6)
indptr, nbr, _ = g.get_neighbours()
npar = A.to_proba_edges(g, 0.75, 1.0)
y0 = (np.random.default_rng(5).random(size=(n, 2)).astype(np.float32) - 0.5)
seeds = [1000 + 7919 * s for s in range(n_seeds)]
K = {"AE_DEBUG_KNOBS": "1", "AE_SL_FORCE_CLASSES": "1"}
VARIANTS = {
    "exact": (A.AE_CE_SEQUENTIAL, {}),
    "per_class": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1")),
    "merged": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1")),
    "optimistic": (A.AE_CE_SLICED, {}),
    "late": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1", AE_SL_DBG="64")),
    "ov_every": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1", AE_SL_OV_EVERY_SLICE="1")),
    "base11": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1", AE_SL_BASE_CLASSES="1")),
    "pc15": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1", AE_SL_CLASS_CAP="15")),
    "pc11": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1", AE_SL_CLASS_CAP="11")),
    "pc13": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1", AE_SL_CLASS_CAP="13")),
    "pc19": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1", AE_SL_CLASS_CAP="19")),
    "pc15_ov": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1", AE_SL_CLASS_CAP="15", AE_SL_OV_EVERY_SLICE="1")),
    "pc15_nolines": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1", AE_SL_CLASS_CAP="15", AE_SL_NO_LINES="1")),
    "pc11_nolines": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1", AE_SL_NO_LINES="1")),
    "pc_snap1": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1", AE_SL_NEG_SNAPSHOT="1")),
    "pc_snap4": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1", AE_SL_NEG_SNAPSHOT="4")),
    "pc_snap16": (A.AE_CE_SLICED, dict(K, AE_SL_NO_MERGE="1", AE_SL_NEG_SNAPSHOT="16")),
    # the class window of the merged launches: a workgroup starts once the classes `window` positions before it are through, rows written through
    "win1": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1", AE_SL_WINDOW="1")),
    "win2": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1", AE_SL_WINDOW="2")),
    "win3": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1", AE_SL_WINDOW="3")),
    "win4": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1", AE_SL_WINDOW="4")),
    "win8": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1", AE_SL_WINDOW="8")),
    "win6": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1", AE_SL_WINDOW="6")),
    "win32": (A.AE_CE_SLICED, dict(K, AE_SL_MERGE="1", AE_SL_WINDOW="32")),   # (beyond the palette: nobody waits -- what the written-through rows and the loads past the caches do alone)
    "ordered": (A.AE_CE_ORDERED, {}),
    "event": (A.AE_CE_EVENT, {}),
}


def rows(mode, knobs):
    os.environ.update(knobs)
    try:
        out = []
        t0 = time.perf_counter()
        for sd in seeds:
            y, ce, _ = T._run_ce(A, g, npar, y0, 40, mode, seed=sd)
            out.append(T._metrics(indptr, nbr, y, ce))
        print("  (%.2f s per seed: 40 batches, handle and metrics included)" % ((time.perf_counter() - t0) / len(seeds)), flush=True)
        return np.array(out)
    finally:
        for k in knobs:
            os.environ.pop(k, None)


def ratio(a, b):
    se = np.sqrt(a.var(0, ddof=1) / len(a) + b.var(0, ddof=1) / len(b)) / b.mean(0)
    return np.round(a.mean(0) / b.mean(0), 4), np.round(2 * se, 4)


res = {}
for name in which:
    res[name] = rows(*VARIANTS[name])
    if "exact" in res:
        r, se = ratio(res[name], res["exact"])
        print("%-10s / exact: ce, q25, q50, q75 = %s  2 SE %s  (%d seeds a side)" % (name, r, se, n_seeds), flush=True)
if "per_class" in res:
    for name in which:
        if name not in ("exact", "per_class"):
            r, se = ratio(res[name], res["per_class"])
            print("%-10s / per_class: %s  2 SE %s" % (name, r, se), flush=True)
