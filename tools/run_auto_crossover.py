"""where AE_CE_AUTO should hand over from the ordered dataflow to the time-sliced mode now that under-filled slices run merged: both
modes on the exact kNN graph of Higgs-shaped points (hubness-weighted negatives, dmap start), several sizes.
usage: python tools/run_auto_crossover.py [sizes, e.g. 400000,800000,1200000,1650000]   (AE_DEBUG_KNOBS=1 AE_SL_FORCE_CLASSES=1 for the class path)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sizes = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "400000,800000,1200000,1650000").split(",")]
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

for n in sizes:
    x = bench.higgs_shaped_points(n)
    kg = A.KGraph.bruteforce_l2(x, 6)
    hub = kg.hubness()
    y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).embed_from_kgraph(kg), 10.0)
    npar = A.to_proba_edges(kg, 1.0, 1.0)
    out = {}
    for name, mode in (("ordered", A.AE_CE_ORDERED), ("sliced", A.AE_CE_SLICED)):
        r = bench.time_mode(A, L, kg, npar, y0, 2, mode, 4, 1, hub=hub)
        out[name] = r["ms_per_step"]
        info = r["eo"].slice_info() if mode == A.AE_CE_SLICED else None
        del r
    print("n %d (%d M samples per batch, max in-degree %d): ordered %.2f ms, sliced %.2f ms %s" % (n, 60 * n // 1000000, int(hub.max()), out["ordered"], out["sliced"], info), flush=True)
    del kg, npar
