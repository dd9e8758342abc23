"""AE_CE_SLICED on the exact kNN graph of the Higgs-shaped points (real in-degree skew, hubness-weighted negatives: bench.py's c3_knn_shape),
alone -- for a per-batch kernel table under rocprofv3.  usage: python tools/run_c3knn_sliced.py [n] [steps] [hub 0/1]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_650_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
use_hub = (int(sys.argv[3]) if len(sys.argv) > 3 else 1) != 0
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

x = bench.higgs_shaped_points(n)
kg = A.KGraph.bruteforce_l2(x, 6)
hub = kg.hubness()
print("max in-degree %d" % int(hub.max()), flush=True)
y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).embed_from_kgraph(kg), 10.0)
npar = A.to_proba_edges(kg, 1.0, 1.0)
r = bench.time_mode(A, L, kg, npar, y0, 2, A.AE_CE_SLICED, steps, 1, hub=hub if use_hub else None)
print("sliced ms/step %.2f  slices %s  info %s" % (r["ms_per_step"], r.get("rounds"), r["eo"].slice_info()), flush=True)
