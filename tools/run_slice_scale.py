"""AE_CE_SLICED vs AE_CE_SEQUENTIAL vs AE_CE_HOGWILD at a scale shape on the node-permuted lattice (time per batch; CE after a few batches)
usage: python tools/run_slice_scale.py [n] [k] [d] [steps]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_650_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 4
indptr, nbr, dst = bench.lattice_graph(n, k, seed=7, permute=os.environ.get("NO_PERMUTE") is None)
kg = A.KGraph(indptr, nbr, dst, k)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
npar = A.to_proba_edges(kg, 1.0, 1.0)
modes = [("sliced", A.AE_CE_SLICED)] if os.environ.get("ONLY_SLICED") else [("sliced", A.AE_CE_SLICED), ("rounds", A.AE_CE_HOGWILD)]
if os.environ.get("WITH_SEQ", "1") == "1":
    modes.append(("sequential", A.AE_CE_SEQUENTIAL))
for name, mode in modes:
    r = bench.time_mode(A, L, kg, npar, y0, d, mode, steps, 1)
    print("%-10s ms/step %.2f  ce_after %.0f  launches/batch %d" % (name, r["ms_per_step"], r["ce_after"], r["rounds"]), flush=True)
    del r
