"""Time of creating an AE_CE_SLICED handle (edge colouring, static records) at a scale shape.  usage: python tools/run_create_time.py [n] [k] [d]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AE_DEBUG_KNOBS"] = "1"
os.environ["AE_CE_PROF"] = "1"
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sys.argv = ["bench.py"]
import bench  # noqa: E402
indptr, nbr, dst = bench.lattice_graph(n, k, seed=7, permute=True)
kg = A.KGraph(indptr, nbr, dst, k)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
npar = A.to_proba_edges(kg, 1.0, 1.0)
for rep in range(3):
    L.check(L.load().ae_synchronize())
    t0 = time.perf_counter()
    eo = A.EntropyOptim(kg, npar, A.EmbedderParams(asked_dim=d, ce_mode=A.AE_CE_SLICED), y0)
    L.check(L.load().ae_synchronize())
    print("create %.3f s" % (time.perf_counter() - t0), eo.slice_info(), flush=True)
    del eo
