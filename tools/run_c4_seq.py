"""AE_CE_SEQUENTIAL at the C4 shape (11 M nodes, k 6, 8-D; ~105 GB of scratch): time per batch"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

n, k, d = 11_000_000, 6, 8
indptr, nbr, dst = bench.lattice_graph(n, k, seed=7, permute=True)
kg = A.KGraph(indptr, nbr, dst, k)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
npar = A.to_proba_edges(kg, 1.0, 1.0)
r = bench.time_mode(A, L, kg, npar, y0, d, A.AE_CE_SEQUENTIAL, 2, 1)
print("sequential C4-shape ms/step %.1f  dataflow kernel alone %.1f  ce_after %.0f" % (r["ms_per_step"], r["dominant_ms"], r["ce_after"]))
