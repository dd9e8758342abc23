"""ONE scale shape in the time-sliced mode only (profiling: tools/prof_c4knn.sh): python tools/run_step_shape.py c4_knn|c5_knn|c3_knn [batches]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1] if len(sys.argv) > 1 else "c4_knn"
batches = int(sys.argv[2]) if len(sys.argv) > 2 else 2
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

if which == "c4_knn":
    gr, d, hubw = bench.config_graphs(A, "c4"), 8, True
elif which == "c3_knn":   # configs[2]'s large graph: merged slices with the class window
    gr, d, hubw = bench.exact_knn_graph(A, bench.higgs_shaped_points(1_650_000), 6, "Higgs-shaped points"), 2, True
    gr.setdefault("n", 1_650_000); gr.setdefault("k", 6)
else:
    gr, d, hubw = bench.config_graphs(A, "c5"), 16, False
n, k = gr["n"], gr["k"]
kg = A.KGraph(gr["indptr"], gr["nbr"], gr["dist"], k)
hub = kg.hubness() if hubw else None
npar = A.to_proba_edges(kg, 1.0, 1.0)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
eo = A.EntropyOptim(kg, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=25, ce_mode=A.AE_CE_SLICED, grad_step=1.0, hubness_weighting=hubw), y0, hub_counts=hub)
S = 10 * eo.get_nb_edges()
eo.gradient_iteration_threaded(S, 0.96, 1)
L.check(L.load().ae_synchronize())
t0 = time.perf_counter()
for it in range(2, 2 + batches):
    eo.gradient_iteration_threaded(S, 1.0 - it / 25, it)
L.check(L.load().ae_synchronize())
cl, ov, _, slices = eo.slice_info()
print("RESULT %s: %.1f ms per batch, %d classes, overflow %.4f, %d slices, %.0f events per step" % (which, (time.perf_counter() - t0) / batches * 1e3, cl, ov, slices, S / max(1, cl * slices)))
