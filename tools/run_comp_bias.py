"""Where does the time-sliced mode's bias on the 16-component graph come from?  One device, 64 000 nodes, 20 batches, ratios to SEQUENTIAL."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AE_DEBUG_KNOBS"] = "1"
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402

n, k, d, nb = int(os.environ.get("N", "64000")), 6, int(os.environ.get("D", "2")), 20
x, bounds = bench.mixture_points_gpu(n, 28, 16 if n <= 200000 else 64, seed=5, mean_sigma=10.0)
indptr, nbr, dist = bench.component_knn_graph(A, x, bounds, k, permute_seed=int(os.environ["PERMUTE"]) if "PERMUTE" in os.environ else None)
y0 = A.set_data_box(np.random.default_rng(2).normal(size=(n, d)).astype(np.float32), 10.0)
g = A.KGraph(indptr, nbr, dist, k)
npar = A.to_proba_edges(g, 1.0, 1.0)


def edge_q(y, qs=(0.05, 0.25, 0.5, 0.75, 0.95)):
    src = np.repeat(np.arange(n), k)
    return np.quantile(np.linalg.norm(y[src] - y[nbr], axis=1), qs)


HUB = None


def run(mode, knobs=None, prec=0):
    knobs = knobs or {}
    saved = {q: os.environ.get(q) for q in knobs}
    os.environ.update(knobs)
    try:
        eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=nb, ce_mode=mode, grad_step=1.0, ce_precision=prec, hubness_weighting=HUB is not None), y0,
                            hub_counts=HUB)
        S = 10 * eo.get_nb_edges()
        for it in range(1, nb + 1):
            eo.gradient_iteration_threaded(S, 1.0 * (1 - it / nb), it)
        return eo.get_embedded(), eo.ce_compute_threaded()
    finally:
        for q, v in saved.items():
            if v is None:
                os.environ.pop(q, None)
            else:
                os.environ[q] = v


yr, cer = run(A.AE_CE_SEQUENTIAL)
qr = edge_q(yr)
cases = [("sliced classes, tile forced (windows)", A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_TILE_MIN": "1"}),
         ("sliced optimistic, tile", A.AE_CE_SLICED, {})]
for name, mode, knobs in cases:
    try:
        HUB = np.ones(n, np.uint32) if name.startswith("HUB") else None
        y, ce = run(mode, knobs, 1 if name.endswith('f32') else 0)
        print("%-22s CE ratio %.4f  quantile ratios %s" % (name, ce / cer, np.round(edge_q(y) / qr, 3)), flush=True)
    except Exception as ex:
        print(name, "failed:", str(ex)[:200], flush=True)
