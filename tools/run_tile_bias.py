"""Does the LDS tile of negatives bias the time-sliced mode?  1 M Higgs-shaped points (configs[3]'s generator; kNN inside components, node ids permuted), k 6 -> D-dim,
hubness weighting as examples/higgs.rs, 20 batches from a random start; ratios to AE_CE_SEQUENTIAL.  usage: D=8 HUBW=1 python tools/run_tile_bias.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AE_DEBUG_KNOBS"] = "1"
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402

n, k, d, nb = int(os.environ.get("N", "1000000")), 6, int(os.environ.get("D", "8")), int(os.environ.get("NB", "20"))
if os.environ.get("GRAPH") == "comp":   # 64 well separated tight components (the sharded tests' graph) instead of the Higgs-shaped blobs
    x_, bounds_ = bench.mixture_points_gpu(n, 28, 64, seed=5, mean_sigma=10.0)
    indptr, nbr, dist = bench.component_knn_graph(A, x_, bounds_, k, permute_seed=int(os.environ.get("PERMUTE", "9")) or None)
    del x_
else:
    gr = bench.config_graphs(A, "c4", permute_seed=int(os.environ.get("PERMUTE", "9")) or None, n_override=n)
    indptr, nbr, dist = gr["indptr"], gr["nbr"], gr["dist"]
g = A.KGraph(indptr, nbr, dist, k)
hub = g.hubness() if os.environ.get("HUBW", "1") == "1" else None
if os.environ.get("HUBSYN") == "1":   # a heavy-tailed synthetic weighting (1 % of the nodes 500x as likely): a wrong negative law shows
    hub = np.ones(n, np.uint32)
    hub[np.random.default_rng(5).choice(n, n // 100, replace=False)] = 500
npar = A.to_proba_edges(g, 1.0, 1.0)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)


def edge_q(y, qs=(0.05, 0.25, 0.5, 0.75, 0.95)):
    src = np.repeat(np.arange(n), k)
    return np.quantile(np.linalg.norm(y[src] - y[nbr], axis=1), qs)


def run(mode, knobs=None, seed=4664397):
    knobs = knobs or {}
    saved = {q: os.environ.get(q) for q in knobs}
    os.environ.update(knobs)
    try:
        eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=nb, ce_mode=mode, grad_step=1.0, hubness_weighting=hub is not None, seed=seed), y0, hub_counts=hub)
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


seeds = [int(x) for x in os.environ.get("SEEDS", "11,22,33,44").split(",")]
cases = {"sequential": (A.AE_CE_SEQUENTIAL, {}), "sliced, no tile": (A.AE_CE_SLICED, {"AE_SL_NO_TILE": "1"}), "sliced (tile where it applies)": (A.AE_CE_SLICED, {}),
         "sliced classes, no tile": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1"}),
         "lambda 0.125": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "0.125"}),
         "lambda 0.125 every repeat moved": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "0.125", "AE_SL_SPREAD_ALL": "1"}),
         "lambda 0.125 repeats left": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "0.125", "AE_SL_NO_SPREAD": "1"}),
         "lambda 0.25": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "0.25"}),
         "lambda 0.25 every repeat moved": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "0.25", "AE_SL_SPREAD_ALL": "1"}),
         "lambda 0.25 repeats left": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "0.25", "AE_SL_NO_SPREAD": "1"}),
         "lambda 0.5": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "0.5"}),
         "lambda 0.5 every repeat moved": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "0.5", "AE_SL_SPREAD_ALL": "1"}),
         "lambda 0.5 repeats left": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "0.5", "AE_SL_NO_SPREAD": "1"}),
         "lambda 1": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "1"}),
         "lambda 1 every repeat moved": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "1", "AE_SL_SPREAD_ALL": "1"}),
         "lambda 1 repeats left": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_NO_TILE": "1", "AE_SL_NO_FIT": "1", "AE_SL_LAMBDA": "1", "AE_SL_NO_SPREAD": "1"}),
         "thick 0.75": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_TILE_MIN": "1", "AE_SL_LAMBDA": "0.75"}),
         "thick 1.0": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_TILE_MIN": "1", "AE_SL_LAMBDA": "1.0"}),
         "sliced classes, tile forced": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_TILE_MIN": "1"}),
         # round 6: a tile of 128 rows for a workgroup's 256 samples (half a request per event instead of one)
         "tile 128 rows, forced": (A.AE_CE_SLICED, {"AE_SL_FORCE_CLASSES": "1", "AE_SL_TILE_MIN": "1", "AE_SL_DBG": "256"})}
only = os.environ.get("CASES")
res = {}
for name, (mode, knobs) in cases.items():
    if only and not any(name.startswith(o) for o in only.split(";")):
        continue
    rows = []
    for sd in seeds:
        y, ce = run(mode, knobs, sd)
        rows.append(np.concatenate([[ce], edge_q(y)]))
    res[name] = np.array(rows)
    ref = res["sequential"].mean(axis=0)
    print("%-32s (mean CE %.5e)" % (name, res[name][:, 0].mean()))
    print("%-32s CE %.4f +- %.4f   quantiles %s +- %s   (ratios to the sequential mean over %d seeds)" % (
        name, (res[name][:, 0] / ref[0]).mean(), (res[name][:, 0] / ref[0]).std(), np.round((res[name][:, 1:] / ref[1:]).mean(axis=0), 3),
        np.round((res[name][:, 1:] / ref[1:]).std(axis=0), 3), len(seeds)), flush=True)
