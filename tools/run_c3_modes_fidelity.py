"""configs[2]-shaped fidelity of the modes AE_CE_AUTO can resolve to: exact kNN graph of Higgs-shaped points (k = 6, 2 columns, hubness
weighting, dmap start, 40 batches), the ordered dataflow and the time-sliced mode (merged slices on the class path) against the exact
mode, several seeds a side.  usage: python tools/run_c3_modes_fidelity.py [n] [seeds]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400000
seeds = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "1,2,3,4").split(",")]
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402

x = bench.higgs_shaped_points(n)
kg = A.KGraph.bruteforce_l2(x, 6)
indptr, nbr, _ = kg.get_neighbours()
hub = kg.hubness()
y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).embed_from_kgraph(kg), 10.0)
npar = A.to_proba_edges(kg, 0.75, 1.0)
src = np.repeat(np.arange(n), 6)
nb = 40


def run(mode, seed):
    eo = A.EntropyOptim(kg, npar, A.EmbedderParams(asked_dim=2, nb_grad_batch=nb, grad_step=1.0, seed=seed, ce_mode=mode, hubness_weighting=True), y0, hub_counts=hub)
    S = 10 * eo.get_nb_edges()
    for it in range(1, nb + 1):
        eo.gradient_iteration_threaded(S, 1.0 * (1 - it / nb), it)
    y = eo.get_embedded()
    q = np.quantile(np.linalg.norm(y[src] - y[nbr], axis=1), [0.05, 0.25, 0.5, 0.75])
    info = eo.slice_info() if mode == A.AE_CE_SLICED else None
    return [eo.ce_compute_threaded(), *q], info


rows = {}
for name, mode, knobs in (("sequential", A.AE_CE_SEQUENTIAL, {}), ("ordered", A.AE_CE_ORDERED, {}), ("time-sliced", A.AE_CE_SLICED, {}),
                          # round 6: the merged slices without their class window (the form of rounds 4-5)
                          ("time-sliced, no class window", A.AE_CE_SLICED, {"AE_DEBUG_KNOBS": "1", "AE_SL_WINDOW": "0"})):
    os.environ.update(knobs)
    try:
        out = [run(mode, sd) for sd in seeds]
    finally:
        for kk in knobs:
            os.environ.pop(kk, None)
    rows[name] = np.array([o[0] for o in out])
    print(name, "mean", np.round(rows[name].mean(0), 5), out[0][1], flush=True)
b = rows["sequential"]
for name in ("ordered", "time-sliced", "time-sliced, no class window"):
    a = rows[name]
    se = np.sqrt(a.var(0, ddof=1) / len(a) + b.var(0, ddof=1) / len(b)) / b.mean(0)
    print("%s / sequential (n = %d, %d seeds a side): ce, q05, q25, q50, q75 = %s  2 SE %s" % (name, n, len(seeds), np.round(a.mean(0) / b.mean(0), 4), np.round(2 * se, 4)))
