"""Hub chains of the time-sliced mode (ce_slice_kernels.h, sl_step_body) against AE_CE_SEQUENTIAL on graphs with hubs, small enough for
the sequential mode: (a) 60 k Higgs-shaped blobs, k = 6 (in-degrees up to ~105), the classes forced (the cost model runs a graph of this size optimistically); (b) a star: node 0 is every node's farthest neighbour (in-degree n - 1: chains of hundreds of events per step, across chunk and
workgroup boundaries).  Prints CE and edge-length quantile ratios and the time per batch.
usage: python tools/run_hub_chain_check.py [blobs|star|all]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AE_DEBUG_KNOBS"] = "1"
import annembed_amd as A  # noqa: E402
sys.argv, argv = ["bench.py"], sys.argv
import bench  # noqa: E402

which = argv[1] if len(argv) > 1 else "all"


def edge_q(indptr, nbr, y, qs=(0.05, 0.25, 0.5, 0.75, 0.95)):
    src = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr.astype(np.int64)))
    return np.quantile(np.linalg.norm(y[src] - y[nbr], axis=1), qs)


def run(g, npar, y0, nb, mode, d=2, prec=0):
    par = A.EmbedderParams(asked_dim=d, nb_grad_batch=nb, ce_mode=mode, grad_step=1.0, ce_precision=prec)
    eo = A.EntropyOptim(g, npar, par, y0)
    S = 10 * eo.get_nb_edges()
    t0 = time.perf_counter()
    for it in range(1, nb + 1):
        eo.gradient_iteration_threaded(S, 1.0 * (1 - it / nb), it)
    y = eo.get_embedded()
    dt = (time.perf_counter() - t0) / nb
    info = eo.slice_info() if mode == A.AE_CE_SLICED else None
    drawn = eo.samples_drawn()[0] if mode == A.AE_CE_SLICED else None
    return y, eo.ce_compute_threaded(), dt, info, drawn


def compare(name, g, npar, y0, nb, d, knobs_list):
    indptr, nbr, _ = g.get_neighbours()
    yr, cer, tr, _, _ = run(g, npar, y0, nb, A.AE_CE_SEQUENTIAL, d)
    qr = edge_q(indptr, nbr, yr)
    print("%s: sequential CE %.6g, %.1f ms per batch" % (name, cer, tr * 1e3), flush=True)
    for knobs in knobs_list:
        saved = {k: os.environ.get(k) for k in knobs}
        os.environ.update(knobs)
        try:
            for prec in (0, 1):
                y, ce, t, info, drawn = run(g, npar, y0, nb, A.AE_CE_SLICED, d, prec)
                q = edge_q(indptr, nbr, y)
                S = 10 * len(nbr) * nb
                print("  %s %s: CE ratio %.4f, quantile ratios %s, %.1f ms per batch, slice info %s, drawn/expected %.5f, finite %s" % (
                    knobs, "f32" if prec else "f64", ce / cer, np.round(q / qr, 3), t * 1e3, info, drawn / S, bool(np.isfinite(y).all())), flush=True)
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


if which in ("blobs", "all"):
    n = 60000
    g = A.KGraph.bruteforce_l2(bench.higgs_shaped_points(n), 6)
    npar = A.to_proba_edges(g, 0.75, 1.0)
    y0 = (np.random.default_rng(5).random(size=(n, 2)).astype(np.float32) - 0.5)
    compare("blobs 60k k6", g, npar, y0, 40, 2, [{"AE_SL_FORCE_CLASSES": "1"}, {"AE_SL_CLASS_CAP": "10"}, {"AE_SL_CLASS_CAP": "14"}, {}])
if which in ("star", "all"):
    n, k = 200000, 6
    indptr, nbr, dist = bench.lattice_graph(n, k, seed=11, permute=True)
    nbr = nbr.reshape(n, k)
    rows = np.arange(1, n)
    rows = rows[(nbr[rows] != 0).all(1)]
    nbr[rows, k - 1] = 0
    g = A.KGraph(indptr, nbr.reshape(-1), dist, k)
    print("star: in-degree of node 0 = %d" % int(g.hubness()[0]), flush=True)
    npar = A.to_proba_edges(g, 1.0, 1.0)
    for d in (2, 8):
        y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
        compare("star 200k k6 d%d" % d, g, npar, y0, 6, d, [{"AE_SL_FORCE_CLASSES": "1"}, {}])
