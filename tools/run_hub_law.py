"""Hubness-weighted negatives (NodeSampler, embedder.rs:915-930) in the time-sliced mode's tile, under synthetic heavy-tailed weights: does the
tile keep the reference's law?  300 k Higgs-shaped points, k 6 -> 8 columns, 20 batches from a random start; ratios to AE_CE_SEQUENTIAL with
the same weights (and, for the test's power, the exact mode with UNIFORM negatives, and another seed of the exact mode).
Round 4: the per-row alias tile (every tile row a draw of the alias table) stays inside the seed spread; a two-level tile (alias table over
aligned 16-node windows, row inside the window by weight -- coalesced, 96 instead of 512 requests per workgroup, configs[3] 177 -> 173 ms)
sat 2.5 ... 4.7 % low in CE with the median edge +20 ... +70 % on the class path: under a heavy tail a window is one node, the 256
samples of a workgroup -- the members of a chain are neighbours in space -- share 16 negatives.  Not kept."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AE_DEBUG_KNOBS"] = "1"
sys.argv = ["bench.py"]
import bench
import annembed_amd as A
n, k, d, nb = 300000, 6, 8, 20
gr = bench.config_graphs(A, "c4", permute_seed=9, n_override=n)
indptr, nbr, dist = gr["indptr"], gr["nbr"], gr["dist"]
g = A.KGraph(indptr, nbr, dist, k)
npar = A.to_proba_edges(g, 1.0, 1.0)
y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
def q(y):
    src = np.repeat(np.arange(n), k)
    return np.quantile(np.linalg.norm(y[src] - y[nbr], axis=1), (0.25, 0.5, 0.75))
def run(mode, hub, knobs={}, seed=11):
    os.environ.update(knobs)
    eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=nb, ce_mode=mode, grad_step=1.0, hubness_weighting=hub is not None, seed=seed), y0, hub_counts=hub)
    S = 10 * eo.get_nb_edges()
    for it in range(1, nb + 1):
        eo.gradient_iteration_threaded(S, 1.0 * (1 - it / nb), it)
    for kk in knobs: os.environ.pop(kk)
    return eo.get_embedded(), eo.ce_compute_threaded()
for frac, wt in ((100, 500), (20, 30), (10, 10)):
    hub = np.ones(n, np.uint32); hub[np.random.default_rng(5).choice(n, n // frac, replace=False)] = wt
    # CE is computed with the same formula regardless of the sampler: comparable
    ys, cs = run(A.AE_CE_SEQUENTIAL, hub); ys2, cs2 = run(A.AE_CE_SEQUENTIAL, hub, seed=22); yu, cu = run(A.AE_CE_SEQUENTIAL, None)
    yt, ct = run(A.AE_CE_SLICED, hub, {"AE_SL_TILE_MIN": "1", "AE_SL_FORCE_CLASSES": "1"}); yo, co = run(A.AE_CE_SLICED, hub, {"AE_SL_TILE_MIN": "1", "AE_SL_NO_MATCH": "1"})
    print("1/%d of the nodes x%d: uniform-law/weighted CE %.4f q %s | other seed %.4f %s | sliced tile classes %.4f %s | optimistic %.4f %s" % (
        frac, wt, cu / cs, np.round(q(yu) / q(ys), 3), cs2 / cs, np.round(q(ys2) / q(ys), 3), ct / cs, np.round(q(yt) / q(ys), 3), co / cs, np.round(q(yo) / q(ys), 3)), flush=True)
