import sys, os, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A
from oracle import oracle as O
from tests.util import synthetic_graph
n, k = 20000, 10
indptr, nbr, dist, x, lab = synthetic_graph(n=n, dim=10, k=k, seed=7, ncomp=8)
g = A.KGraph(indptr, nbr, dist)
rc, p0, s0 = O.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
rc, y0, _ = O.dmap_embed_from_kgraph(indptr, nbr, dist, k, O.DiffusionParams(2, 5.0, 12))
y0 = O.set_data_box(y0, 10.0)
npar = A.NodeParams.from_host(g, p0, s0)
nb = 20
yo, c0, c1 = O.entropy_optimize(indptr, nbr, p0, s0, y0, nb)
def report(tag, y, ce):
    r = A.quality_estimate_from_edge_length(g, y, 30)
    print(tag, "ce", round(ce), "no-match", r.nb_without_match, "mean match %.2f" % r.mean_nbmatch, "ratio q50 %.3f mean %.3f" % (r.median_ratio, r.mean_ratio), "radius q50 %.4f" % r.radii_quantiles[2], flush=True)
report("oracle-seq ", yo, c1)
ys, _, c2 = O.entropy_optimize(indptr, nbr, p0, s0, y0, nb, seed=999)
report("oracle-seq2", ys, c2)
for mode, env in (("gpu s2 pr8", {"AE_CE_STORE": "2"}), ("gpu s3 pr8", {"AE_CE_STORE": "3"}), ("gpu s2 pr4", {"AE_CE_STORE": "2", "AE_CE_PER_ROUND": "4"}), ("gpu s3 pr4", {"AE_CE_STORE": "3", "AE_CE_PER_ROUND": "4"})):
    for kk in ("AE_CE_STORE", "AE_CE_PER_ROUND"): os.environ.pop(kk, None)
    os.environ.update(env)
    y, _, ce = A.entropy_optimize(g, npar, A.EmbedderParams(nb_grad_batch=nb), y0)
    report(mode, y, ce)
