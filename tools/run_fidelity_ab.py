import sys, os, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A
from oracle import oracle as O
from tests.util import synthetic_graph
res = {}
for (n, dim, k, seed, ncomp, nb, init) in ((20000, 8, 8, 2, 6, 6, "dmap"), (1500, 8, 6, 3, 1, 8, "dmap"), (20000, 8, 8, 2, 6, 6, "rand")):
    indptr, nbr, dist, _, _ = synthetic_graph(n=n, dim=dim, k=k, seed=seed, ncomp=ncomp)
    g = A.KGraph(indptr, nbr, dist)
    rc, p0, s0 = O.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    if init == "dmap":
        rc, y0, _ = O.dmap_embed_from_kgraph(indptr, nbr, dist, k, O.DiffusionParams(2, 5.0, 12))
    else:
        y0 = np.random.default_rng(0).normal(size=(n, 2)).astype(np.float32)
    y0 = O.set_data_box(y0, 10.0)
    npar = A.NodeParams.from_host(g, p0, s0)
    yo, oce0, oce1 = O.entropy_optimize(indptr, nbr, p0, s0, y0, nb)
    src = np.repeat(np.arange(n), k)
    lo = np.linalg.norm(yo[src] - yo[nbr], axis=1)
    for mode, env in (("s3-pr4", {"AE_CE_STORE": "3", "AE_CE_PER_ROUND": "4"}), ("s3-pr6", {"AE_CE_STORE": "3", "AE_CE_PER_ROUND": "6"}), ("s3-pr12", {"AE_CE_STORE": "3", "AE_CE_PER_ROUND": "12"}), ("s3-pr16", {"AE_CE_STORE": "3", "AE_CE_PER_ROUND": "16"}), ("s2-pr4", {"AE_CE_STORE": "2", "AE_CE_PER_ROUND": "4"})):
        for kk in ("AE_CE_GROUP", "AE_CE_STORE", "AE_CE_PER_ROUND"): os.environ.pop(kk, None)
        os.environ.update(env)
        y, c0, c1 = A.entropy_optimize(g, npar, A.EmbedderParams(nb_grad_batch=nb), y0)
        lg = np.linalg.norm(y[src] - y[nbr], axis=1)
        print(mode, init, "n", n, "ce ratio", round(c1/oce1, 3), "q25/50/75/95 ratios", [round(float(np.quantile(lg,q)/np.quantile(lo,q)),3) for q in (.25,.5,.75,.95)], flush=True)
