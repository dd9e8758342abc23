"""event-ordered vs sequential, flat Embedder.embed() from the dmap initialisation on the Higgs-shaped 60 k graph (k = 6, scale_rho 0.75)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A  # noqa: E402
from tools.run_event_check import blobs, edge_q  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
hubw = os.environ.get("NOHUB") is None
g = A.KGraph.bruteforce_l2(blobs(n), 6)
indptr, nbr, _ = g.get_neighbours()
res = {}
for name, mode in (("seq", A.AE_CE_SEQUENTIAL), ("event", A.AE_CE_EVENT)):
    ces, qs = [], []
    for seed in range(3):
        par = A.EmbedderParams(asked_dim=2, nb_grad_batch=40, scale_rho=0.75, beta=1.0, grad_step=1.0, nb_sampling_by_edge=10, dmap_init=True,
                               hubness_weighting=hubw, ce_mode=mode, seed=100 + seed)
        e = A.Embedder(g, par)
        e.embed()
        ces.append(e.get_cross_entropy()[1])
        qs.append(edge_q(indptr, nbr, e.get_embedded()))
    res[name] = (np.mean(ces), np.std(ces), np.mean(qs, axis=0))
    print(name, "ce mean %.0f (sd %.0f)" % (res[name][0], res[name][1]), "q", np.round(res[name][2], 5), flush=True)
print("event ce/seq %.4f" % (res["event"][0] / res["seq"][0]), "q/seq", np.round(res["event"][2] / res["seq"][2], 3))
