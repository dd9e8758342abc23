"""event-ordered vs sequential mode on a SMALL graph (the first stage of the hierarchical schedule: n / 24 nodes, 5 x 40 batches from
the dmap initialisation, hubness weighting).  usage: python tools/run_small_graph_check.py [n_small] [batches]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A  # noqa: E402
from tools.run_event_check import blobs, edge_q  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 200
hubw = os.environ.get("NOHUB") is None
x = blobs(60000)[:n]
g = A.KGraph.bruteforce_l2(x, 6)
indptr, nbr, _ = g.get_neighbours()
res = {}
for name, mode in (("seq", A.AE_CE_SEQUENTIAL), ("event", A.AE_CE_EVENT), ("rounds", A.AE_CE_HOGWILD)):
    ces, qs = [], []
    for seed in range(4):
        par = A.EmbedderParams(asked_dim=2, nb_grad_batch=nb, scale_rho=0.75, beta=1.0, grad_step=1.0, nb_sampling_by_edge=10, dmap_init=True,
                               hubness_weighting=hubw, ce_mode=mode, seed=100 + seed)
        e = A.Embedder(g, par)
        e.embed()
        ces.append(e.get_cross_entropy()[1])
        qs.append(edge_q(indptr, nbr, e.get_embedded()))
    res[name] = (np.mean(ces), np.std(ces), np.mean(qs, axis=0))
    print(name, "ce mean %.0f (sd %.0f)" % (res[name][0], res[name][1]), "q", np.round(res[name][2], 5), flush=True)
for name in ("event", "rounds"):
    print(name, "ce/seq %.4f" % (res[name][0] / res["seq"][0]), "q/seq", np.round(res[name][2] / res["seq"][2], 3))
