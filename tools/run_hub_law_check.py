"""The hubness-weighted tile (alias-table draws staged in LDS, rounds mode, n >= 2^20) against the gathered form (AE_CE_NO_TILE) and
the time-sliced mode on a graph with REAL, skewed in-degrees: exact kNN graph (k = 6) of 1.2 M points of the 28-d blobs generator.
Same start, same schedule; final CE and edge-length quantiles must agree between the two rounds forms if the tile draws the
reference's law.  usage: python tools/run_hub_law_check.py [n] [nb_batch]   (run twice: with and without AE_DEBUG_KNOBS=1 AE_CE_NO_TILE=1)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A  # noqa: E402
from tools.run_event_check import blobs, edge_q  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 10
kg = A.KGraph.bruteforce_l2(blobs(n), 6)
indptr, nbr, _ = kg.get_neighbours()
hub = kg.hubness()
print("in-degree: max %d, 99.9 %% quantile %d, mean %.2f" % (hub.max(), np.quantile(hub, 0.999), hub.mean()), flush=True)
npar = A.to_proba_edges(kg, 0.75, 1.0)
y0 = (np.random.default_rng(5).random(size=(n, 2)).astype(np.float32) - 0.5)
modes = [("rounds", A.AE_CE_HOGWILD)] + ([("sliced", A.AE_CE_SLICED)] if not os.environ.get("AE_CE_NO_TILE") else [])
for name, mode in modes:
    eo = A.EntropyOptim(kg, npar, A.EmbedderParams(nb_grad_batch=nb, ce_mode=mode, hubness_weighting=True), y0, hub_counts=hub)
    S = 10 * eo.get_nb_edges()
    for it in range(1, nb + 1):
        eo.gradient_iteration_threaded(S, 1.0 - it / nb, it)
    y = eo.get_embedded()
    print("%-7s tile=%s  ce %.0f  q %s" % (name, "off" if os.environ.get("AE_CE_NO_TILE") else "on", eo.ce_compute_threaded(), np.round(edge_q(indptr, nbr, y), 4)), flush=True)
    del eo
