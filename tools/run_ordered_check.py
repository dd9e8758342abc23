"""AE_CE_ORDERED (end points sequentially consistent in the sequential order, negatives unsynchronised) against AE_CE_SEQUENTIAL
(bit-exact vs the oracle): time per batch and fidelity over full schedules.  usage: python tools/run_ordered_check.py [mnist|blobs6] [n]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402
from tools.run_event_check import blobs, edge_q  # noqa: E402


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "mnist"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    if kind == "mnist":
        sys.argv = ["bench.py"]
        import bench
        kg = A.KGraph.bruteforce_l2(bench.synth_points(n, 784, seed=1).cpu().numpy(), 12)
        rho, nb = 1.0, 25
    else:
        kg = A.KGraph.bruteforce_l2(blobs(n), 6)
        rho, nb = 0.75, 40
    indptr, nbr, _ = kg.get_neighbours()
    npar = A.to_proba_edges(kg, rho, 1.0)
    y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).embed_from_kgraph(kg), 10.0)
    ref = None
    for name, mode, seeds in (("sequential", A.AE_CE_SEQUENTIAL, (4664397, 12345, 777)), ("ordered", A.AE_CE_ORDERED, (4664397, 12345, 777)),
                              ("event", A.AE_CE_EVENT, (4664397,)), ("sliced", A.AE_CE_SLICED, (4664397,))):
        ces, qs, ms = [], [], []
        for sd in seeds:
            eo = A.EntropyOptim(kg, npar, A.EmbedderParams(nb_grad_batch=nb, ce_mode=mode, seed=sd), y0)
            S = 10 * eo.get_nb_edges()
            L.check(L.load().ae_synchronize())
            t0 = time.perf_counter()
            for it in range(1, nb + 1):
                eo.gradient_iteration_threaded(S, 1.0 - it / nb, it)
            L.check(L.load().ae_synchronize())
            ms.append((time.perf_counter() - t0) / nb * 1e3)
            ces.append(eo.ce_compute_threaded())
            qs.append(edge_q(indptr, nbr, eo.get_embedded()))
        ce, q = float(np.mean(ces)), np.mean(qs, axis=0)
        if ref is None:
            ref = (ce, q)
        print("%-11s %.2f ms/batch  ce %.0f (ratio %.4f; runs %s)  q ratio %s" % (name, min(ms), ce, ce / ref[0], " ".join("%.4f" % (c / ref[0]) for c in ces),
                                                                                   np.round(q / ref[1], 3)), flush=True)


if __name__ == "__main__":
    main()
