"""Fidelity of the SHARDED CE loop (rounds mode, source nodes split over `world` shards, owned rows exchanged E times per
batch) against the un-sharded sequential mode (bit-exact vs the oracle), measured on ONE GPU with the lockstep entry
(ae_entropy_optim_gradient_iteration_lockstep: kernel for kernel and exchange for exchange what `world` processes with a
communicator attached run).  usage: python tools/run_shard_fidelity.py [blobs6|mnist] [n] [nb_batch] [out.json]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import annembed_amd as A  # noqa: E402
from annembed_amd.dist import shard_range  # noqa: E402
from tools.run_event_check import blobs, edge_q  # noqa: E402


def main():
    out_path = sys.argv[4] if len(sys.argv) > 4 else None
    kind = sys.argv[1] if len(sys.argv) > 1 else "blobs6"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
    nb_batch = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    if kind == "blobs6":
        kg = A.KGraph.bruteforce_l2(blobs(n), 6)
        rho, d = 0.75, 2
    else:
        sys.argv = ["bench.py"]
        import bench
        kg = A.KGraph.bruteforce_l2(bench.synth_points(n, 784, seed=1).cpu().numpy(), 12)
        rho, d = 1.0, 2
    indptr, nbr, dist = kg.get_neighbours()
    npar = A.to_proba_edges(kg, rho, 1.0)
    y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(d, 5.0, 12)).embed_from_kgraph(kg), 10.0)
    out = {"kind": kind, "n": n, "nb_batch": nb_batch, "runs": []}

    def finish(name, world, exch, y, rounds):
        full = A.EntropyOptim(kg, npar, A.EmbedderParams(nb_grad_batch=nb_batch, ce_mode=A.AE_CE_SEQUENTIAL, asked_dim=d), y)
        ce = full.ce_compute_threaded()
        q = edge_q(indptr, nbr, y)
        out["runs"].append(dict(mode=name, world=world, exchanges=exch, rounds=rounds, ce=ce, q=q.tolist()))
        print("%-10s world %d exchanges %2d rounds %2d  ce %.0f  q %s" % (name, world, exch, rounds, ce, np.round(q, 4)), flush=True)

    par = A.EmbedderParams(nb_grad_batch=nb_batch, ce_mode=A.AE_CE_SEQUENTIAL, asked_dim=d)
    eo = A.EntropyOptim(kg, npar, par, y0)
    S = 10 * eo.get_nb_edges()
    for it in range(1, nb_batch + 1):
        eo.gradient_iteration_threaded(S, 1.0 - it / nb_batch, it)
    finish("sequential", 1, 0, eo.get_embedded(), 1)
    del eo
    par = A.EmbedderParams(nb_grad_batch=nb_batch, ce_mode=A.AE_CE_HOGWILD, asked_dim=d)
    for world in (1, 2, 4, 8):
        for exch in ((1,) if world == 1 else (1, 4, 1000)):
            shards = [A.EntropyOptim(kg, npar, par, y0, node_lo=shard_range(n, world, r)[0], node_hi=shard_range(n, world, r)[1]) for r in range(world)]
            ns = [10 * sh.get_nb_edges() for sh in shards]
            for it in range(1, nb_batch + 1):
                A.EntropyOptim.gradient_iteration_lockstep(shards, ns, 1.0 - it / nb_batch, it, exch)
            ys = [sh.get_embedded() for sh in shards]
            for y in ys[1:]:
                assert np.array_equal(y, ys[0]), "replicas differ after the closing exchange"
            finish("rounds", world, exch, ys[0], shards[0].samples_drawn()[1])
            del shards
    ref = out["runs"][0]
    for r in out["runs"][1:]:
        r["ce_vs_sequential"] = r["ce"] / ref["ce"]
        r["q_vs_sequential"] = (np.array(r["q"]) / np.array(ref["q"])).tolist()
        print("world %d exchanges %4d: ce / sequential %.3f   quantiles / sequential %s" % (r["world"], r["exchanges"], r["ce_vs_sequential"], np.round(r["q_vs_sequential"], 3)))
    if out_path:
        json.dump(out, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    main()
