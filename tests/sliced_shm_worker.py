"""worker of tests/test_gpu_configs.py::test_sharded_sliced_*: one rank of a `world`-rank FAITHFUL sharded CE schedule (AE_CE_AUTO on a
node range -> the time-sliced mode) over the library's shared-memory communicator; all ranks share this box's GPU.
usage: sliced_shm_worker.py <dir> <rank> <world> <segment name> <exchanges per batch> <batches> [seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, rank, world, name, exchanges, nb_batch = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
    import annembed_amd as A
    from annembed_amd.dist import HostMemComm, shard_range
    g0 = np.load(os.path.join(out_dir, "graph.npz"))
    k = int(g0["k"])
    g = A.KGraph(g0["indptr"], g0["nbr"], g0["dist"], k)
    n = len(g0["indptr"]) - 1
    y0 = g0["y0"]
    d = y0.shape[1]
    npar = A.to_proba_edges(g, float(g0["scale_rho"]), 1.0)
    lo, hi = shard_range(n, world, rank)
    comm = HostMemComm(rank, world, name, n * 64 * 4)
    seed = int(sys.argv[7]) if len(sys.argv) > 7 else 4664397
    eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=nb_batch, grad_step=1.0, seed=seed), y0, node_lo=lo, node_hi=hi)  # AE_CE_AUTO
    assert eo.get_ce_mode() == A.AE_CE_SLICED
    comm.attach(eo, exchanges)
    S = 10 * eo.get_nb_edges()
    for it in range(1, nb_batch + 1):
        eo.gradient_iteration_threaded(S, 1.0 * (1 - it / nb_batch), it)
    ce = comm.all_reduce_sum(eo.ce_compute_threaded())
    drawn, _ = eo.samples_drawn()
    np.save(os.path.join(out_dir, "y_rank%d.npy" % rank), eo.get_embedded())
    np.save(os.path.join(out_dir, "info_rank%d.npy" % rank), np.array([ce, float(drawn), float(S * nb_batch), float(eo.comm_bytes())]))
    comm.close()


if __name__ == "__main__":
    main()
