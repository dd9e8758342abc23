"""worker of tests/test_gpu_parity.py::test_embedder_multi_gpu_entry_two_ranks_one_gpu: one rank of a two-rank embedding through
the library's Embedder-level entry (ae_embedder_set_comm) over the shared-memory communicator; both ranks share this box's GPU.
usage: embedder_shm_worker.py <dir> <rank> <world> <segment name> <flat|hier|faithful|faithful_dmap>
(faithful: the default mode -- AE_CE_AUTO on the ranks' node ranges = the time-sliced mode -- on a graph in ANY node order: embed() partitions it)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, rank, world, name, kind = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    import annembed_amd as A
    from annembed_amd.dist import HostMemComm
    g0 = np.load(os.path.join(out_dir, "graph.npz"))
    g = A.KGraph(g0["indptr"], g0["nbr"], g0["dist"])
    n = len(g0["indptr"]) - 1
    comm = HostMemComm(rank, world, name, n * 64 * 4)
    par = A.EmbedderParams(nb_grad_batch=6, ce_mode=A.AE_CE_HOGWILD, grad_step=1.0)
    if kind in ("faithful", "faithful_dmap"):
        # ce_mode = AE_CE_AUTO; the graph comes in ANY node order: embed() partitions it by locality itself
        par = A.EmbedderParams(nb_grad_batch=12, grad_step=1.0, dmap_init=kind == "faithful_dmap", seed=int(os.environ.get("AE_TEST_SEED", "4664397")))
        e = A.Embedder(g, par)
        e.set_comm(comm, int(os.environ.get("AE_TEST_EXCHANGES", "0")))   # 0: the library's choice (4 per batch)
        assert e.embed() == 1
        import json
        with open(os.path.join(out_dir, "part_%s_rank%d.json" % (kind, rank)), "w") as f:
            json.dump(e.get_partition_report(), f)
        np.save(os.path.join(out_dir, "y_%s_rank%d.npy" % (kind, rank)), e.get_embedded())
        np.save(os.path.join(out_dir, "y0_%s_rank%d.npy" % (kind, rank)), e.get_initial_embedding())
        np.save(os.path.join(out_dir, "ce_%s_rank%d.npy" % (kind, rank)), np.array(e.get_cross_entropy()))
        comm.close()
        return
    if kind == "hier":
        small = A.KGraph(g0["s_indptr"], g0["s_nbr"], g0["s_dist"])
        e = A.Embedder.from_hkgraph(A.KGraphProjection(small, g, g0["proj_node"], g0["proj_dist"]), par)
    else:
        e = A.Embedder(g, par)
    e.set_comm(comm, 2)
    bad = A.Embedder(g, A.EmbedderParams(nb_grad_batch=2, ce_mode=A.AE_CE_SEQUENTIAL))  # the bit-exact mode needs the whole graph on one device: refused on every rank
    bad.set_comm(comm, 1)
    try:
        bad.embed()
        raise SystemExit("AE_CE_AUTO with a communicator did not fail")
    except A.AnnembedError as err:
        assert err.code == 1, err
    assert e.embed() == 1
    np.save(os.path.join(out_dir, "y_%s_rank%d.npy" % (kind, rank)), e.get_embedded())
    np.save(os.path.join(out_dir, "y0_%s_rank%d.npy" % (kind, rank)), e.get_initial_embedding())
    np.save(os.path.join(out_dir, "ce_%s_rank%d.npy" % (kind, rank)), np.array(e.get_cross_entropy()))
    comm.close()


if __name__ == "__main__":
    main()
