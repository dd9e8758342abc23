"""worker of tests/test_gpu_parity.py::test_sharded_ce_hip_backend_two_ranks_one_gpu: one rank of a 2-rank gloo group, both ranks
on the same GPU, running annembed_amd.dist.ShardedCE over the HIP library (HipBackend)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, nb_batch = sys.argv[1], int(sys.argv[2])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")
    torch.cuda.set_device(0)
    import annembed_amd as A
    from annembed_amd.dist import HipBackend, ShardedCE, device_tensor, shard_range
    g0 = np.load(os.path.join(out_dir, "graph.npz"))
    indptr, nbr, dst, proba, scale, y0 = g0["indptr"], g0["nbr"], g0["dist"], g0["proba"], g0["scale"], g0["y0"]
    n = len(indptr) - 1
    lo, hi = shard_range(n, world, rank)
    g = A.KGraph(indptr, nbr, dst)
    npar = A.NodeParams.from_host(g, proba, scale)
    eo = A.EntropyOptim(g, npar, A.EmbedderParams(nb_grad_batch=nb_batch, ce_mode=A.AE_CE_HOGWILD), y0, node_lo=lo, node_hi=hi)  # the rounds mode, by name: AE_CE_AUTO refuses a shard
    assert eo.get_ce_mode() == A.AE_CE_HOGWILD

    y_dev = device_tensor(eo)
    sh = ShardedCE(HipBackend(eo), y_dev, n, y0.shape[1], rank, world)
    for it in range(1, nb_batch + 1):
        sh.backend.gradient_iteration(10 * eo.get_nb_edges(), 1.0 * (1 - it / nb_batch), it)
        # gloo moves host tensors: the owned rows are staged through the host around the collective (RCCL refuses two ranks on
        # one device; the in-place device collective is ShardedCE._gather / the library's ae_comm path)
        from annembed_amd import _lib as L
        L.check(L.load().ae_synchronize())
        host = y_dev.cpu()
        own = torch.zeros((max(sh.sizes), host.shape[1]))
        own[:hi - lo] = host[lo:hi]
        parts = [torch.empty_like(own) for _ in range(world)]
        dist.all_gather(parts, own)
        y_dev.copy_(torch.cat([parts[r][:sh.sizes[r]] for r in range(world)]).cuda())
        torch.cuda.synchronize()
    t = torch.tensor([eo.ce_compute_threaded()], dtype=torch.float64)
    dist.all_reduce(t)
    np.save(os.path.join(out_dir, "y_rank%d.npy" % rank), eo.get_embedded())
    if rank == 0:
        np.save(os.path.join(out_dir, "ce.npy"), np.array(float(t[0])))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
