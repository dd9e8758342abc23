"""CPU test of the N>1 path: world_size 2, gloo.  The per-batch protocol of annembed_amd.dist.ShardedCE
(local gradient iteration on the owned source nodes, then all-gather of the owned rows) is run with the
oracle as compute backend and must reproduce the single-process emulation of the same sharded batch."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD_PATH = os.path.join(ROOT, "tests", "golden", "golden_v1.npz")
NB_BATCH = 3


class OracleBackend:
    def __init__(self, O, gold, y, lo, hi):
        self.eo = O.EntropyOptim(gold["g_indptr"], gold["g_nbr"], gold["proba"], gold["scale"], y, node_lo=lo, node_hi=hi)
        self.lo = lo

    def gradient_iteration(self, nb_sample, grad_step, it):
        from annembed_amd.dist import sample_offset
        self.eo.gradient_iteration(nb_sample, grad_step, it, s_begin=sample_offset(self.lo))

    def current(self):
        return self.eo.y


def _shard_samples(gold, lo, hi):
    ip = gold["g_indptr"].astype(np.int64)
    return 10 * int(ip[hi] - ip[lo])


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as O
    from annembed_amd.dist import ShardedCE, shard_range
    gold = np.load(GOLD_PATH)
    n = len(gold["g_indptr"]) - 1
    lo, hi = shard_range(n, world, rank)
    be = OracleBackend(O, gold, gold["y_box"], lo, hi)
    y_t = torch.from_numpy(be.eo.y)  # shares memory with the oracle's coordinate array
    sh = ShardedCE(be, y_t, n, 2, rank, world)
    for it in range(1, NB_BATCH + 1):
        sh.step(_shard_samples(gold, lo, hi), 1.0 * (1 - it / (NB_BATCH + 1)), it)
    total_ce = sh.all_reduce_sum(be.eo.ce())
    np.save(os.path.join(out_dir, "y_rank%d.npy" % rank), be.eo.y)
    np.save(os.path.join(out_dir, "ce_rank%d.npy" % rank), np.array(total_ce))
    dist.destroy_process_group()


def test_sharded_ce_two_ranks_gloo(tmp_path, oracle):
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    y0 = np.load(tmp_path / "y_rank0.npy")
    y1 = np.load(tmp_path / "y_rank1.npy")
    assert np.array_equal(y0, y1), "replicas differ after the all-gather"
    # single-process emulation of the same protocol
    from annembed_amd.dist import emulate_sharded_batch, shard_range
    gold = np.load(GOLD_PATH)
    n = len(gold["g_indptr"]) - 1
    y = gold["y_box"].copy()
    for it in range(1, NB_BATCH + 1):
        nbs = [_shard_samples(gold, *shard_range(n, world, r)) for r in range(world)]
        y = emulate_sharded_batch(lambda r, lo, hi, yr: OracleBackend(oracle, gold, yr, lo, hi), y, n, world, nbs,
                                  1.0 * (1 - it / (NB_BATCH + 1)), it)
    assert np.array_equal(y, y0)
    # the all-reduced CE equals the CE of the full graph on the merged coordinates
    full = oracle.EntropyOptim(gold["g_indptr"], gold["g_nbr"], gold["proba"], gold["scale"], y0)
    assert abs(float(np.load(tmp_path / "ce_rank0.npy")) - full.ce()) < 1e-9 * full.ce()
    assert np.isfinite(y0).all() and not np.array_equal(y0, gold["y_box"])
