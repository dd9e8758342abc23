"""Do the ranks' shares of a sharded configs[3] batch OVERLAP when they run side by side on ONE GPU?  `world` processes, each running its own
rank's share (no communicator: the other shards' rows stay as they are), started together; prints every rank's ms per batch while all of
them run.  Against tools/run_shard_time.py (a rank's share alone on the GPU) this says what desynchronised step chains buy: a step's
load / compute / store phases do not overlap inside one chain, two chains interleave.
usage: python tools/run_shard_concurrent.py <world> [batches]      (spawns its ranks itself; env of the children = this process's)"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CACHE = "/tmp/ae_c4_graph_world%d.npz"


def child(world, rank, batches):
    sys.argv = ["bench.py"]
    import bench
    import annembed_amd as A
    from annembed_amd import _lib as L
    from annembed_amd.dist import shard_range
    cache = CACHE % world
    if rank == 0:
        gr = bench.config_graphs(A, "c4", permute_seed=None, shuffle_within_shards=world)
        np.savez(cache + ".tmp.npz", indptr=gr["indptr"], nbr=gr["nbr"], dist=gr["dist"])
        os.replace(cache + ".tmp.npz", cache)
    else:
        while not os.path.exists(cache):
            time.sleep(0.5)
        time.sleep(1.0)
        z = np.load(cache)
        gr = {"indptr": z["indptr"], "nbr": z["nbr"], "dist": z["dist"]}
    n, k, d = len(gr["indptr"]) - 1, 6, 8
    g = A.KGraph(gr["indptr"], gr["nbr"], gr["dist"], k)
    hub = g.hubness()
    npar = A.to_proba_edges(g, 1.0, 1.0)
    y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
    lo, hi = shard_range(n, world, rank)
    eo = A.EntropyOptim(g, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=25, ce_mode=A.AE_CE_AUTO, grad_step=1.0, hubness_weighting=True), y0,
                        node_lo=lo, node_hi=hi, hub_counts=hub)
    S = 10 * eo.get_nb_edges()
    eo.gradient_iteration_threaded(S, 1.0, 1)   # warm-up (colouring, buffers)
    L.check(L.load().ae_synchronize())
    open("/tmp/ae_conc_ready_%d_%d" % (world, rank), "w").close()
    while not all(os.path.exists("/tmp/ae_conc_ready_%d_%d" % (world, r)) for r in range(world)):
        time.sleep(0.01)
    ts = []
    for it in range(2, 2 + batches):
        t0 = time.perf_counter()
        eo.gradient_iteration_threaded(S, 1.0 * (1 - it / 25), it)
        L.check(L.load().ae_synchronize())
        ts.append(time.perf_counter() - t0)
    cl, ov, _, slices = eo.slice_info()
    print(json.dumps({"world": world, "rank": rank, "concurrent": True, "ms_per_batch": [round(t * 1e3, 1) for t in ts], "classes": cl, "slices": slices,
                      "events_per_step": S / (slices * max(cl, 1)), "t_start": round(time.time() % 1000, 2)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "child":
        child(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[4]))
    else:
        world = int(sys.argv[1])
        batches = int(sys.argv[2]) if len(sys.argv) > 2 else 6
        for f in [CACHE % world] + ["/tmp/ae_conc_ready_%d_%d" % (world, r) for r in range(world)]:
            if os.path.exists(f):
                os.remove(f)
        ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), str(world), str(r), "child", str(batches)]) for r in range(world)]
        rc = [p.wait() for p in ps]
        print("exit codes", rc, flush=True)
