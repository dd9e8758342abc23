"""Per-rank compute of the strong-scaling run (bench.py --gpus N: C4 shape, rounds mode, source nodes sharded) measured on ONE
GPU: the batch time of rank 0's shard for N = 1, 2, 4, 8 (no exchange).  With the all-gather modelled (308 MB received per rank
and exchange at N = 8) this is an ESTIMATE of the scaling curve, not a measurement of it.  usage: python tools/run_rank_compute.py [n k d]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402
from annembed_amd.dist import shard_range  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
indptr, nbr, dst = bench.lattice_graph(n, k, seed=7, permute=True)
kg = A.KGraph(indptr, nbr, dst, k)
y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(d, 5.0, 12)).embed_from_kgraph(kg), 10.0)
npar = A.to_proba_edges(kg, 1.0, 1.0)
base = None
for world in (1, 2, 4, 8):
    lo, hi = shard_range(n, world, 0)
    r = bench.time_mode(A, L, kg, npar, y0, d, A.AE_CE_HOGWILD, 4, 2, lo=lo, hi=hi)
    r.pop("eo")
    ms = r["ms_per_step"]
    base = base or ms
    gather_ms = 0.0 if world == 1 else (world - 1) / world * n * d * 4 / 300e9 * 1e3  # ring all-gather at ~300 GB/s of bus bandwidth
    print("N %d: rank-0 batch %.2f ms (%.2fx of N = 1 / N)  + modelled all-gather %.2f ms  -> estimated speed-up %.2fx" % (
        world, ms, ms / (base / world), gather_ms, base / (ms + gather_ms)), flush=True)
