"""SVD-init at the configs[3] size (BASELINE.json.metric: "SVD-init GFLOP/s, ... Higgs-11M"): the diffusion-map initialisation's randomized
SVD (graphlaplace.rs:97-125) of the laplacian of an 11 M-node graph, timed alone.  usage: python tools/run_svd_init_c4.py [lattice|knn] [n] [ordered: node ids in cluster order instead of shuffled]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, argv = ["bench.py"], sys.argv
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

which = argv[1] if len(argv) > 1 else "lattice"
n = int(argv[2]) if len(argv) > 2 else 11_000_000
if which == "lattice":
    indptr, nbr, dst = bench.lattice_graph(n, 6, seed=7, permute=True)
else:
    gr = bench.config_graphs(A, "c4", n_override=n, permute_seed=None if (len(argv) > 3 and argv[3] == "ordered") else 9)
    indptr, nbr, dst = gr["indptr"], gr["nbr"], gr["dist"]
    print(gr["desc"])
kg = A.KGraph(indptr, nbr, dst, 6)
try:
    print(json.dumps(bench.svd_init_of(A, L, kg, 8)))
except A.AnnembedError as e:
    print("svd init failed:", e)
