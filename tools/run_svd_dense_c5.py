"""The dense range finder (svdapprox.rs:285-333, 721-799) on tall data matrices resident in HBM: ms, TFLOP/s, MFMA and HBM fractions.
usage: python tools/run_svd_dense_c5.py [MxN,MxN,...]   default 60000x784,6250000x128   (the configs[1] matrix; a rank's share of configs[4])"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shapes = sys.argv[1] if len(sys.argv) > 1 else "60000x784,6250000x128"
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

for sh in shapes.split(","):
    m, n = (int(v) for v in sh.split("x"))
    print("SVD_DENSE", json.dumps(bench.svd_dense_shape(A, L, m, n, reps=5 if m * n < 10**9 else 3)), flush=True)
