import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A
from annembed_amd import _lib as L
rng = np.random.default_rng(0)
m, n, l = 60000, 784, 20
a = rng.normal(size=(m, n)).astype(np.float32)
mat = A.MatRepr.from_array2(a)
for rep in range(6):
    L.check(L.load().ae_synchronize())
    t = time.perf_counter(); r = A.SvdApprox(mat).direct_svd(A.RangeRank(20, 5)); dt = time.perf_counter() - t
F = 9*2*m*n*l + 5*4*m*l*l + 4*4*n*l*l + 2*m*n*l
print("direct_svd dense", dt*1e3, "ms", F/dt/1e9, "GFLOP/s", "s0", r.s[:3])
