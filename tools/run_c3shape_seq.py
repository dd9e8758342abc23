import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
import annembed_amd as A
from annembed_amd import _lib as L
n,k,d=1_650_000,6,2
indptr,nbr,dst=bench.lattice_graph(n,k,seed=7,permute=True)
kg=A.KGraph(indptr,nbr,dst,k)
y0=A.set_data_box(np.random.default_rng(1).normal(size=(n,d)).astype(np.float32),10.0)
npar=A.to_proba_edges(kg,1.0,1.0)
r=bench.time_mode(A,L,kg,npar,y0,d,A.AE_CE_SEQUENTIAL,5,2)
print("sequential C3-shape ms/step %.2f dataflow-only %.2f"%(r["ms_per_step"], r["dominant_ms"]))
