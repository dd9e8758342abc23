import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import annembed_amd as A

from tests.test_gpu_configs import _blobs, _edge_q
n,k=60000,6
x=_blobs(n); n_small=n//24
large, small = A.KGraph.bruteforce_l2(x,k), A.KGraph.bruteforce_l2(x[:n_small],k)
x64=x.astype(np.float64)
dd=(x64**2).sum(1)[:,None]+(x64[:n_small]**2).sum(1)[None,:]-2*x64@x64[:n_small].T
pn=dd.argmin(1).astype(np.uint32); pd=np.sqrt(np.maximum(dd.min(1),0)).astype(np.float32)
pn[:n_small]=np.arange(n_small); pd[:n_small]=0
indptr,nbr,_=large.get_neighbours()
for name,mode in (("seq",A.AE_CE_SEQUENTIAL),("seq",A.AE_CE_SEQUENTIAL),("seq",A.AE_CE_SEQUENTIAL),("sliced",A.AE_CE_SLICED),("sliced",A.AE_CE_SLICED),("sliced",A.AE_CE_SLICED),("event",A.AE_CE_EVENT),("event",A.AE_CE_EVENT),("event",A.AE_CE_EVENT),("rounds",A.AE_CE_HOGWILD)):
    par=A.EmbedderParams(asked_dim=2,nb_grad_batch=40,grad_factor=5,scale_rho=0.75,beta=1.0,grad_step=1.0,nb_sampling_by_edge=10,dmap_init=True,hubness_weighting=True,ce_mode=mode)
    emb=A.Embedder.from_hkgraph(A.KGraphProjection(small,large,pn,pd),par)
    emb.embed()
    y=emb.get_embedded(); y0=emb.get_initial_embedding()
    print(name,"ce %.0f"%emb.get_cross_entropy()[1],"q",np.round(_edge_q(indptr,nbr,y),5),"y0[0]",y0[0],"y[0]",y[0],flush=True)
