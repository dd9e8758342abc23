"""In-process A/B of the AE_CE_SLICED step kernel under its debug knobs (one handle, one graph; every setting timed twice).
usage: python tools/run_match_ab.py [n] [k] [d]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AE_DEBUG_KNOBS"] = "1"
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

KN = ("AE_SL_LAMBDA", "AE_SL_NO_TILE", "AE_SL_TILE_MIN", "AE_SL_TILE_ALWAYS", "AE_SL_EPT", "AE_SL_NO_SPREAD", "AE_SL_DBG", "AE_SL_F64")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 11_000_000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    d = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    sys.argv = ["bench.py"]
    import bench
    indptr, nbr, dst = bench.lattice_graph(n, k, seed=7, permute=True)
    kg = A.KGraph(indptr, nbr, dst, k)
    y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
    npar = A.to_proba_edges(kg, 1.0, 1.0)
    os.environ["AE_CE_PROF"] = "1"
    eo = A.EntropyOptim(kg, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=40, ce_mode=A.AE_CE_SLICED), y0)
    os.environ["AE_SL_NO_MATCH"] = "1"
    eo_opt = A.EntropyOptim(kg, npar, A.EmbedderParams(asked_dim=d, nb_grad_batch=40, ce_mode=A.AE_CE_SLICED), y0)
    os.environ.pop("AE_SL_NO_MATCH")
    os.environ.pop("AE_CE_PROF")
    S = 10 * eo.get_nb_edges()
    bps = 24 + 4 * k + 36 * d
    settings = [("base", {}), ("all-optimistic", dict(OPT=1)), ("f64 scalars", dict(AE_SL_F64=1)), ("all-optimistic f64", dict(OPT=1, AE_SL_F64=1)),
                ("ept1", dict(AE_SL_EPT=1)), ("ept2", dict(AE_SL_EPT=2)), ("ept4", dict(AE_SL_EPT=4)), ("no tile", dict(AE_SL_NO_TILE=1)), ("base again", {}),
                ("lambda 0.75", dict(AE_SL_LAMBDA=0.75)), ("lambda 1", dict(AE_SL_LAMBDA=1)), ("all-optimistic lambda 1", dict(OPT=1, AE_SL_LAMBDA=1)),
                ("dbg no math", dict(AE_SL_DBG=1)), ("dbg no stores", dict(AE_SL_DBG=2)), ("dbg no negatives, no tile", dict(AE_SL_DBG=4, AE_SL_NO_TILE=1)),
                ("dbg no record", dict(AE_SL_DBG=8)), ("dbg rows only (no math, stores, negatives, record)", dict(AE_SL_DBG=15, AE_SL_NO_TILE=1)),
                ("dbg rows + stores only", dict(AE_SL_DBG=13, AE_SL_NO_TILE=1))]
    it = 0
    for name, kw in settings:
        for kk in KN:
            os.environ.pop(kk, None)
        h = eo_opt if kw.get("OPT") else eo
        for kk, v in kw.items():
            if kk != "OPT":
                os.environ[kk] = str(v)
        ts = []
        for _ in range(2):
            it += 1
            L.check(L.load().ae_synchronize())
            t0 = time.perf_counter()
            h.gradient_iteration_threaded(S, 0.5, it)
            L.check(L.load().ae_synchronize())
            ts.append((time.perf_counter() - t0) * 1e3)
        print("%-55s %s ms/batch   frac %.3f" % (name, " ".join("%7.2f" % t for t in ts), bps * S / (min(ts) * 1e-3) / 8e12), flush=True)


if __name__ == "__main__":
    main()
