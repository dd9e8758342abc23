"""AE_CE_SLICED on conflict-free matchings (round 3) against AE_CE_SEQUENTIAL (bit-exact vs the oracle) on one GPU.
  fidelity [blobs6|mnist] [n] [nb_batch]   full schedule from the dmap init: final CE / edge-length quantiles as ratios to the
                                           sequential mode, for slice thicknesses AE_SL_LAMBDA in LAMBDAS (env, default "0.5,1,2,4")
                                           and for the previous all-optimistic form (AE_SL_NO_MATCH)
  scale [n] [k] [d] [steps]                time per batch on the node-permuted lattice: matchings vs all-optimistic vs rounds
Debug knobs need AE_DEBUG_KNOBS=1 (set here)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["AE_DEBUG_KNOBS"] = "1"
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402
from tools.run_event_check import blobs, edge_q  # noqa: E402


def knobs(**kw):
    for k in ("AE_SL_LAMBDA", "AE_SL_NO_MATCH", "AE_SL_NO_TILE", "AE_SL_TILE_MIN", "AE_SL_TILE_ALWAYS", "AE_SL_CLASS_CAP", "AE_SL_TAIL", "AE_SL_EPT", "AE_SL_NO_SPREAD", "AE_SL_PASSES"):
        os.environ.pop(k, None)
    for k, v in kw.items():
        if v is not None:
            os.environ[k] = str(v)


def schedule(kg, npar, y0, d, mode, nb_batch, seed=4664397, hub=None):
    par = A.EmbedderParams(nb_grad_batch=nb_batch, ce_mode=mode, asked_dim=d, seed=seed, hubness_weighting=hub is not None)
    eo = A.EntropyOptim(kg, npar, par, y0, hub_counts=hub)
    S = 10 * eo.get_nb_edges()
    L.check(L.load().ae_synchronize())
    t0 = time.perf_counter()
    for it in range(1, nb_batch + 1):
        eo.gradient_iteration_threaded(S, 1.0 - it / nb_batch, it)
    L.check(L.load().ae_synchronize())
    dt = (time.perf_counter() - t0) / nb_batch * 1e3
    return eo.get_embedded(), eo.ce_compute_threaded(), dt, eo.samples_drawn()


def fidelity(kind="blobs6", n=60000, nb_batch=40, out_path=None):
    if kind == "blobs6":
        kg = A.KGraph.bruteforce_l2(blobs(n), 6)
        rho, d = 0.75, 2
    else:
        sys.argv = ["bench.py"]
        import bench
        kg = A.KGraph.bruteforce_l2(bench.synth_points(n, 784, seed=1).cpu().numpy(), 12)
        rho, d = 1.0, 2
    indptr, nbr, dist = kg.get_neighbours()
    hubc = kg.hubness()
    print("graph %s n %d: max in-degree %d" % (kind, n, int(hubc.max())), flush=True)
    npar = A.to_proba_edges(kg, rho, 1.0)
    y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(d, 5.0, 12)).embed_from_kgraph(kg), 10.0)
    res = {"kind": kind, "n": n, "nb_batch": nb_batch, "runs": []}
    knobs()
    seeds = [4664397, 12345, 777][:int(os.environ.get("SEEDS", "3"))]
    seq = []
    for sd in seeds:
        y, ce, ms, _ = schedule(kg, npar, y0, d, A.AE_CE_SEQUENTIAL, nb_batch, seed=sd)
        seq.append((ce, edge_q(indptr, nbr, y)))
        print("sequential seed %d  ce %.0f q %s  %.2f ms/batch" % (sd, ce, np.round(seq[-1][1], 4), ms), flush=True)
    ce = float(np.mean([c for c, _ in seq]))
    q = np.mean([qq for _, qq in seq], axis=0)
    print("sequential mean     ce %.0f (spread %.4f) q %s" % (ce, np.std([c for c, _ in seq]) / ce, np.round(q, 4)), flush=True)
    res["sequential"] = dict(ce=ce, q=q.tolist(), ces=[c for c, _ in seq])
    lambdas = [float(x) for x in os.environ.get("LAMBDAS", "0.5,1,2,4").split(",")]
    variants = [("match lambda %g" % lam, dict(AE_SL_LAMBDA=lam)) for lam in lambdas]
    variants += [("match lambda %g no spread" % lam, dict(AE_SL_LAMBDA=lam, AE_SL_NO_SPREAD=1)) for lam in lambdas if lam >= 2]
    variants += [("all-optimistic", dict(AE_SL_NO_MATCH=1))]
    variants += [("all-optimistic %d passes" % p, dict(AE_SL_NO_MATCH=1, AE_SL_PASSES=p)) for p in (2, 1)]
    for name, kw in variants:
        rr = []
        for sd in seeds[:2]:
            knobs(**kw)
            ys, ces, mss, (drawn, rounds) = schedule(kg, npar, y0, d, A.AE_CE_SLICED, nb_batch, seed=sd)
            rr.append((ces / ce, edge_q(indptr, nbr, ys) / q))
        cr = float(np.mean([c for c, _ in rr]))
        qr = np.mean([qq for _, qq in rr], axis=0)
        print("%-28s ce ratio %.4f (%s) q ratio %s  %.2f ms/batch  slices %d" % (name, cr, " ".join("%.4f" % c for c, _ in rr), np.round(qr, 3), mss, rounds), flush=True)
        res["runs"].append(dict(name=name, ce_ratio=cr, ce_ratios=[c for c, _ in rr], q_ratio=qr.tolist(), ms=mss, slices=rounds))
    knobs()
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


def scale(n=1_650_000, k=6, d=2, steps=3, out_path=None):
    sys.argv = ["bench.py"]
    import bench
    indptr, nbr, dst = bench.lattice_graph(n, k, seed=7, permute=True)
    kg = A.KGraph(indptr, nbr, dst, k)
    y0 = A.set_data_box(np.random.default_rng(1).normal(size=(n, d)).astype(np.float32), 10.0)
    npar = A.to_proba_edges(kg, 1.0, 1.0)
    res = {"n": n, "k": k, "d": d, "runs": []}
    bps = 24 + 4 * k + 36 * d
    variants = [("match", A.AE_CE_SLICED, {}), ("match ept2", A.AE_CE_SLICED, dict(AE_SL_EPT=2)), ("match ept4", A.AE_CE_SLICED, dict(AE_SL_EPT=4)),
                ("match lambda 1", A.AE_CE_SLICED, dict(AE_SL_LAMBDA=1)), ("match lambda 2", A.AE_CE_SLICED, dict(AE_SL_LAMBDA=2)),
                ("match lambda 4", A.AE_CE_SLICED, dict(AE_SL_LAMBDA=4)),
                ("match no tile", A.AE_CE_SLICED, dict(AE_SL_NO_TILE=1)), ("match tile always", A.AE_CE_SLICED, dict(AE_SL_TILE_ALWAYS=1, AE_SL_TILE_MIN=1)),
                ("all-optimistic", A.AE_CE_SLICED, dict(AE_SL_NO_MATCH=1)), ("all-optimistic 2 passes", A.AE_CE_SLICED, dict(AE_SL_NO_MATCH=1, AE_SL_PASSES=2)),
                ("all-optimistic 1 pass", A.AE_CE_SLICED, dict(AE_SL_NO_MATCH=1, AE_SL_PASSES=1)),
                ("rounds", A.AE_CE_HOGWILD, {}), ("ordered", A.AE_CE_ORDERED, {}),
                ("sequential", A.AE_CE_SEQUENTIAL, {})]
    if os.environ.get("VARIANTS"):
        keep = os.environ["VARIANTS"].split(",")
        variants = [v for v in variants if v[0] in keep]
    for name, mode, kw in variants:
        knobs(**kw)
        t0 = time.perf_counter()
        r = bench.time_mode(A, L, kg, npar, y0, d, mode, steps, 1)
        frac = bps * r["nb_sample"] / (r["ms_per_step"] * 1e-3) / 8e12
        print("%-18s ms/step %.2f  frac %.3f  ce_after %.0f  (create + %d batches %.1f s)" % (name, r["ms_per_step"], frac, r["ce_after"], steps + 1, time.perf_counter() - t0), flush=True)
        res["runs"].append(dict(name=name, ms_per_step=r["ms_per_step"], frac=frac, ce_after=r["ce_after"]))
        del r
    knobs()
    if out_path:
        json.dump(res, open(out_path, "w"), indent=1)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "fidelity"
    if what == "fidelity":
        fidelity(sys.argv[2] if len(sys.argv) > 2 else "blobs6", int(sys.argv[3]) if len(sys.argv) > 3 else 60000,
                 int(sys.argv[4]) if len(sys.argv) > 4 else 40, sys.argv[5] if len(sys.argv) > 5 else None)
    else:
        scale(int(sys.argv[2]) if len(sys.argv) > 2 else 1_650_000, int(sys.argv[3]) if len(sys.argv) > 3 else 6,
              int(sys.argv[4]) if len(sys.argv) > 4 else 2, int(sys.argv[5]) if len(sys.argv) > 5 else 3, sys.argv[6] if len(sys.argv) > 6 else None)
