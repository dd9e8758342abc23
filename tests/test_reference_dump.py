"""The reference pin kit: stages of a REAL annembed run, dumped by the `ref_dump` Rust test, against the oracle
and the HIP library.  The files are absent on the build and GPU boxes (no Rust toolchain, no network): the comparisons skip, and a CPU test
keeps the reader and the oracle-side checks alive on a dump of the same format written from the oracle's own run."""
import os

import numpy as np
import pytest

from tests.util import synthetic_graph

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _oracle_checks(O, d):
    """what the oracle must reproduce of a dump: to_proba_edges, embedded scales, the cross entropy of both embeddings, the box"""
    m = d["meta"]
    rc, proba, scale = O.to_proba_edges(d["indptr"], d["nbr"], d["dist"], float(m["scale_rho"]), float(m["beta"]))
    assert rc == 0
    assert np.array_equal(scale, d["scale"]), "node scales (kdumap.rs:149-159) differ from the reference's"
    assert np.allclose(proba, d["proba"], rtol=2e-6, atol=0), "edge probabilities (kdumap.rs:172-218) differ from the reference's"
    eo0 = O.EntropyOptim(d["indptr"], d["nbr"], d["proba"], d["scale"], d["y0"], b=float(m["b"]))
    assert np.array_equal(eo0.emb_scale, d["emb_scale"]), "embedded scales (embedder.rs:1356-1366) differ from the reference's"
    assert abs(eo0.ce() - d["ce"][0]) < 1e-9 * abs(d["ce"][0]), "cross entropy of the initial embedding (embedder.rs:1127-1163)"
    eo1 = O.EntropyOptim(d["indptr"], d["nbr"], d["proba"], d["scale"], d["y"], b=float(m["b"]))
    assert abs(eo1.ce() - d["ce"][1]) < 1e-9 * abs(d["ce"][1]), "cross entropy of the final embedding"
    y0 = d["y0"]
    assert np.abs(y0.mean(0)).max() < 1e-4 and abs(np.abs(y0).max() - 5.0) < 1e-4, "set_data_box (embedder.rs:1376-1408)"


def test_reader_and_oracle_checks_on_an_oracle_made_dump(tmp_path, oracle):
    """the format round trip and every oracle-side check, on a dump the oracle itself wrote (what a reference dump must satisfy)"""
    from annembed_amd import refdump
    O = oracle
    indptr, nbr, dist, _, _ = synthetic_graph(n=500, dim=20, k=10, seed=4, ncomp=3)
    rc, res = O.one_step_embed(indptr, nbr, dist, 10, O.EmbedderParams(nb_grad_batch=6))
    assert rc == 0
    rc, proba, scale = O.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    eo0 = O.EntropyOptim(indptr, nbr, proba, scale, res["y0"])
    eo1 = O.EntropyOptim(indptr, nbr, proba, scale, res["y"])
    refdump.write(str(tmp_path), indptr, nbr, dist, proba, scale, res["y0"], res["y"], eo0.emb_scale, np.array([eo0.ce(), eo1.ce()]),
                  {"n": 500, "knbn": 10, "asked_dim": 2, "nb_grad_batch": 6, "scale_rho": 1.0, "beta": 1.0, "b": 1.0, "grad_step": 2.0, "nb_sampling_by_edge": 10})
    assert refdump.available(str(tmp_path)) and not refdump.available(str(tmp_path / "nowhere"))
    d = refdump.read(str(tmp_path))
    assert np.array_equal(d["nbr"], nbr) and np.array_equal(d["y0"], res["y0"]) and d["meta"]["knbn"] == 10
    _oracle_checks(O, d)
    os.remove(tmp_path / "ref_scale.f32")
    assert not refdump.available(str(tmp_path))


def test_oracle_against_a_real_annembed_dump(oracle):
    from annembed_amd import refdump
    if not refdump.available(GOLDEN):
        pytest.skip("no reference dump in tests/golden (INTEGRATION.md section 8: needs cargo)")
    _oracle_checks(oracle, refdump.read(GOLDEN))


@pytest.mark.gpu
def test_hip_against_a_real_annembed_dump(oracle):
    """the same stages on the device, plus the SGD loop's outcome against the reference's own (unseeded, threaded) run"""
    from annembed_amd import refdump
    if not refdump.available(GOLDEN):
        pytest.skip("no reference dump in tests/golden (INTEGRATION.md section 8: needs cargo)")
    import annembed_amd as A
    d = refdump.read(GOLDEN)
    m = d["meta"]
    g = A.KGraph(d["indptr"], d["nbr"], d["dist"])
    proba, scale = A.to_proba_edges(g, float(m["scale_rho"]), float(m["beta"])).get()
    assert np.array_equal(scale, d["scale"]) and np.allclose(proba, d["proba"], rtol=2e-6, atol=0)
    npar = A.NodeParams.from_host(g, d["proba"], d["scale"])
    nb = int(m["nb_grad_batch"])
    par = A.EmbedderParams(asked_dim=int(m["asked_dim"]), nb_grad_batch=nb, ce_mode=A.AE_CE_SEQUENTIAL, b=float(m["b"]), grad_step=float(m["grad_step"]),
                           nb_sampling_by_edge=int(m["nb_sampling_by_edge"]))
    eo = A.EntropyOptim(g, npar, par, d["y0"])
    assert np.array_equal(eo.get_embedded_scales(), d["emb_scale"])
    assert abs(eo.ce_compute_threaded() - d["ce"][0]) < 1e-9 * abs(d["ce"][0])
    e1 = A.EntropyOptim(g, npar, par, d["y"])
    assert abs(e1.ce_compute_threaded() - d["ce"][1]) < 1e-9 * abs(d["ce"][1])
    # the loop: ours from the reference's start, same schedule, against the reference's own outcome (ONE run of an unseeded 16-thread loop)
    for mode in (A.AE_CE_SEQUENTIAL, A.AE_CE_AUTO):
        par.ce_mode = mode
        _, ce0, ce1 = A.entropy_optimize(g, npar, par, d["y0"])
        print("reference dump: final CE %.6e (reference) vs %.6e (mode %d); initial %.6e" % (d["ce"][1], ce1, mode, ce0))
        assert abs(ce1 - d["ce"][1]) < 0.05 * abs(d["ce"][1])
    # the initialisation: the column space of our dmap start against the reference's
    y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(int(m["asked_dim"]), 5.0, 12)).embed_from_kgraph(g), 10.0)
    qa, _ = np.linalg.qr(y0.astype(np.float64))
    qb, _ = np.linalg.qr(d["y0"].astype(np.float64))
    cosines = np.linalg.svd(qa.T @ qb, compute_uv=False)
    print("reference dump: principal cosines of the two initialisations", cosines)
    assert cosines.min() > np.cos(0.1)
