"""CPU tests: the C-ABI library loads and exports every symbol include/annembed_hip.h declares; parameter
PODs through the ABI; loud failure without a GPU; host-side helpers."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "annembed_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ae_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    from annembed_amd import build
    build.build(verbose=False)
    from annembed_amd import _lib
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert len(names) > 50
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/annembed_hip.h is not exported" % n


def test_bindings_cover_the_header(lib):
    from annembed_amd import _lib
    bound = set(_lib.SIGNATURES) | set(_lib.STRING_GETTERS)
    assert set(_declared_functions()) == bound


def test_version_string(lib):
    assert b"gfx950" in lib.ae_version()


def test_embedder_params_default_matches_reference():
    """EmbedderParams::default, src/embedparams.rs:107-132"""
    import annembed_amd as A
    p = A.EmbedderParams()
    assert (p.asked_dim, p.dmap_init, p.beta, p.b, p.scale_rho, p.grad_step) == (2, True, 1.0, 1.0, 1.0, 2.0)
    assert (p.nb_sampling_by_edge, p.nb_grad_batch, p.grad_factor, p.hierarchy_layer, p.hubness_weighting) == (10, 20, 4, 0, False)
    assert p.seed == 4664397 and p.ce_mode == A.AE_CE_AUTO and p.ce_sampler == A.AE_SAMPLER_ROWCDF
    assert p.ce_precision == A.AE_PRECISION_F64  # the reference's f64 scalars unless the caller opts out (embedder.rs:1207-1229)
    p.set_dim(5)
    p.set_nb_gradient_batch(7)
    assert p.c().asked_dim == 5 and p.c().nb_grad_batch == 7


def test_diffusion_params_clamps():
    """DiffusionParams::new / set_alfa / set_beta / set_epsil, src/diffmaps.rs:95-160"""
    import annembed_amd as A
    dp = A.DiffusionParams(2, 5.0, 12)
    assert (dp.get_alfa(), dp.get_beta(), dp.get_epsil(), dp.get_time(), dp.get_gnbn()) == (0.5, pytest.approx(-0.1), 2.0, 5.0, 12)
    dp.set_alfa(3.0)
    assert dp.get_alfa() == 1.0
    dp.set_alfa(-5.0)
    assert dp.get_alfa() == -2.0
    dp.set_beta(0.5)  # rejected: must stay in [-1.01, 0]
    assert dp.get_beta() == pytest.approx(-0.1)
    dp.set_beta(-0.5)
    assert dp.get_beta() == -0.5
    dp.set_epsil(10.0)
    assert dp.get_epsil() == 4.0
    dp.set_epsil(0.1)
    assert dp.get_epsil() == 0.5
    assert A.DiffusionParams(3).get_time() is None and A.DiffusionParams(3).get_gnbn() is None


def test_no_gpu_fails_loudly(lib):
    """the product has no CPU fallback: without a HIP device every compute entry point returns AE_ERR_NO_DEVICE"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import annembed_amd as A
    indptr = np.array([0, 1, 2], np.uint64)
    with pytest.raises(A.AnnembedError) as e:
        A.KGraph(indptr, np.array([1, 0], np.uint32), np.array([1, 1], np.float32))
    assert e.value.code == 2 and "no CPU fallback" in str(e.value)


def test_product_does_not_import_oracle():
    """the oracle is test infrastructure: nothing under annembed_amd/ may reference it"""
    pkg = os.path.join(ROOT, "annembed_amd")
    for dirpath, _, files in os.walk(pkg):
        if "build" in dirpath.split(os.sep)[-1:]:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "liboracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, f


def test_shard_ranges():
    from annembed_amd.dist import sample_offset, shard_range, shard_sizes
    for n, w in ((10, 3), (60000, 8), (7, 7), (11_000_000, 8)):
        r = [shard_range(n, w, k) for k in range(w)]
        assert r[0][0] == 0 and r[-1][1] == n
        assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
        assert max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1
    assert sample_offset(0) == 0 and sample_offset(5) == 5 << 24


def test_bench_roofline_arithmetic_and_permuted_lattice():
    """bench.py's own code (no GPU): the roofline object is algorithmic bytes per launch / average launch duration with
    B = 24 + 4 k + 36 d bytes per SGD sample; the scale graph is a k-regular ring lattice whose node ids are a
    random permutation of the ring positions (edges are NOT memory-local), identical for the same seed on every rank."""
    import bench
    run = dict(rounds=15, kernel_ms=0.75, ms_per_step=0.8, nb_sample=7_200_000, batches_timed=20, mode=0)
    r = bench.roofline_of(run, 12, 2)
    assert r["bytes_per_sample"] == 24 + 48 + 72 and r["launches_per_batch"] == 15
    assert abs(r["bytes_per_launch"] - 144 * 7_200_000 / 15) < 1e-6
    assert abs(r["achieved"] - r["bytes_per_launch"] / (r["launch_avg_ms"] * 1e-3) / 1e9) < 1e-9 * r["achieved"]
    assert abs(r["frac"] - r["achieved"] / 8000.0) < 1e-15 and r["bound"] == "hbm" and r["traffic"] is None
    n, k = 5000, 6
    ip, nb, ds = bench.lattice_graph(n, k, seed=7, permute=True)
    ip2, nb2, ds2 = bench.lattice_graph(n, k, seed=7, permute=True)
    assert np.array_equal(nb, nb2) and np.array_equal(ds, ds2) and len(nb) == n * k and ip[-1] == n * k
    rows = nb.reshape(n, k).astype(np.int64)
    assert (rows != np.arange(n)[:, None]).all() and all(len(set(r_)) == k for r_ in rows[:200])
    assert np.bincount(nb, minlength=n).min() == k and np.bincount(nb, minlength=n).max() == k  # a permuted REGULAR graph
    assert (np.diff(ds.reshape(n, k), axis=1) >= 0).all()
    gap = np.abs(rows - np.arange(n)[:, None])
    assert np.median(np.minimum(gap, n - gap)) > n / 10  # neighbours are far apart in memory ...
    _, nb0, _ = bench.lattice_graph(n, k, seed=7, permute=False)
    gap0 = np.abs(nb0.reshape(n, k).astype(np.int64) - np.arange(n)[:, None])
    assert np.minimum(gap0, n - gap0).max() <= 3  # ... which they are not on the plain ring


def test_oracle_knn_definition_matches_f64_bruteforce():
    """oracle.knn_bruteforce_l2 (the definition the device producer is held to) against an independent f64 brute force:
    same neighbours on well-separated data, distances within f32 rounding; ties resolved by index."""
    from oracle import oracle as O
    from tests.util import gaussian_mixture, knn_graph
    x, _ = gaussian_mixture(400, 12, 3, seed=9)
    ip, nb, ds = O.knn_bruteforce_l2(x, 6)
    ip2, nb2, ds2 = knn_graph(x, 6)
    assert np.array_equal(ip, ip2) and np.array_equal(nb, nb2) and np.allclose(ds, ds2, rtol=2e-6, atol=1e-6)
    xt = np.zeros((6, 2), np.float32)  # all points identical: every row lists the other indices in increasing order
    _, nbt, dst = O.knn_bruteforce_l2(xt, 3)
    assert np.array_equal(nbt.reshape(6, 3)[0], [1, 2, 3]) and np.array_equal(nbt.reshape(6, 3)[5], [0, 1, 2]) and not dst.any()


def test_bench_final_line_is_short_and_parses():
    """the driver keeps the tail of stdout: bench.py's LAST line must be < 4 KB and parse on its own (round 4's 40 KB line did not)"""
    import glob
    import json
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*", "bench_r*.json")))
    assert files
    seen = 0
    for f in files[-3:]:
        lines = open(f).read().strip().splitlines()
        detail = [t for t in lines if t.startswith("bench_details: ")]
        if detail:  # round 5 on: everything measured is the prefixed line, the compact line comes last
            assert len(lines[-1]) < 4096 and json.loads(lines[-1])["roofline"]["frac"] > 0
            text = detail[-1][len("bench_details: "):]
        else:
            text = lines[-1]
        full = json.loads(text)
        line = bench.compact_line(full)
        assert len(line) < 4096 and "\n" not in line
        back = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                  "roofline", "cpu_baseline"):
            assert k in back, k
        assert back["value"] == full["value"] and back["roofline"]["frac"] == full["roofline"]["frac"]
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(back["roofline"])
        if full.get("cpu_baseline"):
            assert {"value", "cores", "kind", "sample"} <= set(back["cpu_baseline"])
        seen += 1
    assert seen


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus N` without a launcher starts N rank processes before any GPU call; a failing rank fails the run.
    (Here there is no GPU: every rank must exit with bench.py's "needs a GPU" message, and the parent with a non-zero code.)"""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["AE_BENCH_SPAWN_TIMEOUT"] = "240"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--backend", "gloo"],
                       capture_output=True, text=True, env=env, timeout=300)
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
        assert "needs a GPU" in r.stderr
        assert "started with WORLD_SIZE" not in r.stderr   # both ranks got RANK / WORLD_SIZE from the parent
