"""File formats of the path's callers: CSV wire format of src/tools/io.rs and the .kgraph
container.  CPU only (host logic)."""
import numpy as np
import pytest

from annembed_amd import io as aio


def test_format_5e_is_rust_lower_exp():
    # Rust `format!("{:.5e}", v)` for f32 v: mantissa with 5 decimals, bare exponent (src/tools/io.rs:37, :60)
    cases = {1234.5678: "1.23457e3", 0.0: "0.00000e0", -0.015625: "-1.56250e-2", 1.0: "1.00000e0", 9.999996: "1.00000e1",
             6.02214076e23: "6.02214e23", 1e-10: "1.00000e-10", -5.0: "-5.00000e0"}
    for v, s in cases.items():
        assert aio.format_5e(v) == s, (v, aio.format_5e(v), s)
    assert aio.format_5e(float("inf")) == "inf" and aio.format_5e(float("nan")) == "NaN"
    # the f32 value is what is printed (to_f32, :37): 0.1f32 = 0.100000001490116...
    assert aio.format_5e(np.float64(0.1)) == "1.00000e-1" and aio.format_5e(16777217.0) == "1.67772e7"


def test_write_csv_and_read_back(tmp_path):
    y = np.array([[1.5, -2.25], [1e-3, 1234.5678], [0.0, 3.0]], np.float32)
    p = tmp_path / "emb.csv"
    assert aio.write_csv_labeled_array2(p, [7, 3, 9], y) == 1
    assert p.read_text() == "7,1.50000e0,-2.25000e0\n3,1.00000e-3,1.23457e3\n9,0.00000e0,3.00000e0\n"
    p2 = tmp_path / "emb2.csv"
    aio.write_csv_array2(p2, y)
    assert p2.read_text() == "1.50000e0,-2.25000e0\n1.00000e-3,1.23457e3\n0.00000e0,3.00000e0\n"
    # reader: headers '#' / '%' skipped, FIRST record dropped (io.rs:170-186), constant field count enforced
    src = tmp_path / "data.csv"
    src.write_text("# a header\n% another\n1.0,2.0,3.0\n4.0,5.0,6.0\n7.5,8.5,9.5\n")
    assert aio.get_header_size(src) == 2
    x = aio.get_toembed_from_csv(src, ",", 1.0)
    assert x.dtype == np.float32 and np.array_equal(x, np.array([[4, 5, 6], [7.5, 8.5, 9.5]], np.float32))
    assert aio.get_toembed_from_csv(src, ",", 0.0).shape == (0, 3)  # xsi >= 0 always: nothing sampled
    bad = tmp_path / "bad.csv"
    bad.write_text("1,2,3\n4,5\n")
    with pytest.raises(ValueError, match="non constant number of fields"):
        aio.get_toembed_from_csv(bad)
    one = tmp_path / "one.csv"
    one.write_text("1;2;3\n4;5;6\n")
    with pytest.raises(ValueError, match="only one field"):
        aio.get_toembed_from_csv(one, ",")
    assert np.array_equal(aio.get_toembed_from_csv(one, ";"), np.array([[4, 5, 6]], np.float32))
    bad2 = tmp_path / "bad2.csv"
    bad2.write_text("1,2\n3,x\n")
    with pytest.raises(ValueError, match="error decoding"):
        aio.get_toembed_from_csv(bad2)


def test_kgraph_file_roundtrip_and_validation(tmp_path):
    rng = np.random.default_rng(0)
    n = 50
    deg = rng.integers(1, 7, n)
    indptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.uint64)
    nbr = rng.integers(0, n, int(indptr[-1])).astype(np.uint32)
    dist = rng.random(int(indptr[-1])).astype(np.float32)
    ids = rng.permutation(1000)[:n].astype(np.uint64)
    p = tmp_path / "g.kgraph"
    aio.write_kgraph(p, indptr, nbr, dist, data_ids=ids)
    d = aio.read_kgraph(p)
    assert np.array_equal(d["indptr"], indptr) and np.array_equal(d["nbr"], nbr) and np.array_equal(d["dist"], dist)
    assert np.array_equal(d["data_ids"], ids) and d["max_nbng"] == int(deg.max())
    raw = p.read_bytes()
    assert raw[:8] == b"AEKGRAPH" and len(raw) == 32 + 8 * (n + 1) + 8 * int(indptr[-1]) + 8 * n
    (tmp_path / "t.kgraph").write_bytes(raw[:-3])
    with pytest.raises(ValueError, match="truncated"):
        aio.read_kgraph(tmp_path / "t.kgraph")
    (tmp_path / "m.kgraph").write_bytes(b"XXKGRAPH" + raw[8:])
    with pytest.raises(ValueError, match="not a .kgraph"):
        aio.read_kgraph(tmp_path / "m.kgraph")
    (tmp_path / "x.kgraph").write_bytes(raw + b"\0")
    with pytest.raises(ValueError, match="trailing"):
        aio.read_kgraph(tmp_path / "x.kgraph")
    with pytest.raises(ValueError):
        aio.write_kgraph(tmp_path / "bad.kgraph", indptr, nbr[:-1], dist)


def test_embed_cli_flags_mirror_reference_defaults():
    """src/bin/embed.rs:224-321: flag names and defaults (batch 20, stepg 2., nbsample 10, layer 0, scale 1.0, dim 2, no
    quality; HnswParams::my_default :66-74 without the hnsw subcommand)."""
    from annembed_amd import embed_cli as E
    ns, h = E.parse(["--csv", "x.csv"])
    assert (ns.batch, ns.stepg, ns.nbsample, ns.hierarchy, ns.scale, ns.dimension, ns.quality, ns.outfile) == (20, 2.0, 10, 0, 1.0, 2, None, None)
    assert h == {"max_conn": 64, "ef_c": 512, "knbn": 10, "distance": "DistL2", "scale_modification": 1.0}
    ns, h = E.parse(["--csv", "x.csv", "-o", "y.csv", "--batch", "5", "--dim", "3", "--layer", "1", "-q", "0.5", "hnsw", "--dist", "DistCosine",
                     "--nbconn", "48", "--ef", "400", "--knbn", "6", "--scale_modify_f", "0.5"])
    assert (ns.outfile, ns.batch, ns.dimension, ns.hierarchy, ns.quality) == ("y.csv", 5, 3, 1, 0.5)
    assert h == {"max_conn": 48, "ef_c": 400, "knbn": 6, "distance": "DistCosine", "scale_modification": 0.5}
    with pytest.raises(SystemExit):
        E.parse(["--csv", "x.csv", "hnsw", "--dist", "DistFoo", "--nbconn", "4", "--ef", "4", "--knbn", "4"])
    with pytest.raises(SystemExit):
        E.parse(["--batch", "5"])  # --csv is required


def _write_idx(dirpath, prefix, n, seed):
    import struct
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(n, 28, 28), dtype=np.uint8)
    lab = rng.integers(0, 10, size=n, dtype=np.uint8)
    (dirpath / ("%s-images-idx3-ubyte" % prefix)).write_bytes(struct.pack(">IIII", 2051, n, 28, 28) + img.tobytes())
    (dirpath / ("%s-labels-idx1-ubyte" % prefix)).write_bytes(struct.pack(">II", 2049, n) + lab.tobytes())
    return img, lab


def test_mnist_idx_reader(tmp_path):
    """IDX files as src/utils/mnistio.rs:56-201 reads them: big-endian header, magic 2051 / 2049, 60000 or 10000 items of 28 x 28;
    the flattened matrix of examples/mnist_fashion.rs:43-66 is row-major pixels, train then test."""
    import io
    import struct
    itr, ltr = _write_idx(tmp_path, "train", 60000, 1)
    ite, lte = _write_idx(tmp_path, "t10k", 10000, 2)
    tr = aio.load_mnist_train_data(tmp_path)
    assert np.array_equal(tr.get_images(), itr) and np.array_equal(tr.get_labels(), ltr)
    x, lab = aio.mnist_images_as_vectors(tmp_path)
    assert x.shape == (70000, 784) and x.dtype == np.float32 and lab.shape == (70000,)
    # image k, row i, column j -> vector[k][28 i + j]  (the iteration order of images.slice(s![.., .., k]).iter())
    assert x[3, 28 * 5 + 7] == float(itr[3, 5, 7]) and x[60000 + 11, 28 * 27 + 1] == float(ite[11, 27, 1])
    assert np.array_equal(lab[:60000], ltr) and np.array_equal(lab[60000:], lte)
    assert aio.find_mnist_dir([str(tmp_path)]) == str(tmp_path) and aio.find_mnist_dir([str(tmp_path / "nowhere")]) is None
    # the reference's asserts (mnistio.rs:69, 79, 89, 99, 135, 145) as ValueErrors
    for head in (struct.pack(">IIII", 2049, 60000, 28, 28), struct.pack(">IIII", 2051, 1234, 28, 28), struct.pack(">IIII", 2051, 10000, 32, 28)):
        with pytest.raises(ValueError):
            aio.read_image_file(io.BytesIO(head + bytes(784)))
    with pytest.raises(ValueError, match="truncated"):
        aio.read_image_file(io.BytesIO(struct.pack(">IIII", 2051, 10000, 28, 28) + bytes(100)))
    with pytest.raises(ValueError):
        aio.read_label_file(io.BytesIO(struct.pack(">II", 2051, 10000) + bytes(10000)))
    with pytest.raises(ValueError):
        aio.read_label_file(io.BytesIO(struct.pack(">II", 2049, 10000) + bytes(9999)))
