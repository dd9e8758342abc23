"""Host-side file formats of the path's callers.

* CSV wire formats of src/tools/io.rs: `write_csv_labeled_array2` (:23-45), `write_csv_array2` (:48-67) -- values
  printed with Rust's `{:.5e}` of the f32 value, records terminated by '\\n' -- and the reader `get_toembed_from_csv`
  (:115-226) with its header rule (`get_header_size` :70-110: leading lines starting with '#' or '%') and its quirk:
  the FIRST record only fixes the number of fields and is NOT returned (:170-186).
* `.kgraph`: a trivial binary CSR container for a KGraph, so that graphs produced by a real
  annembed / hnsw_rs run can be fed to this library (INTEGRATION.md shows the Rust writer):
      magic  8 bytes  b"AEKGRAPH"
      u32    version (1)        u32  max_nbng
      u64    n                  u64  nnz
      u64[n+1] indptr           u32[nnz] nbr            f32[nnz] dist           u64[n] data_ids
  little endian, rows sorted by increasing distance (src/fromhnsw/kgraph.rs:508-509).
* MNIST / Fashion-MNIST IDX files (src/utils/mnistio.rs:56-201): `read_image_file` (magic 2051, big-endian header, 60000 or 10000
  items of 28 x 28 bytes), `read_label_file` (magic 2049), `load_mnist_train_data` / `load_mnist_test_data` (the four file names),
  and `mnist_images_as_vectors`: the flattening of examples/mnist_fashion.rs:43-66 (row-major pixels as f32, train then test) --
  the hook that lets this build be held against the only results the reference publishes on this path (embedder.rs:585-618).
No device code here: this is the boundary's file plumbing.
"""
import os
import csv
import io as _io
import math
import struct

import numpy as np

KGRAPH_MAGIC = b"AEKGRAPH"


def format_e(x, prec):
    """Rust `{:.<prec>e}` of an f64: mantissa with `prec` decimals, bare exponent ("5.07e0", "1.36e-1")"""
    v = float(x)
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "inf" if v > 0 else "-inf"
    mant, exp = ("%.*e" % (prec, v)).split("e")
    return "%se%d" % (mant, int(exp))


def format_5e(x):
    """Rust `format!("{:.5e}", x as f32)`: 6 significant digits, exponent without sign padding ("1.23457e3", "-1.56250e-2")"""
    v = float(np.float32(x))
    if math.isnan(v):
        return "NaN"
    if math.isinf(v):
        return "inf" if v > 0 else "-inf"
    mant, exp = ("%.5e" % v).split("e")
    return "%se%d" % (mant, int(exp))


def write_csv_labeled_array2(path, labels, mat):
    """write_csv_labeled_array2, src/tools/io.rs:23-45: label, then the row"""
    mat = np.asarray(mat)
    if mat.ndim != 2 or len(labels) != mat.shape[0]:
        raise ValueError("labels and rows differ")
    with open(path, "w", newline="") as f:
        w = csv.writer(f, lineterminator="\n")
        for i in range(mat.shape[0]):
            w.writerow([str(labels[i])] + [format_5e(v) for v in mat[i]])
    return 1


def write_csv_array2(path, mat):
    """write_csv_array2, src/tools/io.rs:48-67"""
    mat = np.asarray(mat)
    with open(path, "w", newline="") as f:
        w = csv.writer(f, lineterminator="\n")
        for i in range(mat.shape[0]):
            w.writerow([format_5e(v) for v in mat[i]])
    return 1


def get_header_size(path):
    """get_header_size, src/tools/io.rs:70-110: number of leading lines beginning with '#' or '%'"""
    n = 0
    with open(path, "rb") as f:
        while True:
            c = f.read(1)
            if not c:
                raise EOFError("get_header_size: unexpected end of file")  # read_exact fails, :93
            if c in (b"#", b"%"):
                n += 1
                while True:
                    c = f.read(1)
                    if not c:
                        raise EOFError("get_header_size: unexpected end of file")
                    if c == b"\n":
                        break
            else:
                return n


def get_toembed_from_csv(path, delim=",", sampling_fraction=1.0, dtype=np.float32, rng=None):
    """get_toembed_from_csv, src/tools/io.rs:115-226 -> (rows, nb_fields) array.  The first record after the header
    only fixes the field count and is dropped (:170-186); every later record is kept with probability
    `sampling_fraction` (:189-192; the reference's thread RNG is unseeded, `rng` here)."""
    nb_header = get_header_size(path)
    rng = rng if rng is not None else np.random.default_rng()
    rows = []
    nb_fields = 0
    with open(path, "r", newline="") as f:
        for _ in range(nb_header):
            f.readline()
        for nb_record, record in enumerate(csv.reader(f, delimiter=delim)):
            if nb_record == 0:
                nb_fields = len(record)
                if nb_fields < 2:
                    raise ValueError("found only one field in record, check the delimitor , got %r as delimitor" % delim)
                continue
            if len(record) != nb_fields:
                raise ValueError("non constant number of fields at record %d first record has %d" % (nb_record + 1, nb_fields))
            if rng.random() >= sampling_fraction:
                continue
            try:
                rows.append([float(x) for x in record])
            except ValueError:
                raise ValueError("error decoding a field of record %d : %r" % (nb_record + 1, record))
    return np.asarray(rows, dtype).reshape(len(rows), nb_fields)


def write_kgraph(path, indptr, nbr, dist, max_nbng=None, data_ids=None):
    indptr = np.ascontiguousarray(indptr, "<u8")
    nbr = np.ascontiguousarray(nbr, "<u4")
    dist = np.ascontiguousarray(dist, "<f4")
    n, nnz = len(indptr) - 1, len(nbr)
    if len(dist) != nnz or int(indptr[-1]) != nnz or int(indptr[0]) != 0:
        raise ValueError("inconsistent CSR arrays")
    if max_nbng is None:
        max_nbng = int(np.diff(indptr.astype(np.int64)).max()) if n else 0
    data_ids = np.arange(n, dtype="<u8") if data_ids is None else np.ascontiguousarray(data_ids, "<u8")
    if len(data_ids) != n:
        raise ValueError("data_ids must have one entry per node")
    with open(path, "wb") as f:
        f.write(KGRAPH_MAGIC)
        f.write(struct.pack("<IIQQ", 1, int(max_nbng), n, nnz))
        for a in (indptr, nbr, dist, data_ids):
            f.write(a.tobytes())


def read_kgraph(path):
    """-> dict(indptr, nbr, dist, max_nbng, data_ids); validates sizes, leaves the graph invariants (sorted rows, no
    empty row, indices in range) to KGraph's device-side validation"""
    with open(path, "rb") as f:
        head = f.read(8 + 24)
        if len(head) != 32 or head[:8] != KGRAPH_MAGIC:
            raise ValueError("not a .kgraph file")
        version, max_nbng, n, nnz = struct.unpack("<IIQQ", head[8:])
        if version != 1:
            raise ValueError("unsupported .kgraph version %d" % version)

        def arr(dt, count):
            b = f.read(count * np.dtype(dt).itemsize)
            if len(b) != count * np.dtype(dt).itemsize:
                raise ValueError("truncated .kgraph file")
            return np.frombuffer(b, dt).copy()

        out = {"indptr": arr("<u8", n + 1), "nbr": arr("<u4", nnz), "dist": arr("<f4", nnz), "data_ids": arr("<u8", n), "max_nbng": max_nbng}
        if f.read(1):
            raise ValueError("trailing bytes in .kgraph file")
    if int(out["indptr"][0]) != 0 or int(out["indptr"][-1]) != nnz:
        raise ValueError("corrupt indptr")
    return out


# ------------------------------------------------------------------------------------------------------------------
# MNIST IDX files -- src/utils/mnistio.rs
# ------------------------------------------------------------------------------------------------------------------
IDX_IMAGE_MAGIC = 2051  # mnistio.rs:69
IDX_LABEL_MAGIC = 2049  # mnistio.rs:135


def read_image_file(f):
    """read_image_file, src/utils/mnistio.rs:56-121: four big-endian u32 (magic 2051, items, rows 28, columns 28), then items x 28 x 28
    bytes.  Returns uint8[items, 28, 28] (image k = out[k]; the reference stores Array3[[row, column, k]] -- same bytes, same order
    within an image).  The reference asserts the magic, items in {60000, 10000} and the 28 x 28 shape: ValueError here."""
    head = f.read(16)
    if len(head) != 16:
        raise ValueError("IDX image file: truncated header")
    magic, nbitem, nbrow, nbcolumn = struct.unpack(">IIII", head)
    if magic != IDX_IMAGE_MAGIC:
        raise ValueError("IDX image file: magic %d, expected %d (mnistio.rs:69)" % (magic, IDX_IMAGE_MAGIC))
    if nbitem not in (60000, 10000):
        raise ValueError("IDX image file: %d items, expected 60000 or 10000 (mnistio.rs:79)" % nbitem)
    if nbrow != 28 or nbcolumn != 28:
        raise ValueError("IDX image file: %d x %d images, expected 28 x 28 (mnistio.rs:89,99)" % (nbrow, nbcolumn))
    body = f.read(nbitem * 784)
    if len(body) != nbitem * 784:
        raise ValueError("IDX image file: truncated (read_exact fails, mnistio.rs:106)")
    return np.frombuffer(body, np.uint8).reshape(nbitem, 28, 28).copy()


def read_label_file(f):
    """read_label_file, src/utils/mnistio.rs:123-147: magic 2049, items, then one byte per item"""
    head = f.read(8)
    if len(head) != 8:
        raise ValueError("IDX label file: truncated header")
    magic, nbitem = struct.unpack(">II", head)
    if magic != IDX_LABEL_MAGIC:
        raise ValueError("IDX label file: magic %d, expected %d (mnistio.rs:135)" % (magic, IDX_LABEL_MAGIC))
    if nbitem not in (60000, 10000):
        raise ValueError("IDX label file: %d items, expected 60000 or 10000 (mnistio.rs:145)" % nbitem)
    body = f.read(nbitem)
    if len(body) != nbitem:
        raise ValueError("IDX label file: truncated")
    return np.frombuffer(body, np.uint8).copy()


class MnistData:
    """MnistData, src/utils/mnistio.rs:16-54"""

    def __init__(self, image_filename, label_filename):
        with open(image_filename, "rb") as f:
            self.images = read_image_file(f)
        with open(label_filename, "rb") as f:
            self.labels = read_label_file(f)
        if len(self.images) != len(self.labels):
            raise ValueError("images and labels differ in number")

    def get_labels(self):
        return self.labels

    def get_images(self):
        return self.images


def load_mnist_train_data(dname):
    """load_mnist_train_data, src/utils/mnistio.rs:150-165"""
    return MnistData(os.path.join(dname, "train-images-idx3-ubyte"), os.path.join(dname, "train-labels-idx1-ubyte"))


def load_mnist_test_data(dname):
    """load_mnist_test_data, src/utils/mnistio.rs:167-184"""
    return MnistData(os.path.join(dname, "t10k-images-idx3-ubyte"), os.path.join(dname, "t10k-labels-idx1-ubyte"))


def mnist_images_as_vectors(dname, with_test=True):
    """The data matrix of examples/mnist_fashion.rs:38-66 / mnist_digits.rs: every image as 784 f32 (row-major pixels), the 60000
    training images followed by the 10000 test images -> (float32[n, 784], uint8[n] labels)."""
    tr = load_mnist_train_data(dname)
    xs, ls = [tr.get_images().reshape(-1, 784)], [tr.get_labels()]
    if with_test:
        te = load_mnist_test_data(dname)
        xs.append(te.get_images().reshape(-1, 784))
        ls.append(te.get_labels())
    return np.ascontiguousarray(np.concatenate(xs).astype(np.float32)), np.concatenate(ls)


def find_mnist_dir(candidates=None):
    """first directory that holds the four IDX files (`data/mnist/*-idx3-ubyte`), or None"""
    names = ("train-images-idx3-ubyte", "train-labels-idx1-ubyte", "t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for d in candidates or (os.environ.get("AE_MNIST_DIR", ""), os.path.join(root, "data", "fashion-mnist"), os.path.join(root, "data", "mnist")):
        if d and all(os.path.exists(os.path.join(d, nm)) for nm in names):
            return d
    return None
