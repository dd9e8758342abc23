"""Depth of the dependency DAG of one CE batch (CPU, no GPU needed): a sample depends on the previous sample that touched either
of its end points (attraction: embedder.rs:1228-1239 reads and writes y_i and y_j) -- and, for the bit-exact sequential mode, on
the previous writer of each of its five negatives as well.  The batch cannot finish in fewer dependent steps than this depth,
whatever the kernel: with a cross-CU hand-off of 2-4 us it bounds a faithful batch from below.
usage: python tools/dependency_depth.py [n] [k]   (bench.py's MNIST-shaped generator, exact kNN on the CPU)"""
import ctypes
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

CODE = r'''
#include <stdint.h>
long depth(long S, const long* nodes, int per, int writes, long* last, long* hist, long nh) {
    long mx = 0;
    for (long s = 0; s < S; s++) {
        const long* p = nodes + s * per;
        long d = 0;
        for (int t = 0; t < per; t++) if (last[p[t]] > d) d = last[p[t]];
        d += 1;
        for (int t = 0; t < writes; t++) last[p[t]] = d;
        if (d > mx) mx = d;
        if (d < nh) hist[d]++;
    }
    return mx;
}
'''


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    t0 = time.time()
    x = bench.synth_points(n, 784, seed=1, device="cpu")
    nb, ds = bench.knn_rows(x, 0, n, k)
    nbr = nb.numpy().astype(np.uint32).reshape(-1)
    dist = ds.numpy().reshape(-1)
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    rc, proba, scale = O.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    w_in = np.zeros(n)
    np.add.at(w_in, nbr.astype(np.int64), proba)
    print("graph: %d nodes, k = %d (%.0f s); largest in-weight %.1f, largest in-degree %d" % (n, k, time.time() - t0, w_in.max(), np.bincount(nbr, minlength=n).max()))
    rng = np.random.default_rng(0)
    S = 10 * len(nbr)
    e = rng.choice(len(nbr), size=S, p=proba / proba.sum())
    i = np.repeat(np.arange(n), k)[e]
    j = nbr[e].astype(np.int64)
    negs = rng.integers(0, n, size=(S, 5))
    d = tempfile.mkdtemp()
    open(os.path.join(d, "d.c"), "w").write(CODE)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", os.path.join(d, "d.so"), os.path.join(d, "d.c")])
    lib = ctypes.CDLL(os.path.join(d, "d.so"))
    lib.depth.restype = ctypes.c_long
    for name, nodes, per in (("end points only (event-ordered mode)", np.stack([i, j], 1), 2), ("end points + the 5 negatives (sequential mode)", np.concatenate([np.stack([i, j], 1), negs], 1), 7)):
        nodes = np.ascontiguousarray(nodes, np.int64)
        last = np.zeros(n, np.int64)
        hist = np.zeros(1 << 20, np.int64)
        mx = lib.depth(ctypes.c_long(S), nodes.ctypes.data_as(ctypes.c_void_p), ctypes.c_int(per), ctypes.c_int(2), last.ctypes.data_as(ctypes.c_void_p),
                       hist.ctypes.data_as(ctypes.c_void_p), ctypes.c_long(len(hist)))
        cum = np.cumsum(hist[1:mx + 1])
        print("%-52s samples %d  depth %d  (half of the samples lie deeper than level %d); busiest node: %d samples, mean %.0f" % (
            name, S, mx, int(np.searchsorted(cum, 0.5 * S)) + 1, int((np.bincount(i, minlength=n) + np.bincount(j, minlength=n)).max()), 2 * S / n))


if __name__ == "__main__":
    main()
