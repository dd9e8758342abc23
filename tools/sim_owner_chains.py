"""What would the ordered dataflow's critical path be if consecutive samples on a busy node were handed on INSIDE a wave (as the time-sliced
mode's chains through a target's row are) instead of through memory?  CPU simulation on the C2-shaped graph: every sample finishes at
max(finish of the previous sample on each end point + the price of that hand-off) -- c_slow through memory (2.58 us measured), c_fast
lane to lane -- with samples grouped, inside windows of W resident lanes, by an OWNER end point into chains cut every 64 samples.
usage: python tools/sim_owner_chains.py [n] [k]"""
import ctypes
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.argv, argv = ["bench.py"], sys.argv
import bench  # noqa: E402
from oracle import oracle as O  # noqa: E402

CODE = r'''
#include <stdint.h>
#include <stdlib.h>
// owner_rule: 0 none (every hand-off slow), 1 the busier end point, 2 always the target
double sim(long S, long n, const long* si, const long* sj, const long* cnt, long W, int owner_rule, double cf, double cs, long* n_fast) {
    double* last_t = calloc(n, sizeof(double));
    long* last_owner = malloc(n * sizeof(long));
    long* last_win = malloc(n * sizeof(long));
    long* chain = calloc(n, sizeof(long));
    for (long v = 0; v < n; v++) { last_owner[v] = -1; last_win[v] = -1; }
    double mx = 0.; long nf = 0;
    for (long s = 0; s < S; s++) {
        long i = si[s], j = sj[s], win = s / W;
        long u = j, v = i;
        if (owner_rule == 1 && cnt[i] > cnt[j]) { u = i; v = j; }
        int fast = 0;
        if (owner_rule) {
            if (last_win[u] != win) chain[u] = 0;
            fast = last_owner[u] == u && last_win[u] == win && (chain[u] % 64) != 0;
        }
        double t = last_t[u] + (fast ? cf : cs), t2 = last_t[v] + cs;
        if (t2 > t) t = t2;
        last_t[u] = last_t[v] = t;
        last_owner[u] = u; last_owner[v] = u;
        last_win[u] = win; last_win[v] = win;
        if (owner_rule) chain[u]++;
        nf += fast;
        if (t > mx) mx = t;
    }
    *n_fast = nf;
    free(last_t); free(last_owner); free(last_win); free(chain);
    return mx;
}
'''


def main():
    n = int(argv[1]) if len(argv) > 1 else 60000
    k = int(argv[2]) if len(argv) > 2 else 12
    x = bench.synth_points(n, 784, seed=1, device="cpu")
    nb, ds = bench.knn_rows(x, 0, n, k)
    nbr = nb.numpy().astype(np.uint32).reshape(-1)
    dist = ds.numpy().reshape(-1)
    indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
    rc, proba, scale = O.to_proba_edges(indptr, nbr, dist, 1.0, 1.0)
    rng = np.random.default_rng(0)
    S = 10 * len(nbr)
    e = rng.choice(len(nbr), size=S, p=proba / proba.sum())
    si = np.ascontiguousarray(np.repeat(np.arange(n), k)[e], np.int64)
    sj = np.ascontiguousarray(nbr[e], np.int64)
    cnt = np.ascontiguousarray(np.bincount(si, minlength=n) + np.bincount(sj, minlength=n), np.int64)
    print("graph %d x k %d: %d samples, busiest node %d samples, mean %.0f" % (n, k, S, cnt.max(), 2 * S / n))
    d = tempfile.mkdtemp()
    open(os.path.join(d, "s.c"), "w").write(CODE)
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", os.path.join(d, "s.so"), os.path.join(d, "s.c")])
    lib = ctypes.CDLL(os.path.join(d, "s.so"))
    lib.sim.restype = ctypes.c_double
    for rule, name in ((0, "every hand-off through memory (today)"), (1, "owner = the busier end point"), (2, "owner = the target")):
        for W in (1 << 14, 1 << 15, 1 << 17, S):
            for cf in (0.1, 0.3):
                nf = ctypes.c_long()
                t = lib.sim(ctypes.c_long(S), ctypes.c_long(n), si.ctypes.data_as(ctypes.c_void_p), sj.ctypes.data_as(ctypes.c_void_p), cnt.ctypes.data_as(ctypes.c_void_p),
                            ctypes.c_long(W), ctypes.c_int(rule), ctypes.c_double(cf), ctypes.c_double(2.58), ctypes.byref(nf))
                print("%-40s window %8d  c_fast %.1f us: critical path %7.2f ms, %4.1f %% of the hand-offs fast" % (name, W, cf, t * 1e-3, 100.0 * nf.value / S))
                if rule == 0:
                    break
            if rule == 0:
                break


if __name__ == "__main__":
    main()
