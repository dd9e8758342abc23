"""AE_CE_ORDERED / AE_CE_SEQUENTIAL on the C2 workload under the launch knobs of launch_dataflow (in-process sweep).
usage: python tools/run_ordered_sweep.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["AE_DEBUG_KNOBS"] = "1"
import annembed_amd as A  # noqa: E402
sys.argv = ["bench.py"]
import bench  # noqa: E402

n, k = 60000, int(os.environ.get("K", "12"))
x = bench.synth_points(n, 784, seed=1)
nb_t, ds_t = bench.knn_rows(x, 0, n, k)
indptr = np.arange(n + 1, dtype=np.uint64) * np.uint64(k)
g = A.KGraph(indptr, nb_t.cpu().numpy().astype(np.uint32).reshape(-1), ds_t.cpu().numpy().reshape(-1), k)
npar = A.to_proba_edges(g, 1.0, 1.0)
y0 = A.set_data_box(A.DiffusionMaps(A.DiffusionParams(2, 5.0, 12)).embed_from_kgraph(g), 10.0)
nb = 6
for mode, name in ((A.AE_CE_ORDERED, "ordered"), (A.AE_CE_SEQUENTIAL, "sequential")):
    for block in (64, 128, 256):
        for stride in (1, 2, 4, 8, 16):
            for grid in (0, 2048) if mode == A.AE_CE_ORDERED else (0,):
                os.environ["AE_DF_BLOCK"] = str(block)
                os.environ["AE_DF_LANE_STRIDE"] = str(stride)
                os.environ.pop("AE_DF_GRID", None)
                if grid:
                    os.environ["AE_DF_GRID"] = str(grid)
                eo = A.EntropyOptim(g, npar, A.EmbedderParams(nb_grad_batch=25, ce_mode=mode), y0)
                S = 10 * eo.get_nb_edges()
                eo.gradient_iteration_threaded(S, 1.0 - 1 / 25, 1)
                eo.dataflow_time()
                t0 = time.perf_counter()
                for it in range(2, nb + 1):
                    eo.gradient_iteration_threaded(S, 1.0 - it / 25, it)
                eo.get_embedded()
                dt = (time.perf_counter() - t0) / (nb - 1) * 1e3
                ms, cnt = eo.dataflow_time()
                print("%-10s block %3d stride %2d grid %4s: %.2f ms/batch, dataflow kernel %.2f ms" % (name, block, stride, grid or "auto", dt, ms), flush=True)
                del eo
