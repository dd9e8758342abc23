"""Selected scale shapes of bench.py on their own (development / profiling): python tools/run_scale_shapes.py c4_knn,c4,c5_knn,c5,c3_knn,c3 [steps]
Prints one JSON object per shape (the entries bench.py puts under "scale_shapes")."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
which = sys.argv[1].split(",") if len(sys.argv) > 1 else ["c4_knn"]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sys.argv = ["bench.py"]
import bench  # noqa: E402
import annembed_amd as A  # noqa: E402
from annembed_amd import _lib as L  # noqa: E402

seq = os.environ.get("WITH_SEQUENTIAL") == "1"
for w in which:
    if w == "c4_knn":
        r = bench.scale_shape(A, L, w, 11_000_000, 6, 8, steps, seq, hub_weighting=os.environ.get("NO_HUBW") != "1", graph=bench.config_graphs(A, "c4"))
    elif w == "c4":
        r = bench.scale_shape(A, L, w, 11_000_000, 6, 8, steps, seq)
    elif w == "c5_knn":
        r = bench.scale_shape(A, L, w, 6_250_000, 10, 16, steps, False, graph=bench.config_graphs(A, "c5"))
    elif w == "c5":
        r = bench.scale_shape(A, L, w, 6_250_000, 10, 16, steps, False)
    elif w == "c3_knn":
        r = bench.scale_shape(A, L, w, 1_650_000, 6, 2, steps, seq, hub_weighting=True, dmap_start=True,
                              graph=bench.exact_knn_graph(A, bench.higgs_shaped_points(1_650_000), 6, "Higgs-shaped points"))
    elif w == "c3":
        r = bench.scale_shape(A, L, w, 1_650_000, 6, 2, steps, seq)
    else:
        raise SystemExit("unknown shape " + w)
    brief = {k: (v if not isinstance(v, dict) else {kk: vv for kk, vv in v.items() if kk in ("ms_per_step", "ce_after", "dtype", "ce_mode")} | (
        {"frac_whole_batch": v["roofline"]["frac_whole_batch"], "sliced": v["roofline"].get("sliced")} if "roofline" in v else {})) for k, v in r.items()}
    print("SHAPE", w, json.dumps(brief), flush=True)
    print("FULL", w, json.dumps(r), flush=True)
