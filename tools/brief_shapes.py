"""prints the SHAPE lines of tools/run_scale_shapes.py logs compactly: python tools/brief_shapes.py file..."""
import json
import sys
for f in sys.argv[1:]:
    for line in open(f):
        if line.startswith("SHAPE"):
            w = line.split(" ", 2)
            d = json.loads(w[2])
            print(f, w[1], "max in-degree", d.get("max_in_degree"))
            for k, v in d.items():
                if isinstance(v, dict) and "ms_per_step" in v:
                    sl = v.get("sliced") or {}
                    print("   %-24s %8.1f ms  frac %.3f  ce_after %.5g  %s" % (k, v["ms_per_step"], v.get("frac_whole_batch", 0), v.get("ce_after", 0),
                          ("classes %s overflow %.4f" % (sl.get("classes"), sl.get("overflow_mass_fraction", 0))) if sl else ""))
        elif "colouring" in line or "rror" in line or line.startswith("rc "):
            print(f, line.rstrip()[:300])
