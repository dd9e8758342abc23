"""Top rows of the rocprofv3 --kernel-trace --stats tables under a directory.  usage: python tools/kstats.py <dir> [rows] [out.txt]"""
import csv
import glob
import sys

d = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 16
out = open(sys.argv[3], "w") if len(sys.argv) > 3 else None
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:rows]:
        line = "%-110s calls %7s total_ms %10.3f avg_us %10.2f  %6s %%" % (r["Name"][:110], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"])
        print(line)
        if out:
            out.write(line + "\n")
