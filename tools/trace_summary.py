"""Per kernel and grid: launches and average duration from a rocprofv3 --kernel-trace csv directory.
    python tools/trace_summary.py DIR [substring]"""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
want = sys.argv[2] if len(sys.argv) > 2 else "fpsg"
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"]
    if want not in n:
        continue
    m = re.search(r"::\(anonymous namespace\)::([A-Za-z0-9_]+(<[^>]*>)?)", n)
    key = (m.group(1) if m else n[:60], int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items()):
    print(f"{k[0]:44s} grid=({k[1]},{k[2]}) n={len(v):4d} avg={sum(v) / len(v):8.1f} us")
