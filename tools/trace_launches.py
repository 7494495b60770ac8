"""Per-launch durations of one kernel from a rocprofv3 --kernel-trace csv, grouped by grid size:
   python tools/trace_launches.py <dir> <name filter> [launches per step]"""
import collections
import csv
import glob
import sys

d, flt = sys.argv[1], sys.argv[2]
per_step = int(sys.argv[3]) if len(sys.argv) > 3 else 0
rows = []
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]),
                         (int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])), r["Kernel_Name"].split("(")[0][-40:]))
rows.sort()
if per_step:
    rows = rows[-per_step:]
    for i, (_, dur, grid, name) in enumerate(rows):
        print(f"{i:3d} {name:40s} grid={grid} {dur / 1e3:9.1f} us")
    print(f"total {sum(r[1] for r in rows) / 1e6:.3f} ms over {len(rows)} launches")
else:
    agg = collections.defaultdict(list)
    for _, dur, grid, name in rows:
        agg[(name, grid)].append(dur)
    for (name, grid), v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{name:40s} grid={grid} n={len(v):4d} avg={sum(v) / len(v) / 1e3:9.1f} us total={sum(v) / 1e6:8.3f} ms")
