"""Per-kernel averages of rocprofv3 --pmc counters: python tools/pmc_table.py <dir> [name filter]"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    names = {}
    for r in csv.DictReader(open(f)):
        per[(int(r["Dispatch_Id"]), r["Counter_Name"])] += float(r["Counter_Value"])
        names[int(r["Dispatch_Id"])] = r["Kernel_Name"]
    for (did, c), v in per.items():
        k = names[did].split("(")[0].replace("void ", "")
        if flt in k:
            agg[k][c][0] += 1
            agg[k][c][1] += v
for k, cs in agg.items():
    print(k[:110])
    for c, (n, tot) in sorted(cs.items()):
        print(f"   {c:32s} n={n:4d} avg={tot / n:16.1f}")
