"""Per-DISPATCH rocprofv3 --pmc counters of the kernels whose name contains a filter, in dispatch order (one forward = one run of rows):
   python tools/pmc_dispatches.py <dir> <name filter> [last N dispatches]
Derived columns: mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 / (GRBM_GUI_ACTIVE / 8) (share of the SIMD-cycles the matrix pipe is busy; 256 CUs x 4 SIMDs),
clock = GRBM_GUI_ACTIVE / 8 / duration when the kernel trace of the same run is present."""
import collections
import csv
import glob
import sys

d, flt = sys.argv[1], sys.argv[2]
last = int(sys.argv[3]) if len(sys.argv) > 3 else 0
per = collections.defaultdict(dict)
names = {}
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        did = int(r["Dispatch_Id"])
        per[did][r["Counter_Name"]] = per[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        names[did] = r["Kernel_Name"]
dur = {}
for f in glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
ids = [i for i in sorted(per) if flt in names[i]]
if last < 0:  # the last forward: from its last stem launch on
    stems = [i for i in sorted(per) if "stem" in names[i]]
    ids = [i for i in ids if not stems or i >= stems[-1]]
elif last:
    ids = ids[-last:]
cols = sorted({c for i in ids for c in per[i]})
print("dispatch  us       " + " ".join(f"{c[-18:]:>18s}" for c in cols) + "   mfma_busy  clock_GHz  name")
for i in ids:
    c = per[i]
    g = c.get("GRBM_GUI_ACTIVE", 0.0) / 8
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024 / g if g else 0.0
    clk = g / dur[i] / 1e3 if i in dur and dur[i] > 0 else 0.0
    nm = names[i].split("(")[0].replace("void ", "").replace("ph::", "")
    print(f"{i:8d} {dur.get(i, 0):8.1f} " + " ".join(f"{c.get(k, 0):18.0f}" for k in cols) + f"   {busy:9.3f}  {clk:9.2f}  {nm[:60]}")
