"""Kernel durations and inter-kernel gaps of the LAST forward in a rocprofv3 --kernel-trace CSV (small-batch regime: are we bound by kernels or by gaps?).

    python tools/trace_gaps.py <dir with *kernel_trace.csv> [kernels per forward]
"""
import csv, glob, sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
last = rows[-n:]
busy = gap = 0
prev_end = None
for s, e, name in last:
    g = 0 if prev_end is None else s - prev_end
    print(f"{name[:70]:70s} {(e - s) / 1e3:8.2f} us   gap {g / 1e3:7.2f} us")
    busy += e - s
    gap += max(g, 0)
    prev_end = e
print(f"span {(last[-1][1] - last[0][0]) / 1e3:.1f} us: kernels {busy / 1e3:.1f} us, gaps {gap / 1e3:.1f} us")
