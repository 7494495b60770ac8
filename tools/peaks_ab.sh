#!/bin/bash
# kernel times (rocprofv3) and event-timed calls of the peak kernels for each library build under tools/ab/: bash tools/peaks_ab.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONPATH=$GRAFT_REPO_ROOT
cp sleap_nn_amd/lib/libposehip.so /tmp/base.so
for f in tools/ab/lib_*.so; do
  cp $f sleap_nn_amd/lib/libposehip.so
  rm -rf /tmp/prof_ab
  echo "== $f"
  timeout -k 10 120 python3 tools/peaks_bench.py 2>&1 | grep "one-pass"
  timeout -k 10 120 python3 tools/peaks_bench.py 32 zeros 2>&1 | grep "one-pass" | sed "s/^/  zeros: /"
  timeout -k 10 120 python3 tools/peaks_bench.py 32 sparse 2>&1 | grep "one-pass" | sed "s/^/  sparse: /"
  timeout -k 10 120 rocprofv3 --output-format csv --kernel-trace --stats -d /tmp/prof_ab -- python3 tools/peaks_bench.py > /tmp/prof_ab.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("/tmp/prof_ab/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "onepass" in r["Name"] or "place" in r["Name"]:
        print("  ", r["Name"][:50], r["Calls"], "avg %.1f us" % (float(r["AverageNs"])/1e3), "min %.1f" % (float(r["MinNs"])/1e3))
PY
done
cp /tmp/base.so sleap_nn_amd/lib/libposehip.so
