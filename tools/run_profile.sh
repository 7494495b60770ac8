#!/bin/bash
# Round evidence: bench line, rocprofv3 kernel stats of the same command, HBM traffic PMC passes.
#   bash tools/run_profile.sh <tag>      (on the GPU box, from the repo root; outputs under gpurun_out/<tag>/)
TAG=${1:-r1c}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
python3 bench.py --legs-file $O/bench_legs.json > $O/bench.log 2>&1
tail -1 $O/bench.log > $O/bench.json
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt-precisions --no-h2d-leg --no-extra-legs > $O/trace.log 2>&1
grep '^{"metric"' $O/trace.log | tail -1 > $O/bench_under_rocprof.json
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/bench_kernel_stats.csv
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/pmc_FETCH_SIZE -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-h2d-leg --no-extra-legs --no-graph > $O/pmc_fetch.log 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE --kernel-trace -d $O/pmc_WRITE_SIZE -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-h2d-leg --no-extra-legs --no-graph > $O/pmc_write.log 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc_SQ -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-alt-precisions --no-h2d-leg --no-extra-legs --no-graph > $O/pmc_sq.log 2>&1
mkdir -p $O/summary
python3 tools/summarize_pmc.py $O $TAG 5 > $O/summary/traffic.txt 2>&1
cp profiles/${TAG}_conv_traffic.json $O/conv_traffic.json 2>/dev/null
python3 tools/pmc_table.py $O/pmc_SQ conv3x3 > $O/summary/sq_conv.txt 2>&1
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*counter_collection.csv" -size +4M -delete
ls -la $O $O/summary
