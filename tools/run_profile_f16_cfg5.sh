#!/bin/bash
# Evidence set of the plain-fp16 forward at BASELINE cfg5's shape (768 x 768, 16 frames; tools/f16_rows_check.py): rocprofv3 kernel stats, per-launch HBM traffic
# (FETCH_SIZE / WRITE_SIZE passes), SQ counters per launch, per-op HIP-event table of the three routings.   bash tools/run_profile_f16_cfg5.sh <tag>
TAG=${1:-r6_f16}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O/summary
cd $GRAFT_REPO_ROOT
python3 tools/f16_rows_check.py 16 768 time 1 > $O/summary/per_op.txt 2>&1
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- python3 tools/f16_rows_check.py 16 768 only1 1 > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/summary/kernel_stats.csv
for C in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --output-format csv --pmc $C --kernel-trace -d $O/pmc_$N -- python3 tools/f16_rows_check.py 16 768 pmc 1 > $O/pmc_$N.log 2>&1
done
python3 tools/pmc_dispatches.py $O/pmc_GRBM_GUI_ACTIVE "" -1 > $O/summary/sq_counters.txt 2>&1
python3 tools/summarize_f16_traffic.py $O 18 > $O/summary/traffic.json 2> $O/summary/traffic.err
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
ls -la $O/summary; head -c 600 $O/summary/traffic.json
