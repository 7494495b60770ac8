#!/bin/bash
# Counters of the cfg3 forward at one precision (tools/f16_speed.py): kernel stats, L2-fabric traffic, L2 hit rate, SQ busy.
#   bash tools/run_profile_f16.sh <tag> <precision>      (GPU box, repo root; outputs under gpurun_out/<tag>/)
TAG=${1:-r2f}; PREC=${2:-split}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O/summary
cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- python3 tools/f16_speed.py $PREC > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/summary/kernel_stats.csv
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --output-format csv --pmc $C --kernel-trace -d $O/pmc_$N -- python3 tools/f16_speed.py $PREC 32 > $O/pmc_$N.log 2>&1
  python3 tools/pmc_table.py $O/pmc_$N conv3x3 > $O/summary/pmc_$N.txt 2>&1
done
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
cat $O/summary/pmc_*.txt | head -80
