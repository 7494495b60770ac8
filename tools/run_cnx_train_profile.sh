#!/bin/bash
# ConvNeXt-tiny training-step kernel trace  (bash tools/run_cnx_train_profile.sh <tag> [B])
TAG=${1:-r1u}
B=${2:-32}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- python3 tools/convnext_train_bench.py $B 384 > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
python3 tools/trace_launches.py $O/trace "" > $O/launch_classes.txt 2>&1
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc_SQ -- python3 tools/convnext_train_bench.py $B 384 > $O/pmc_sq.log 2>&1
python3 tools/pmc_table.py $O/pmc_SQ > $O/sq_all.txt 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/pmc_F -- python3 tools/convnext_train_bench.py $B 384 > $O/pmc_f.log 2>&1
python3 tools/pmc_table.py $O/pmc_F > $O/fetch_all.txt 2>&1
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*counter_collection.csv" -size +4M -delete
