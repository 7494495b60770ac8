#!/bin/bash
# Training-step evidence: kernel trace + SQ counters of tools/train_bench.py  (bash tools/run_train_profile.sh <tag> [B])
TAG=${1:-r1t}
B=${2:-32}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- python3 tools/train_bench.py $B 1024 > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
python3 tools/trace_launches.py $O/trace wgrad > $O/wgrad_launches.txt 2>&1
python3 tools/trace_launches.py $O/trace conv3x3 > $O/conv_launches.txt 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc_SQ -- python3 tools/train_bench.py $B 1024 > $O/pmc_sq.log 2>&1
python3 tools/pmc_table.py $O/pmc_SQ wgrad > $O/sq_wgrad.txt 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT --kernel-trace -d $O/pmc_SQ2 -- python3 tools/train_bench.py $B 1024 > $O/pmc_sq2.log 2>&1
python3 tools/pmc_table.py $O/pmc_SQ2 wgrad > $O/sq2_wgrad.txt 2>&1
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
find $O -name "*counter_collection.csv" -size +4M -delete
tail -2 $O/trace.log; cat $O/wgrad_launches.txt; cat $O/sq_wgrad.txt; cat $O/sq2_wgrad.txt | head -40
