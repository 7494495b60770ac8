#!/bin/bash
# SQ counters of the conv kernels in a plain cfg3 forward (F(4x4,3x3) + F(2x2,3x3) kernels): bash tools/run_pmc_w4.sh <tag>
TAG=${1:-w4pmc}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- python3 tools/w2_fwd.py 5 > $O/trace.log 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --kernel-trace -d $O/pmc_SQ -- python3 tools/w2_fwd.py 3 > $O/sq.log 2>&1
python3 tools/pmc_table.py $O/pmc_SQ conv3x3 > $O/sq_conv.txt 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE --kernel-trace -d $O/pmc_SQ2 -- python3 tools/w2_fwd.py 3 > $O/sq2.log 2>&1
python3 tools/pmc_table.py $O/pmc_SQ2 conv3x3 > $O/sq2_conv.txt 2>&1
find $O -name "*.db" -delete
find $O -name "*counter_collection.csv" -size +4M -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
