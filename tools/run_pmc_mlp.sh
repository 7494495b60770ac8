#!/bin/bash
# SQ counters per dispatch of cnblock_mlp_kernel in the cfg4 inference forward:  bash tools/run_pmc_mlp.sh <tag> [filter] [last N]
TAG=${1:-r6_mlp}; FLT=${2:-cnblock_mlp}; N=${3:-4}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O/summary
cd $GRAFT_REPO_ROOT
for C in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA"; do
  N2=$(echo $C | tr ' ' '_' | cut -c1-40)
  rocprofv3 --output-format csv --pmc $C --kernel-trace -d $O/pmc_$N2 -- python3 tools/convnext_bench.py 64 384 > $O/pmc_$N2.log 2>&1
  python3 tools/pmc_dispatches.py $O/pmc_$N2 $FLT $N > $O/summary/pmc_$N2.txt 2>&1
done
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -delete
find $O -name "*counter_collection.csv" -delete
cat $O/summary/*.txt
