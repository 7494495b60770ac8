#!/bin/bash
# extra memory-path counters of the fp16-pipe conv kernels: bash tools/pmc_f16_extra.sh <tag> <precision>
TAG=${1:-r2x}; PREC=${2:-split}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O/summary
cd $GRAFT_REPO_ROOT
for C in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_sum" "TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_sum"; do
  N=$(echo $C | cut -d' ' -f1)
  rocprofv3 --output-format csv --pmc $C --kernel-trace -d $O/pmc_$N -- python3 tools/f16_speed.py $PREC 32 > $O/pmc_$N.log 2>&1
  python3 tools/pmc_table.py $O/pmc_$N conv3x3_f16_persist_kernel\<64 > $O/summary/pmc_$N.txt 2>&1
done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
cat $O/summary/pmc_*.txt
