#!/bin/bash
# ConvNeXt-tiny inference forward kernel trace  (bash tools/run_cnx_infer_profile.sh <tag> [B])
TAG=${1:-r6_cnx_infer}
B=${2:-64}
cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- python3 tools/convnext_bench.py $B 384 > $O/per_op.txt 2>&1
cp $(find $O/trace -name "*kernel_stats.csv" | head -1) $O/infer_kernel_stats.csv
find $O -name "*.db" -delete
find $O -name "*kernel_trace.csv" -size +1M -delete
