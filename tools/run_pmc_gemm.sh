cd /tmp && export TMPDIR=/tmp
export PYTHONPATH=$GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/pmc_gemm
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
rocprofv3 --output-format csv --kernel-trace --stats -d $O/trace -- python3 tools/gemm_bench.py 64 0 "s2 lin 1536" > $O/trace.log 2>&1
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d $O/pmc1 -- python3 tools/gemm_bench.py 64 0 "s2 lin 1536" > $O/pmc1.log 2>&1
rocprofv3 --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_RD TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/pmc2 -- python3 tools/gemm_bench.py 64 0 "s2 lin 1536" > $O/pmc2.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE --kernel-trace -d $O/pmc3 -- python3 tools/gemm_bench.py 64 0 "s2 lin 1536" > $O/pmc3.log 2>&1
python3 tools/pmc_table.py $O gemm > $O/summary.txt 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs head -5 >> $O/summary.txt
tail -3 $O/pmc1.log >> $O/summary.txt; tail -3 $O/pmc2.log >> $O/summary.txt
find $O -name "*.db" -delete; find $O -size +2M -delete
cat $O/summary.txt
