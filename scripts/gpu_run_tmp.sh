cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
PMC_OUT=pmc_c5 PMC_CMD="scripts/gpu_c5.py fp16_mfma 30" bash scripts/gpu_pmc.sh "SQ_INSTS_VALU_MFMA_MOPS_F16" "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_BUSY_CU_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_MFMA SQ_INSTS_VALU" "GRBM_GUI_ACTIVE"
O=$GRAFT_REPO_ROOT/gpurun_out/r2n; rm -rf $O; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --resident --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_stdout.log 2>&1; echo "prof rc $?"
for f in $(find $O/prof -name "*kernel_stats.csv"); do cut -c1-150 $f | head -4; done
find $O -name "*kernel_trace.csv" -delete
