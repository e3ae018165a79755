cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3e; rm -rf $O; mkdir -p $O
bash scripts/gpu_pmc.sh "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" > $O/pmc.log 2>&1; tail -2 $O/pmc.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_stdout.log 2>&1; echo "prof rc $?"
for f in $(find $O/prof -name "*kernel_stats.csv"); do cut -c1-150 $f | head -3; done
find $O -name "*kernel_trace.csv" -delete
