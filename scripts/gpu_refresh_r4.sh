# final refresh of the judged artefacts: GPU suite, driver command plain + under rocprofv3, PMC passes, default bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/refresh
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 $OUT/pytest_gpu.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_cmd.json 2>/dev/null; cut -c150-330 $OUT/bench_driver_cmd.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver -o bench -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_cmd_rocprof.json 2> $OUT/bench_driver_cmd_rocprof.err; echo "rocprof rc $?"
grep -E "k_wav_to_mel|k_minmax" $OUT/driver/*kernel_stats.csv | cut -c1-160
find $OUT/driver -name "*kernel_trace.csv" -delete
PMC_OUT=refresh/pmc bash scripts/gpu_pmc.sh "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" > $OUT/pmc.log 2>&1; tail -2 $OUT/pmc.log
timeout -k 10 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "full bench rc $?"; cut -c150-330 $OUT/bench_default.json
