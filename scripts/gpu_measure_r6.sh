# measurement set for profiles/r6: smoke, driver command (plain and under rocprofv3), PMC passes, full bench, c3 / c4 per-step kernel
# statistics with the exact-fp32 and the split-bf16 convolutions, the forced-process-group bench line
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/measure6
rm -rf $OUT; mkdir -p $OUT
# a step killed at its time limit ends the call: no further GPU step after a hang
guard() { if [ "$1" -ge 124 ]; then echo "a step was killed at its limit (rc $1): stopping"; exit $1; fi; }
python3 -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $OUT/smoke.log | cut -c1-300
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-extras > $OUT/bench_driver_cmd.json 2>/dev/null; guard $?; cut -c1-200 $OUT/bench_driver_cmd.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver -o bench -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_cmd_rocprof.json 2> $OUT/bench_driver_cmd_rocprof.err; rc=$?; echo "rocprof rc $rc"; guard $rc
grep -E "k_wav_to_mel|k_minmax" $OUT/driver/*kernel_stats.csv | cut -c1-160
find $OUT/driver -name "*kernel_trace.csv" -delete
PMC_OUT=measure6/pmc bash scripts/gpu_pmc.sh "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" > $OUT/pmc.log 2>&1; guard $?; tail -3 $OUT/pmc.log
timeout -k 10 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; rc=$?; echo "full bench rc $rc"; guard $rc; cut -c1-300 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
for split in 0 1; do
  for mode in engine graph; do
    IRIS_WINO_SPLIT_BF16=$split timeout -k 10 300 python3 scripts/gpu_fwdprof.py 20 $mode > $OUT/c3_${mode}_split$split.pre.log 2>&1; guard $?; tail -1 $OUT/c3_${mode}_split$split.pre.log
    IRIS_WINO_SPLIT_BF16=$split timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_${mode}_$split -o c3 -- python3 scripts/gpu_fwdprof.py 20 $mode > $OUT/c3_${mode}_split$split.log 2>&1; guard $?; grep "fwd\[" $OUT/c3_${mode}_split$split.log
    python3 scripts/trace_steps.py $OUT/c3_${mode}_$split/c3_kernel_trace.csv k_wav_to_mel 12 $OUT/c3_${mode}_split${split}_step_kernel_stats.csv
    find $OUT/c3_${mode}_$split -name "*kernel_trace.csv" -delete
  done
  IRIS_WINO_SPLIT_BF16=$split timeout -k 10 300 python3 scripts/gpu_c4prof.py 10 > $OUT/c4_split$split.pre.log 2>&1; guard $?; tail -1 $OUT/c4_split$split.pre.log
  IRIS_WINO_SPLIT_BF16=$split timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4_$split -o c4 -- python3 scripts/gpu_c4prof.py 10 > $OUT/c4_split$split.log 2>&1; guard $?; grep "train step" $OUT/c4_split$split.log
  python3 scripts/trace_steps.py $OUT/c4_$split/c4_kernel_trace.csv k_wav_to_mel 6 $OUT/c4_split${split}_step_kernel_stats.csv
  find $OUT/c4_$split -name "*kernel_trace.csv" -delete
done
IRIS_FORCE_PG=1 timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extra-steps 10 --no-cpu-baseline > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err; rc=$?; echo "rccl bench rc $rc"; guard $rc
du -sh $OUT
