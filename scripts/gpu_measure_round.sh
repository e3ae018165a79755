# measurement set for profiles/r<N>: driver command (plain and under rocprofv3), PMC passes, shapes, full bench, c3 / c4 per-step
# kernel statistics, RCCL at world size 1 (bench line + kernel trace)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/measure
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-extras > $OUT/bench_driver_cmd.json 2>/dev/null; cut -c1-200 $OUT/bench_driver_cmd.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/driver -o bench -- python3 bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_driver_cmd_rocprof.json 2> $OUT/bench_driver_cmd_rocprof.err; echo "rocprof rc $?"
grep -E "k_wav_to_mel|k_minmax" $OUT/driver/*kernel_stats.csv | cut -c1-160
find $OUT/driver -name "*kernel_trace.csv" -delete
PMC_OUT=measure/pmc bash scripts/gpu_pmc.sh "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" > $OUT/pmc.log 2>&1; tail -3 $OUT/pmc.log
timeout -k 10 300 python3 scripts/gpu_shapes.py 2>&1 | grep -v amdgpu.ids | tee $OUT/shapes.log
timeout -k 10 900 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "full bench rc $?"; cut -c1-300 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
for mode in module engine graph; do
  timeout -k 10 300 python3 scripts/gpu_fwdprof.py 20 $mode > $OUT/c3_$mode.pre.log 2>&1; tail -1 $OUT/c3_$mode.pre.log
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3_$mode -o c3 -- python3 scripts/gpu_fwdprof.py 20 $mode > $OUT/c3_$mode.log 2>&1; grep "fwd\[" $OUT/c3_$mode.log
  python3 scripts/trace_steps.py $OUT/c3_$mode/c3_kernel_trace.csv k_wav_to_mel 12 $OUT/c3_${mode}_step_kernel_stats.csv
  find $OUT/c3_$mode -name "*kernel_trace.csv" -delete
done
timeout -k 10 300 python3 scripts/gpu_c4prof.py 10 > $OUT/c4.pre.log 2>&1; tail -1 $OUT/c4.pre.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 scripts/gpu_c4prof.py 10 > $OUT/c4.log 2>&1; grep "train step" $OUT/c4.log
python3 scripts/trace_steps.py $OUT/c4/c4_kernel_trace.csv k_wav_to_mel 6 $OUT/c4_step_kernel_stats.csv
find $OUT/c4 -name "*kernel_trace.csv" -delete
# the Winograd weight gradient: microbenchmark against MIOpen on the step's shapes, and the step with / without it (eager, graphed)
timeout -k 10 300 python3 scripts/gpu_wino_wrw_bench.py 2>&1 | grep -v amdgpu.ids > $OUT/wino_wrw_microbench.log; tail -10 $OUT/wino_wrw_microbench.log
for v in 0 1 0 1; do
  echo "IRIS_WINO_TRAIN_WRW=$v:"; IRIS_WINO_TRAIN_WRW=$v timeout -k 10 200 python3 scripts/gpu_graph_train.py 30 2>&1 | grep -v amdgpu.ids | tail -2
done > $OUT/c4_wino_wrw_ab.log 2>&1; cat $OUT/c4_wino_wrw_ab.log
# RCCL at world size 1: the bench line with the forced process group, and a kernel trace of DDP steps
IRIS_FORCE_PG=1 timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 --extra-steps 10 --no-cpu-baseline > $OUT/bench_rccl_world1.json 2> $OUT/bench_rccl_world1.err; echo "rccl bench rc $?"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/rccl_w1 -o rccl_w1 -- python3 scripts/gpu_rccl_world1.py > $OUT/rccl_world1_trace.log 2>&1; echo "rccl trace rc $?"; grep "^{" $OUT/rccl_world1_trace.log
(echo "# kernels with nccl / rccl in their name in the rocprofv3 kernel trace of scripts/gpu_rccl_world1.py (3 + 3 DDP training steps, average_bn_statistics, explicit out-of-place collectives):"; grep -i -c -E "nccl|rccl" $OUT/rccl_w1/*kernel_stats.csv; grep -i -E "nccl|rccl" $OUT/rccl_w1/*kernel_stats.csv | cut -c1-200) >> $OUT/rccl_world1_trace.log
cp $OUT/rccl_w1/*kernel_stats.csv $OUT/rccl_world1_kernel_stats.csv
find $OUT/rccl_w1 -name "*kernel_trace.csv" -delete
IRIS_BENCH_SHARE_GPU=1 timeout -k 10 600 python3 bench.py --gpus 4 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_share4_gloo.json 2> $OUT/bench_share4.err; echo "share4 rc $?"
timeout -k 10 1100 python3 -m pytest tests -q -m gpu -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_gpu.log
