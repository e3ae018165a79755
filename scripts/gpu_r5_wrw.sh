# Winograd weight gradient in the training step: tests, then the step A/B (IRIS_WINO_TRAIN_WRW = 0 / 1, alternating) and the
# per-step kernel statistics with it on
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out/wrw_step
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 600 python3 -m pytest tests/test_transforms_gpu.py -q -x -k "winograd" > $OUT/pytest_wino.log 2>&1; echo "pytest rc $?"; tail -3 $OUT/pytest_wino.log
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "IRIS_WINO_TRAIN_WRW=$v: "
    IRIS_WINO_TRAIN_WRW=$v timeout -k 10 200 python3 scripts/gpu_c4prof.py 20 2>&1 | grep "train step"
  done
done 2>&1 | tee $OUT/c4_wrw_ab.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c4 -o c4 -- python3 scripts/gpu_c4prof.py 10 > $OUT/c4.log 2>&1; grep "train step" $OUT/c4.log
python3 scripts/trace_steps.py $OUT/c4/c4_kernel_trace.csv k_wav_to_mel 6 $OUT/c4_step_kernel_stats.csv
find $OUT/c4 -name "*kernel_trace.csv" -delete
head -12 $OUT/c4_step_kernel_stats.csv | cut -c1-150
