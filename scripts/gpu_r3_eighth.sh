# round 3, eighth GPU call: where the fused kernel's time goes (diag build, per-workgroup clock stamps incl. the epilogue)
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3h
rm -rf $OUT; mkdir -p $OUT
for i in 1 2; do
IRIS_LIB=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_diag.so IRIS_ABLATE=512 timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-events 2>&1 | grep "iris dbg" | tee -a $OUT/epilogue_phases.log
done
IRIS_LIB=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_diag.so IRIS_ABLATE=512 IRIS_EPILOGUE=1 timeout -k 10 200 python3 bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline --no-kernel-events 2>&1 | grep "iris dbg" | tee -a $OUT/epilogue_phases.log
