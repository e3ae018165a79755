cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
for v in diag12 diag16; do
  echo "#### $v rotating"
  IRIS_LIB=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_$v.so bash scripts/gpu_phase.sh 4608 512 2>&1
done | tee gpurun_out/r2a/phase_rot.log
