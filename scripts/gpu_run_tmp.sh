cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
for mode in "" "--resident"; do
for ab in 512 520 584; do
  echo "== $mode IRIS_ABLATE=$ab"
  IRIS_LIB=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_diag.so IRIS_ABLATE=$ab timeout 300 python3 bench.py $mode --steps 50 --warmup 10 --no-cpu-baseline --no-extras 2>&1 | grep -E "256 workgroups"
done; done 2>&1 | tee $O/abl_ring.log
