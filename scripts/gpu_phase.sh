# per-phase cycle breakdown of the fused kernel's frame loop (diag build)
cd $GRAFT_REPO_ROOT
export IRIS_LIB=${IRIS_LIB:-$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_diag.so}
for ab in "$@"; do
  echo "== IRIS_ABLATE=$ab"
  IRIS_ABLATE=$ab timeout 300 python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras 2>&1 | grep -E "iris dbg|kernel_ms" | sed -E 's/.*("kernel_ms": [0-9.]+).*/\1/'
done
