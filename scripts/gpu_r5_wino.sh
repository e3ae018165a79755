cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_wino
if [ -n "$WINO_LIB" ]; then echo "== $WINO_LIB (timing experiment: results wrong)"; timeout -k 10 600 python3 scripts/gpu_wino_bench.py time 2>&1 | grep -v "amdgpu.ids" | tail -14; exit 0; fi
timeout -k 10 600 python3 scripts/gpu_wino_bench.py ${1:-all} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r5_wino/wino_bench.log
