set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocminfo | grep -E "Marketing Name|gfx" | head -4
timeout 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -30 > gpurun_out/pytest_gpu.log; cat gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
timeout 600 python bench.py --steps 200 --warmup 20 2>&1 | tail -3 | tee gpurun_out/bench1.json
