set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_frontend_gpu.py tests/test_transforms_gpu.py -x -q -m gpu 2>&1 | tail -8
timeout 300 python scripts/gpu_shapes.py 2>&1 | tail -8 | tee gpurun_out/shapes_tri.log
IRIS_MAGMEL_GENERIC=1 timeout 300 python scripts/gpu_shapes.py 2>&1 | tail -8 | tee gpurun_out/shapes_generic.log
