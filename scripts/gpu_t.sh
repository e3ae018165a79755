cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_frontend_gpu.py -x -q -m gpu 2>&1 | tail -5
python scripts/gpu_shapes.py 2>&1 | tail -6 | cut -c1-165
