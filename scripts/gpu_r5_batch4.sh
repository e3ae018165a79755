cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b4
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_transforms_gpu.py -x -q -k "winograd or inference_engine or predict_uses or eval_path or conv32 or other_model" 2>&1 | tail -4 | tee $OUT/pytest_wino.log
