cd $GRAFT_REPO_ROOT
O=gpurun_out/r2r; mkdir -p $O
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "magmel or magphase_to_mel or unfused or data_utils or dataset or mixer or eval" 2>&1 | tail -3
for e in "" "IRIS_MAGMEL_FPL1=1"; do echo "== $e"; env $e python3 scripts/gpu_shapes.py 2>&1 | grep -o "^.\{42\}\|magmel.*" | paste - - ; done 2>&1 | tee $O/magmel.log
