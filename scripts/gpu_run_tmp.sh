cd $GRAFT_REPO_ROOT
O=gpurun_out/r2v; mkdir -p $O
for i in 1 2; do for v in prod o2w10 o2w8; do L=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_$v.so; [ $v = prod ] && L=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend.so
 echo "== $v"; IRIS_LIB=$L python3 scripts/gpu_shapes.py 2>&1 | grep -E "^ref|^sml" | cut -c1-150
done; done 2>&1 | tee $O/occ.log
