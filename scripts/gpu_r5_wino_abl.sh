cd $GRAFT_REPO_ROOT
for a in ${ABLS:-15 14 12 4 8 1 0}; do
  L=$GRAFT_REPO_ROOT/scripts/microbench/libwino_abl$a.so; [ $a = 0 ] && L=$GRAFT_REPO_ROOT/scripts/microbench/libwino.so
  echo "== ablate $a"; WINO_LIB=$L WINO_ROWS=short timeout -k 10 300 python3 scripts/gpu_wino_bench.py time 2>&1 | grep "pool 1"
done
