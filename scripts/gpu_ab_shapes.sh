# A/B of two builds over the shapes table (scripts/gpu_shapes.py): gpu_ab_shapes.sh name1 name2 [grep-pattern]
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  for n in "$1" "$2"; do
    L=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend_$n.so
    [ "$n" = prod ] && L=$GRAFT_REPO_ROOT/challenge_amd/csrc/libiris_frontend.so
    IRIS_LIB=$L python3 scripts/gpu_shapes.py 2>/dev/null | grep -E "${3:-N}" | sed -E "s/^/$n: /; s/fused\+minmax\+log[^|]*\| mel only[^|]*\| per frame[^|]*\| //"
  done
done
