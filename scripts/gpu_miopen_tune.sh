# Tune MIOpen's solvers for the CRNN's convolution shapes on THIS GPU (exhaustive search of every solver's tuning space,
# MIOPEN_FIND_ENFORCE=3) and collect the user perf-db / find-db it writes.  The result is shipped in challenge_amd/miopen_db/
# (sj_train.configure_miopen points MIOPEN_USER_DB_PATH at a copy of it), so that a fresh box starts from the tuned kernels:
#   gpurun -- 'bash scripts/gpu_miopen_tune.sh' && cp gpurun_out/miopen_tune/db/*db.txt challenge_amd/miopen_db/
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/miopen_tune
mkdir -p $OUT/db
cp challenge_amd/miopen_db/*db.txt $OUT/db/ 2>/dev/null   # incremental: start from what is shipped
export MIOPEN_USER_DB_PATH=$OUT/db
( while true; do date >> $OUT/heartbeat.log; sleep 45; done ) &
HB=$!
: > $OUT/result.log
run() { echo "== $*" | tee -a $OUT/result.log; "$@" 2>&1 | grep "train step\|fwd\[" | tee -a $OUT/result.log; }
run env MIOPEN_FIND_ENFORCE=3 timeout -k 10 900 python3 scripts/gpu_c4prof.py 3
run env MIOPEN_FIND_ENFORCE=3 timeout -k 10 600 python3 scripts/gpu_fwdprof.py 3 engine
run env MIOPEN_FIND_ENFORCE=3 timeout -k 10 600 python3 scripts/gpu_fwdprof.py 3 module
run env MIOPEN_FIND_ENFORCE=3 timeout -k 10 900 python3 scripts/gpu_c4prof.py 3 bf16
echo "== with the tuned db" | tee -a $OUT/result.log
run timeout -k 10 300 python3 scripts/gpu_c4prof.py 30
run timeout -k 10 300 python3 scripts/gpu_fwdprof.py 30 engine
run timeout -k 10 300 python3 scripts/gpu_fwdprof.py 30 graph
run timeout -k 10 300 python3 scripts/gpu_fwdprof.py 30 module
run timeout -k 10 300 python3 scripts/gpu_c4prof.py 30 bf16
kill $HB
ls -la $OUT/db | tee -a $OUT/result.log
