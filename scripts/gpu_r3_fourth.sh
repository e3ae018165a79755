# round 3, fourth GPU call: fixed cost of a short timed window (steps sweep, active-wait A/B), PMC traffic passes, full bench
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3d
rm -rf $OUT; mkdir -p $OUT
for wait in "" 2000; do
  for k in 10 20 50 200 1000; do
    for i in 1 2 3; do
      ROC_ACTIVE_WAIT_TIMEOUT=$wait timeout -k 10 200 python3 bench.py --steps $k --warmup 5 --no-cpu-baseline --no-extras --no-kernel-events 2>/dev/null | python3 -c "
import sys,json
r=json.loads(sys.stdin.readline()); print('active_wait=[$wait] steps $k: ms_per_step', r['ms_per_step'])"
    done
  done
done
PMC_OUT=r3d/pmc bash scripts/gpu_pmc.sh "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"
timeout -k 10 900 python3 bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "full bench rc $?"; cut -c1-300 $OUT/bench_full.json; tail -5 $OUT/bench_full.err
