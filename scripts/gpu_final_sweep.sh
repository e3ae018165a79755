#!/bin/bash
# Round-end validation of the build as it is: randomised parity sweeps (frontend vs the NumPy oracle, CRNN passes vs the stock torch /
# MIOpen ops - also with the split-bf16 convolutions), the edge geometries, the c4 parity tail.  Logs under gpurun_out/r6/final_*.
out=gpurun_out/r6; mkdir -p $out
rc_all=0
run() {   # name, limit, command...
  name=$1; lim=$2; shift 2
  timeout -k 10 $lim "$@" > $out/final_$name.log 2>&1
  rc=$?; echo "$name rc $rc: $(tail -1 $out/final_$name.log | cut -c1-160)"
  if [ $rc -ge 124 ]; then echo "$name killed at its limit: stopping"; exit $rc; fi
  if [ $rc -ne 0 ]; then rc_all=$rc; fi
}
run fuzz_frontend 800 python3 scripts/gpu_fuzz.py ${FUZZ_N:-150} 11
run fuzz_train 800 python3 scripts/gpu_fuzz_train.py ${FUZZ_T:-80} 12
run fuzz_train_split 800 python3 scripts/gpu_fuzz_train.py ${FUZZ_S:-60} 13 --split
run stress 300 python3 scripts/gpu_stress.py
exit $rc_all
