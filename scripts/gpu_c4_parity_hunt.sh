#!/bin/bash
# the tail of the c4 gradient-parity figure: exact kernels, then the split-bf16 ones (scripts/gpu_c4_parity_hunt.py)
out=gpurun_out/r6; mkdir -p $out
n=${1:-30}
timeout -k 10 520 python3 scripts/gpu_c4_parity_hunt.py $n > $out/c4_parity_hunt_exact.log 2>&1
rc=$?; tail -3 $out/c4_parity_hunt_exact.log
if [ $rc -ge 124 ]; then echo "killed at its limit"; exit $rc; fi
timeout -k 10 520 python3 scripts/gpu_c4_parity_hunt.py $n --split > $out/c4_parity_hunt_split.log 2>&1
rc2=$?; tail -3 $out/c4_parity_hunt_split.log
exit $(( rc > rc2 ? rc : rc2 ))
