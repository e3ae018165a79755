#!/bin/bash
# On the GPU box: the GPU test suite, then the default bench line (+ the driver's command), logs under gpurun_out/r6/.
# A step killed at its time limit ends the call (no further GPU step after a hang).
out=gpurun_out/r6; mkdir -p $out
tag=${1:-run}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -s > $out/pytest_gpu_$tag.log 2>&1
rc=$?; tail -8 $out/pytest_gpu_$tag.log
if [ $rc -ge 124 ]; then echo "pytest killed at its limit (rc $rc): stopping"; exit $rc; fi
timeout -k 10 700 python bench.py > $out/bench_default_$tag.json 2> $out/bench_default_$tag.err
rc2=$?; echo "bench rc $rc2"; tail -c 400 $out/bench_default_$tag.err
exit $(( rc > rc2 ? rc : rc2 ))
