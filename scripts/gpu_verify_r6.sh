#!/bin/bash
# On the GPU box: smoke(), then the GPU suite and the default bench line (scripts/gpu_check_and_bench.sh).
out=gpurun_out/r6; mkdir -p $out
tag=${1:-verify}
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke_$tag.log 2>&1
rc=$?; tail -4 $out/smoke_$tag.log; echo "smoke rc $rc"
if [ $rc -ge 124 ]; then echo "smoke killed at its limit: stopping"; exit $rc; fi
scripts/gpu_check_and_bench.sh $tag
rc2=$?
exit $(( rc > rc2 ? rc : rc2 ))
