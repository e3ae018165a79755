cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -1 $O/pytest_gpu.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])"
