cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_frontend_gpu.py -x -q -m gpu -k "pipelined or side_stream" 2>&1 | tail -3
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'], d['ms_per_step']); print(d['extra']['two_stream_pipeline'])"
