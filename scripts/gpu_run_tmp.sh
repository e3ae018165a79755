cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r2i; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout -k 10 120 python __graft_entry__.py smoke 2>&1 | tail -1
rm -rf $O/prof; mkdir -p $O/prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras > $O/prof_stdout.log 2>&1; echo "prof rc $?"
for f in $(find $O/prof -name "*kernel_stats.csv"); do cut -c1-150 $f | head -6; done
for f in $(find $O/prof -name "*kernel_trace.csv"); do head -50 $f > $f.head; rm $f; done
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; cut -c1-600 $O/bench.json
