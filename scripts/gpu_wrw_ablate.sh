# timing experiments on the Winograd weight-gradient kernel: variants built on the CPU side as scripts/microbench/libwino_<tag>.so
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/wrw
for lib in scripts/microbench/libwino_*.so; do
  tag=$(basename $lib .so | sed s/libwino_//)
  echo "== $tag"
  WINO_LIB=$lib WINO_ROWS=${WINO_ROWS:-short} timeout -k 10 200 python3 scripts/gpu_wino_wrw_bench.py time 2>&1 | grep -E "wino|sum" 
done 2>&1 | tee gpurun_out/wrw/ablate.log
export WINO_LIB=scripts/microbench/libwino_base.so
export WINO_ROWS=${WINO_ROWS:-short}
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/wrw/prof -o wrw -- python3 scripts/gpu_wino_wrw_bench.py time > gpurun_out/wrw/prof.log 2>&1
grep -E "k_wino_wrw|igemm_wrw" gpurun_out/wrw/prof/*kernel_stats.csv | cut -c1-200 | tee -a gpurun_out/wrw/ablate.log
python3 scripts/trace_by_shape.py gpurun_out/wrw/prof/*kernel_trace.csv "k_wino_wrw|igemm_wrw" | tee -a gpurun_out/wrw/ablate.log
find gpurun_out/wrw/prof -name "*kernel_trace.csv" -delete
