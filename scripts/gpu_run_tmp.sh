cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r3k; rm -rf $O; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_transforms_gpu.py -x -q -m gpu -k "wave" 2>&1 | tail -2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o wave -- python3 - > $O/wave_prof.log 2>&1 <<'PY'
import sys, os
sys.path.insert(0, os.getcwd())
import torch
from challenge_amd import sj_train as S
dev = torch.device("cuda", 0)
cfg = S.ARGS().get(['--v', '9', '--n_mels', '80', '--n_frame', '512', '--n_chan', '2', '--batch_size', '64'])
ds = iter(S.make_wave_dataset(cfg, True, sources=S.synthetic_wave_sources(2, 3, 256, n_bg=16, n_voice=64, n_noise=32, seed=0), device=dev, seed=0))
for _ in range(30): next(ds)
torch.cuda.synchronize()
PY
python3 - $O/prof/wave_kernel_stats.csv <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print('%-70s calls=%4s avg_us=%8.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3))
PY
find $O -name "*kernel_trace.csv" -delete
