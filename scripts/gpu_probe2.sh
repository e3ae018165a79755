cd $GRAFT_REPO_ROOT
for fm in NORMAL HYBRID DYNAMIC_HYBRID; do
  echo "=== MIOPEN_FIND_MODE=$fm"
  t0=$(date +%s)
  MIOPEN_FIND_MODE=$fm timeout 600 python scripts/gpu_trainprobe.py 2>&1 | grep iter | tail -2
  echo "elapsed $(( $(date +%s) - t0 )) s"
done
echo "=== cudnn.benchmark"
MIOPEN_FIND_MODE=NORMAL timeout 600 python - <<'PY' 2>&1 | grep iter | tail -2
import torch, runpy, sys
torch.backends.cudnn.benchmark = True
sys.argv=['x']
runpy.run_path('scripts/gpu_trainprobe.py', run_name='__main__')
PY
