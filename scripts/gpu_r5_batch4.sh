cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r5b4
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 900 python3 -m pytest tests/test_transforms_gpu.py -x -q -k "winograd or inference_engine or predict_uses or eval_path" 2>&1 | tail -6 | tee $OUT/pytest_wino.log
for w in 1 0; do
IRIS_WINO=$w timeout -k 10 600 python3 - <<'PY' 2>&1 | grep -v amdgpu.ids | tee -a $OUT/c3_wino_ab.log
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from challenge_amd import sj_train as S
S.configure_miopen()
dev = torch.device("cuda", 0)
cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '512', '--n_chan', '1', '--batch_size', '64'])
torch.manual_seed(0)
model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
fe = S.WaveFrontend(1024, 256, 64, 16000, 1, 64, 130816, dev, training=True, device_draw=True, seed=99)
wav = torch.randn(64, 1, 130816, device=dev) * 0.1
eng = S.InferenceEngine(model, fe, wav)
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
te = timed(eng.eager); tg = timed(eng.replay) if eng.graph_ok else float('nan')
print(f"IRIS_WINO={os.environ.get('IRIS_WINO')}: wino_convs {eng.wino_convs} hip_convs {eng.hip_convs} | engine eager {1e3*te:.3f} ms, hipGraph replay {1e3*tg:.3f} ms = {64*130816/16000/tg:.0f} audio-s/s (graph_ok {eng.graph_ok} {eng.graph_error})")
PY
done
