cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
timeout -k 10 400 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python3 -c "
import json;d=json.load(open('$O/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['step_frac'], d['roofline']['launches_timed'], d['stft_mel_fwd_audio_s_per_s']); e=d['extra']; print(e['two_stream_pipeline'], e['c5_stereo_2048_128mel']['fp32_banded_default']['k1_us'], e['c5_stereo_2048_128mel']['fp16_mfma']['k1_us'], [ (x['batch'], x['k1_us'], x['k1_frac_of_8TBs']) for x in e['k1_batch_sweep']])"
