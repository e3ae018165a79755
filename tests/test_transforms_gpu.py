"""GPU tests of the drop-in modules (transforms / data_utils / sj_train), written to read
like the reference's transforms_test.py and pipeline_test.py, with the oracle as checker."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import frontend_ref as R

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda", 0)


@pytest.fixture(scope="module")
def kats(golden_dir):
    with open(os.path.join(golden_dir, "ref_kats.json")) as f:
        return json.load(f)


def mods():
    from challenge_amd import data_utils, sj_train, transforms
    return transforms, data_utils, sj_train


def test_mask(dev, kats):
    T, _, _ = mods()
    for key in ("mask_axis0", "mask_axis1"):
        k = kats[key]
        org = torch.tensor(k["org"], device=dev)
        out = T.mask_apply(org, k["axis"], np.stack([k["offsets"], k["sizes"]], 1))
        assert out.dtype == org.dtype and np.array_equal(out.cpu().numpy(), np.array(k["expected"]))
    # random masks: product of n_mask zero bands, each shorter than max_mask_size
    T.set_seed(7)
    x = torch.ones(3, 40, 5, device=dev)
    for _ in range(20):
        out = T.mask(x, axis=1, max_mask_size=6, n_mask=3).cpu().numpy()
        rows = out[0, :, 0]
        assert np.all(out == rows[None, :, None]) and set(np.unique(rows)) <= {0.0, 1.0}
        assert (rows == 0).sum() <= 3 * 5 and rows[-1] == 1  # the last index is never masked
    with pytest.raises(ValueError):
        for _ in range(50):
            T.mask(torch.ones(4, 3, device=dev), axis=0, max_mask_size=16)


def test_random_shift(dev, kats):
    T, _, _ = mods()
    k = kats["random_shift"]
    out = T.random_shift_apply(torch.tensor(k["org"], device=dev), k["axis"], k["width"], k["offset"])
    assert np.array_equal(out.cpu().numpy(), np.array(k["expected"]))
    assert tuple(T.random_shift(torch.ones(5, 4, device=dev), axis=1, width=3).shape) == (5, 4)


def test_magphase_to_mel(dev, kats):
    T, _, _ = mods()
    k = kats["magphase_to_mel_shapes"]
    rng = np.random.default_rng(0)
    to_mel = T.magphase_to_mel(k["n_mels"])
    ref_mel = R.magphase_to_mel(k["n_mels"])
    for case in k["cases"]:
        magphase = np.abs(rng.standard_normal(case["in"])).astype(np.float32)
        mel = to_mel(torch.from_numpy(magphase).to(dev))
        assert list(mel.shape) == case["out"]
        ref = ref_mel(magphase)
        assert np.abs(mel.cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    mel, y = to_mel(torch.from_numpy(magphase).to(dev), "labels")
    assert y == "labels"
    with pytest.raises(ValueError):
        to_mel(torch.zeros(257, 4, device=dev))
    with pytest.raises(RuntimeError):
        to_mel(torch.zeros(257, 10, 4))
    # arbitrary bin count + custom edges (mel-only plan)
    x = np.abs(rng.standard_normal((2, 100, 7, 2))).astype(np.float32)
    f = T.magphase_to_mel(12, 100, 8000, lower_edge_hertz=50.0, upper_edge_hertz=3900.0)
    ref = R.magphase_to_mel(12, 100, 8000, lower_edge_hertz=50.0, upper_edge_hertz=3900.0)(x)
    assert np.abs(f(torch.from_numpy(x).to(dev)).cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    # external matrix (the route to exact TensorFlow parity): here a perturbed triangular W
    # and a dense one; the closure must use exactly the matrix it was given
    w = R.linear_to_mel_weight_matrix(80, 257, 16000)
    x = np.abs(rng.standard_normal((3, 257, 9, 4))).astype(np.float32)
    for ext in (w * (1 + 1e-3 * rng.standard_normal(w.shape).astype(np.float32)), np.abs(rng.standard_normal(w.shape)).astype(np.float32)):
        g = T.magphase_to_mel(80, 257, 16000, mel_matrix=ext)
        assert np.array_equal(g.mel_matrix, ext)
        ref = np.einsum("bftc,fm->bmtc", x[..., :2].astype(np.float64), ext.astype(np.float64))
        assert np.abs(g(torch.from_numpy(x).to(dev)).cpu().numpy() - ref).max() <= 1e-5 * np.abs(ref).max()
    with pytest.raises(ValueError):
        T.magphase_to_mel(80, 257, 16000, mel_matrix=w[:100])


def test_log_magphase(dev, kats):
    T, _, _ = mods()
    k = kats["log_magphase"]
    out = T.log_magphase(torch.tensor(k["specs"], dtype=torch.float32, device=dev), n_chan=k["n_chan"])
    assert np.allclose(out.cpu().numpy(), np.array(k["expected"]), atol=1e-5)


def test_minmax_norm_magphse(dev):
    T, _, _ = mods()
    rng = np.random.default_rng(1)
    mag = rng.standard_normal((5, 10, 2))
    phase = (2 * rng.random((5, 10, 2)) - 1) * np.pi
    out = T.minmax_norm_magphase(torch.from_numpy(np.concatenate([mag, phase], -1)).to(dev)).cpu().numpy()
    assert np.allclose(out.min(axis=(1, 2)), 0, atol=1e-6) and np.allclose(out.max(axis=(1, 2)), 1, atol=1e-6)


def test_complex_to_magphase_and_back(dev, kats):
    T, _, _ = mods()
    k = kats["phasors"]
    c = torch.tensor(k["complex"], dtype=torch.float32, device=dev)
    mp = torch.tensor(k["magphase"], dtype=torch.float32, device=dev)
    assert np.allclose(T.complex_to_magphase(c).cpu().numpy(), np.array(k["magphase"]), atol=1e-6)
    assert np.allclose(T.magphase_to_complex(mp).cpu().numpy(), np.array(k["complex"]), atol=1e-6)
    rng = np.random.default_rng(2)
    z = rng.standard_normal((3, 17, 9, 4)).astype(np.float32)
    out, y = T.complex_to_magphase(torch.from_numpy(z).to(dev), 5)
    assert y == 5 and np.allclose(out.cpu().numpy(), R.complex_to_magphase(z), atol=1e-5)
    back = T.magphase_to_complex(out).cpu().numpy()
    assert np.allclose(back, z, atol=1e-5)


def test_phase_vocoder(dev, kats):
    T, _, _ = mods()
    k = kats["phase_vocoder_shapes"]
    spec = torch.randn(k["n_freq"], k["time"], k["chan2"], device=dev)
    assert T.phase_vocoder(spec, 1.0) is spec
    for rate in k["rates"]:
        assert list(T.phase_vocoder(spec, rate=rate).shape) == [k["n_freq"], int(np.ceil(k["time"] / rate)), k["chan2"]]


def test_data_utils_chain_matches_oracle(dev):
    T, D, _ = mods()
    rng = np.random.default_rng(3)
    wav = rng.standard_normal((2, 6000)).astype(np.float32) * 3.0
    spec = D.load_wav_array(wav, 16000, dev)                      # normalize + STFT(512) -> [257, T, 4]
    ref_spec = R.load_wav_array(wav, 512)
    assert tuple(spec.shape) == ref_spec.shape == (257, 1 + 6000 // 256, 4)
    assert np.abs(spec.cpu().numpy() - ref_spec).max() <= 3e-6 * np.abs(ref_spec).max()
    x = D.stft_filter(16)(spec)                                   # the eval path, metrics.py:50-54
    x = T.complex_to_magphase(x)
    x = T.magphase_to_mel(80)(x)
    x = D.minmax(x)
    x = D.log_on_mel(x)
    r = R.stft_filter(16)(ref_spec)
    r = R.log_on_mel(R.minmax(R.magphase_to_mel(80)(R.complex_to_magphase(r))))
    assert tuple(x.shape) == r.shape == (80, 24, 2)
    assert np.abs(np.exp(x.cpu().numpy()) - np.exp(r)).max() <= 1e-5   # per-mel-row min-max (unbatched quirk)
    # any other sample rate goes through the device resampler first (data_utils.py:20-21), as the oracle's chain does
    spec44 = D.load_wav_array(wav, 44100, dev)
    ref44 = R.load_wav_array(wav, 512, sample_rate=44100)
    n16 = -(-160 * 6000 // 441)
    assert tuple(spec44.shape) == ref44.shape == (257, 1 + n16 // 256, 4)
    assert np.abs(spec44.cpu().numpy() - ref44).max() <= 3e-6 * np.abs(ref44).max()
    n = D.normalize(torch.from_numpy(wav).to(dev)).cpu().numpy()
    assert np.abs(n - R.normalize(wav)).max() <= 1e-6
    m = torch.rand(3, 8, 9, 2, device=dev)
    a = D.minmax_log_on_mel(m).cpu().numpy()
    assert np.abs(np.exp(a) - np.exp(R.minmax_log_on_mel(m.cpu().numpy()))).max() <= 2e-6


def test_augment_on_complex_spec(dev):
    T, D, _ = mods()
    T.set_seed(11)
    spec = torch.randn(257, 512, 4, device=dev)
    out, y = D.augment(spec, "y")
    assert y == "y" and tuple(out.shape) == tuple(spec.shape)
    keep = (out != 0).float()
    t_zero = (keep.sum((0, 2)) == 0).sum().item()
    f_zero = (keep.sum((1, 2)) == 0).sum().item()
    assert t_zero <= 6 * 23 and f_zero <= 15
    assert torch.equal(out[keep.bool()], spec[keep.bool()])


def test_make_dataset_end_to_end(dev):
    T, D, S = mods()
    T.set_seed(0)
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '80', '--n_frame', '128', '--batch_size', '3', '--max_voices', '4',
                        '--max_noises', '2', '--synthetic'])
    ds = S.make_dataset(cfg, training=True)
    for x, y in ds.take(2):
        assert tuple(x.shape) == (3, 80, 128, 2) and tuple(y.shape) == (3, 4, 3)
        assert x.is_cuda and torch.isfinite(x).all()
        flat = x.reshape(3, -1)
        assert torch.allclose(flat.max(1).values, torch.zeros(3, device=dev), atol=1e-6)
        assert set(np.unique(y.cpu().numpy())) <= {0.0, 1.0}
    cfg1 = S.ARGS().get(['--v', '1', '--n_chan', '3', '--n_frame', '64', '--batch_size', '2', '--synthetic',
                         '--name', 'filter_nominmax'])
    x, y = next(iter(S.make_dataset(cfg1, training=False)))   # stereo_mono -> 3 channels, no min-max, filter
    assert tuple(x.shape) == (2, 80, 64, 3) and tuple(y.shape) == (2, 64, 3)
    # 'nominmax': the log of the un-normalised mel - its maximum is whatever the drawn mixture gives (0.25 .. 3 observed), not the
    # exact 0 = ln(1) that min-max leaves behind
    assert abs(float(x.max())) > 1e-3
    # n_chan == 1 maps mono_chan, whose broadcast add leaves an odd channel axis on stereo
    # sources (the reference quirk, data_utils.py:73-76): rejected loudly rather than mimicked
    cfg2 = S.ARGS().get(['--v', '1', '--n_chan', '1', '--n_frame', '64', '--batch_size', '2', '--synthetic'])
    with pytest.raises(ValueError):
        next(iter(S.make_dataset(cfg2, training=False)))


def test_wave_frontend_matches_oracle(dev):
    _, _, S = mods()
    rng = np.random.default_rng(5)
    wav = R.normalize(rng.standard_normal((4, 16000)).astype(np.float32)).reshape(4, 1, 16000)
    fe = S.WaveFrontend(1024, 256, 64, 16000, 1, 4, 16000, dev, training=True, filter_bins=3)
    fe.rng = np.random.default_rng(9)
    out = fe(torch.from_numpy(wav).to(dev)).cpu().numpy()
    # replay the draws: ONE set per sample (6 time bands, then 1 frequency band, batch-vectorised) and nothing else
    check = np.random.default_rng(9)
    tb, fb = S._du.augment_draw_batch(4, 63, 513, check)
    assert tb.shape == (4, 6, 2) and fb.shape == (4, 1, 2)
    assert fe.rng.bit_generator.state == check.bit_generator.state  # the frontend consumed exactly these draws (round 1 drew twice)
    fb = np.concatenate([fb, np.tile(np.array([[[1, 3]]], np.int32), (4, 1, 1))], axis=1)
    ref = R.wav_to_logmel(wav, 1024, 256, 64, 16000, t_bands=tb, f_bands=fb)
    assert out.shape == ref.shape == (4, 64, 63, 1)
    assert np.abs(np.exp(out) - np.exp(ref)).max() <= 5e-6


def test_train_step_on_gpu(dev):
    _, _, S = mods()
    torch.manual_seed(0)
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '128', '--n_chan', '1'])
    model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
    fe = S.WaveFrontend(1024, 256, 64, 16000, 1, 8, 128 * 256 - 256, dev, training=True)
    wav = torch.randn(8, 1, 128 * 256 - 256, device=dev) * 0.1
    x = fe(wav)
    assert tuple(x.shape) == (8, 64, 128, 1)
    y = (torch.rand(8, 4, 3, device=dev) > 0.9).float()
    first = float(model.train_step((x, y))['loss'])
    for _ in range(10):
        last = float(model.train_step((x, y))['loss'])
    assert math.isfinite(last) and last < first
    # inference copy with the BatchNorms folded in: the same function on the device (MIOpen convolutions, channels_last)
    folded = S.fold_batchnorm(model)
    with torch.no_grad():
        ref, got = model.eval()(x), folded(x)
    assert got.is_cuda and float((ref - got).abs().max()) <= 1e-4, float((ref - got).abs().max())


def test_wave_frontend_device_draw(dev):
    _, _, S = mods()
    fe = S.WaveFrontend(1024, 256, 64, 16000, 1, 16, 130816, dev, training=True, device_draw=True, filter_bins=3)
    tb, fb = fe.draw_bands_device(16, 512)
    tb, fb = tb.cpu().numpy(), fb.cpu().numpy()
    assert tb.shape == (16, 6, 2) and fb.shape == (16, 1, 2)
    assert tb[..., 1].max() < 24 and tb[..., 1].min() >= 0 and np.all(tb[..., 0] + tb[..., 1] < 512 + (tb[..., 1] == 0))
    assert fb[..., 1].max() < 16 and np.all(fb[..., 0] + fb[..., 1] < 513 + (fb[..., 1] == 0)) and tb[..., 0].min() >= 0
    x = fe(torch.randn(16, 1, 130816, device=dev) * 0.1)
    assert tuple(x.shape) == (16, 64, 512, 1) and torch.isfinite(x).all()


def test_fused_agc_matches_torch_reference(dev):
    """iris_agc_clip (one launch) == adaptive_clip_grad + clip_grad_value_ (sj_train.py:145-155, :435)."""
    _, _, S = mods()
    torch.manual_seed(1)
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '64', '--n_chan', '2'])
    model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    params = list(model.parameters())
    for i, p in enumerate(params):  # a mix of small and large gradients so that both branches run
        scale = 10.0 if i % 3 == 0 else (1e-4 if i % 3 == 1 else 0.05)
        p.grad = (torch.randn_like(p) * scale)
    ref = S.adaptive_clip_grad(params, [p.grad.clone() for p in params])
    ref = [torch.clamp(g, -0.01, 0.01) for g in ref]
    ref_noclip = S.adaptive_clip_grad(params, [p.grad.clone() for p in params])
    keep = [p.grad.clone() for p in params]
    agc = S.FusedAGC(params)
    agc(0.01, 1e-3, 0.01)
    assert not agc._slow
    for p, r in zip(params, ref):
        assert torch.allclose(p.grad, r, rtol=2e-5, atol=1e-9), (tuple(p.shape), float((p.grad - r).abs().max()))
    for p, g in zip(params, keep):
        p.grad.copy_(g)
    agc(0.01, 1e-3, None)
    for p, r in zip(params, ref_noclip):
        assert torch.allclose(p.grad, r, rtol=2e-5, atol=1e-9)


def test_eval_path_on_gpu(dev):
    """metrics.evaluate's chain (metrics.py:40-81) without the scoring: features match the
    oracle, predictions are thresholded frames of the clip's length."""
    from challenge_amd import inference as I
    _, D, S = mods()
    rng = np.random.default_rng(21)
    wav = rng.standard_normal((2, 40000)).astype(np.float32)
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '80', '--n_frame', '64', '--n_chan', '2'])
    spec = D.load_wav_array(wav, 16000, dev)
    feats = I.features_for_eval(spec, cfg)
    r = R.stft_filter(16)(R.load_wav_array(wav, 512))
    r = R.log_on_mel(R.minmax(R.magphase_to_mel(80)(R.complex_to_magphase(r))))
    assert tuple(feats.shape) == r.shape == (80, 157, 2)
    assert np.abs(np.exp(feats.cpu().numpy()) - np.exp(r)).max() <= 1e-5
    torch.manual_seed(0)
    model = S.get_model(cfg).to(dev)
    out = I.evaluate_wav(model, wav, cfg, 16000, overlap_hop=32, device=dev)
    assert tuple(out.shape) == (157, 3) and set(np.unique(out.cpu().numpy())) <= {0.0, 1.0}
    # the inference engine stands in for the model: same averaged predictions before the threshold (stereo input: the
    # first layer runs the 2-channel stencil)
    model = model.to(memory_format=torch.channels_last)
    eng = S.InferenceEngine(model)
    assert eng.fused_lstm and eng.fused_convs == 14
    plain = torch.nn.Sequential(model)      # no `.predict`: the literal module in eval mode, not the model's own cached engine
    for smoothing in (False, True):
        pa = I.predict_frames(plain, feats, cfg, overlap_hop=32, smoothing=smoothing, threshold=False)
        pb = I.predict_frames(eng, feats, cfg, overlap_hop=32, smoothing=smoothing, threshold=False)
        pc = I.predict_frames(model, feats, cfg, overlap_hop=32, smoothing=smoothing, threshold=False)   # model.predict
        assert float((pa - pb).abs().max()) <= 1e-4 and float((pa - pc).abs().max()) <= 1e-4   # BEFORE the threshold
        a = I.predict_frames(plain, feats, cfg, overlap_hop=32, smoothing=smoothing)
        b = I.predict_frames(eng, feats, cfg, overlap_hop=32, smoothing=smoothing)
        flipped = a != b
        assert bool(((pa - 0.5).abs()[flipped] <= 1e-4).all())   # a frame may only flip if it sits on the threshold itself


def test_device_mixer_matches_oracle(dev):
    """Batched on-device sample synthesis (iris_mix_specs) == merge_complex_specs of the oracle
    (pipeline.py:6-110) sample by sample, bit for bit: ragged sources, padded groups, tiled
    backgrounds, silent voice frames, overlapping labels (dropped voices), noises."""
    from challenge_amd.mixer import DeviceMixer
    rng = np.random.default_rng(11)
    F, C2, n_frame, n_classes = 33, 4, 48, 3

    def clip(t, silent_from=None):
        x = rng.standard_normal((F, t, C2)).astype(np.float32)
        if silent_from is not None:
            x[:, silent_from:] = -np.abs(x[:, silent_from:])  # max <= 0: inactive frames
        return x
    backgrounds = [clip(t) for t in (20, 70, 48)]
    voices = [clip(t, s) for t, s in ((30, 20), (55, None), (41, 5), (64, 50), (25, None), (48, 30), (36, None))]
    labels = np.eye(n_classes, dtype=np.float32)[rng.integers(0, n_classes, len(voices))]
    noises = [clip(t) for t in (18, 90, 40, 52)]
    mixer = DeviceMixer(backgrounds, voices, labels, noises, n_frame=n_frame, max_voices=4, max_noises=3,
                        n_classes=n_classes, device=dev, min_ratio=1, seed=5)
    draws = mixer.draw(16)
    spec, lab = mixer.mix(16, draws)
    assert spec.shape == (16, F, n_frame, C2) and lab.shape == (16, 4, n_frame, n_classes)
    spec, lab = spec.cpu().numpy(), lab.cpu().numpy()
    dropped = 0
    for i, d in enumerate(draws):
        def padded(bank, idx, length):
            out = np.zeros((len(idx), F, length, C2), np.float32)
            for j, k in enumerate(idx):
                out[j, :, :bank[k].shape[1]] = bank[k]
            return out
        v = padded(voices, d["voices"], d["v_len"])
        n = padded(noises, d["noises"], d["n_len"])
        ref_spec, ref_lab = R.merge_complex_specs_apply(backgrounds[d["bg"]], v, labels[d["voices"]], n, d,
                                                        n_frame=n_frame, n_classes=n_classes, min_ratio=1)
        assert np.array_equal(spec[i], ref_spec), i
        assert np.array_equal(lab[i], ref_lab), i
        dropped += int(d["n_voices"] - (ref_lab.max(axis=(1, 2)) > 0).sum())
    assert dropped > 0  # the overlap rule was exercised
    # a second batch continues the shuffled streams; no min_ratio override -> padding path
    mixer2 = DeviceMixer(backgrounds, voices, labels, None, n_frame=n_frame, max_voices=3, n_classes=n_classes,
                         device=dev, seed=6)
    d2 = mixer2.draw(5)
    s2, l2 = mixer2.mix(5, d2)
    for i, d in enumerate(d2):
        v = np.zeros((3, F, d["v_len"], C2), np.float32)
        for j, k in enumerate(d["voices"]):
            v[j, :, :voices[k].shape[1]] = voices[k]
        ref_spec, ref_lab = R.merge_complex_specs_apply(backgrounds[d["bg"]], v, labels[d["voices"]], None, d,
                                                        n_frame=n_frame, n_classes=n_classes)
        assert np.array_equal(s2[i].cpu().numpy(), ref_spec) and np.array_equal(l2[i].cpu().numpy(), ref_lab)


@pytest.mark.parametrize("n_chan,name", [(2, "run"), (2, "run_filter"), (3, "run_nominmax")])
def test_device_dataset_equals_per_sample_stages(dev, n_chan, name):
    """make_device_dataset == the per-sample stage order of make_dataset (sj_train.py:103-129) on the
    same mixed samples and the same mask draws: bands handed to the mel kernel instead of multiplied into
    the complex batch, channel maps applied to the batch."""
    T, D, S = mods()
    from challenge_amd.mixer import DeviceMixer
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '40', '--n_frame', '64', '--n_chan', str(n_chan), '--batch_size', '6',
                        '--max_voices', '4', '--max_noises', '3', '--name', name])
    backgrounds, voices, labels, noises = S.synthetic_sources(2, 3, freq=257, n_bg=3, n_voice=7, n_noise=4, seed=3)
    onehot = np.eye(3, dtype=np.float32)[np.asarray(labels)]
    mixer = DeviceMixer(backgrounds, voices, onehot, noises, n_frame=64, max_voices=4, max_noises=3, n_classes=3,
                        device=dev, min_ratio=1, seed=9)
    spec, lab = mixer.mix(6)
    rng = np.random.default_rng(1)
    draws = [D.augment_draw(64, 257, rng) for _ in range(6)]
    tb, fb = np.stack([d[0] for d in draws]), np.stack([d[1] for d in draws])
    k = int(round(200 / (16000 / 256)))
    chan_map = {1: D.mono_chan, 2: None, 3: D.stereo_mono}[n_chan]
    # reference order, sample by sample: to_frame_labels, augment, channel map, stft_filter, batch
    xs, ys = [], []
    for i in range(6):
        x, y = D.to_frame_labels(spec[i], lab[i])
        x = T.mask_apply(T.mask_apply(x, -2, tb[i]), -3, fb[i])
        if chan_map is not None:
            x, y = chan_map(x, y)
        if 'filter' in name:
            x, y = D.stft_filter(k)(x, y)
        xs.append(x)
        ys.append(y)
    to_mel = S.complex_to_mel(40, 257)
    ref_x, ref_y = to_mel(torch.stack(xs), torch.stack(ys))
    # device order: batch first, bands into the mel kernel
    x, y = D.to_frame_labels(spec, lab)
    if chan_map is not None:
        x, y = chan_map(x, y)
    fbb = fb if 'filter' not in name else np.concatenate([fb, np.tile(np.array([[[1, k]]], np.int32), (6, 1, 1))], 1)
    out_x, out_y = to_mel(x, y, t_bands=tb, f_bands=fbb)
    assert torch.equal(out_x, ref_x) and torch.equal(out_y, ref_y)
    # and the assembled dataset: shapes / value ranges of the reference graph, batch after batch
    ds = S.make_device_dataset(cfg, training=True, sources=(backgrounds, voices, labels, noises), device=dev, seed=4)
    it = iter(ds)
    for _ in range(3):
        bx, by = next(it)
        assert bx.shape == (6, 40, 64, {1: 1, 2: 2, 3: 3}[n_chan]) and by.shape == (6, 2, 3)
        assert torch.isfinite(bx).all() and float(by.min()) >= 0 and float(by.max()) <= 1
        if 'nominmax' not in name:
            assert float(bx.max()) <= 1e-6 and float(bx.min()) >= np.log(1e-8) - 1e-3


def test_wave_mixer_matches_oracle_and_the_spectrum_domain_mixer(dev):
    """Waveform-domain batched synthesis (iris_mix_waves; SURVEY.md section 8 (f) rank 1, second half):
    bit-exact against the oracle's waveform-domain restatement (ragged sources, padded groups, tiled
    backgrounds, silent tails, dropped voices), and - the STFT being linear - equal to the spectrum-domain
    mixer run on the sources' STFTs on every frame whose window crosses no crop / pad / tiling boundary."""
    from challenge_amd.mixer import DeviceMixer, WaveMixer
    FE = __import__("challenge_amd.frontend", fromlist=["FrontendPlan"])
    rng = np.random.default_rng(13)
    C, hop, n_fft, n_frame, n_classes = 2, 64, 256, 48, 3

    def clip(n, silent_from=None):
        x = (rng.standard_normal((C, n)) * 0.3).astype(np.float32)
        if silent_from is not None:
            x[:, silent_from:] = 0
        return x
    backgrounds = [clip(n) for n in (hop * 20 + 7, hop * 90, hop * 48 + 33)]
    voices = [clip(n, s) for n, s in ((hop * 30, hop * 20), (hop * 55 + 5, None), (hop * 41, hop * 5), (hop * 64, hop * 50),
                                      (hop * 25 + 60, None), (hop * 48, hop * 30), (hop * 36, None))]
    labels = np.eye(n_classes, dtype=np.float32)[rng.integers(0, n_classes, len(voices))]
    noises = [clip(n) for n in (hop * 18, hop * 90 + 9, hop * 40, hop * 52)]
    mixer = WaveMixer(backgrounds, voices, labels, noises, n_frame=n_frame, n_fft=n_fft, hop=hop, max_voices=4,
                      max_noises=3, n_classes=n_classes, device=dev, min_ratio=1, seed=5)
    draws = mixer.draw(16)
    wav, lab = mixer.mix(16, draws)
    assert wav.shape == (16, C, (n_frame - 1) * hop) and lab.shape == (16, 4, n_frame, n_classes)
    wav_h, lab_h = wav.cpu().numpy(), lab.cpu().numpy()
    dropped = 0
    for i, d in enumerate(draws):
        ref_wav, ref_lab = R.mix_waves_apply(backgrounds[d["bg"]], [voices[k] for k in d["voices"]], labels[d["voices"]],
                                             [noises[k] for k in d["noises"]], d, n_frame=n_frame, n_classes=n_classes,
                                             hop=hop, n_fft=n_fft, min_ratio=1)
        assert np.array_equal(wav_h[i], ref_wav), i
        assert np.array_equal(lab_h[i], ref_lab), i
        dropped += int(d["n_voices"] - (ref_lab.max(axis=(1, 2)) > 0).sum())
    assert dropped > 0
    # activity flags of the waveform sources == the oracle's
    for v, act in zip(voices, mixer.voice_active):
        assert np.array_equal(act.cpu().numpy(), R.wave_frame_active(v, n_fft, hop))
    # linearity: the same draws through the spectrum-domain mixer on the sources' STFTs (frame counts agree: 1 + L // hop)
    to_spec = lambda w: R.to_ref_layout(R.stft(w, n_fft, hop))  # noqa: E731
    smix = DeviceMixer([to_spec(b) for b in backgrounds], [to_spec(v) for v in voices], labels,
                       [to_spec(n) for n in noises], n_frame=n_frame, max_voices=4, max_noises=3, n_classes=n_classes,
                       device=dev, min_ratio=1, seed=5)
    spec, slab = smix.mix(16, draws)
    assert torch.equal(slab, lab)                      # same labels: same activity, same overlap rule
    plan = FE.FrontendPlan(n_fft, hop, 20, 16000, C, 16, (n_frame - 1) * hop, dev)
    got = plan.stft(wav).cpu().numpy()                 # [B, F, n_frame, 2C]
    spec = spec.cpu().numpy()
    checked = 0
    for i, d in enumerate(draws):
        t_bg = 1 + backgrounds[d["bg"]].shape[1] // hop
        edges = [0, n_frame - 1] + [k * t_bg - d["bg_offset"] for k in range(1, 6)] + [k * t_bg - d["bg_offset"] - 1 for k in range(1, 6)]
        pad_v = max(n_frame - int(d["v_len"]), 0)
        for j in range(d["n_voices"]):
            tv = 1 + voices[d["voices"][j]].shape[1] // hop
            edges += [pad_v - d["v_offset"][j], pad_v - d["v_offset"][j] + tv - 1]
        pad_n = max(n_frame - int(np.float32(0.5) * np.float32(d["n_len"])), 0)
        for j in range(d["n_noises"]):
            tn = 1 + noises[d["noises"][j]].shape[1] // hop
            edges += [pad_n - d["n_offset"][j], pad_n - d["n_offset"][j] + tn - 1]
        interior = np.array([all(abs(t - e) > 2 for e in edges) for t in range(n_frame)])
        if t_bg < n_frame + d["bg_offset"]:            # a tiled background drifts (period t_bg frames vs L samples): skip
            continue
        checked += int(interior.sum())
        scale = np.abs(spec[i]).max()
        assert np.abs(got[i][:, interior] - spec[i][:, interior]).max() <= 3e-5 * scale, i
    assert checked >= 40


def test_make_wave_dataset_matches_oracle_chain(dev):
    """sj_train.make_wave_dataset: waveforms resident on the device -> WaveMixer -> fused kernel with SpecAugment /
    filter bands -> log-mel, against the oracle chain (mix_waves_apply is checked above; here mixer output ->
    wav_to_logmel with the same bands), for the stereo and the mono-sum configurations."""
    T, D, S = mods()
    from challenge_amd.mixer import WaveMixer
    for n_chan, name in ((2, ''), (1, 'filter')):
        cfg = S.ARGS().get(['--v', '9', '--n_mels', '40', '--n_frame', '64', '--n_chan', str(n_chan), '--batch_size', '5',
                            '--max_voices', '4', '--max_noises', '3', '--name', name])
        src = S.synthetic_wave_sources(2, 3, 256, n_bg=3, n_voice=7, n_noise=4, seed=3)
        ds = iter(S.make_wave_dataset(cfg, training=True, sources=src, device=dev, seed=9))
        x, y = next(ds)
        assert tuple(x.shape) == (5, 40, 64, n_chan) and tuple(y.shape) == (5, 2, 3) and x.is_cuda
        # replay: same seeds -> same mix and the same band draws
        onehot = np.eye(3, dtype=np.float32)[np.asarray(src[2])]
        mixer = WaveMixer(src[0], src[1], onehot, src[3], n_frame=64, n_fft=512, hop=256, max_voices=4, max_noises=3,
                          n_classes=3, device=dev, min_ratio=1, seed=9)
        wav, lab = mixer.mix(5)
        wav = wav.cpu().numpy()
        if n_chan == 1:
            wav = wav[:, :1] + wav[:, 1:]
        tb, fb = D.augment_draw_batch(5, 64, 257, np.random.default_rng(10))
        if name == 'filter':
            k = int(round(200 / (16000 / 256)))
            fb = np.concatenate([fb, np.tile(np.array([[[1, k]]], np.int32), (5, 1, 1))], axis=1)
        ref = R.wav_to_logmel(wav, 512, 256, 40, 16000, t_bands=tb, f_bands=fb)
        assert np.abs(np.exp(x.cpu().numpy()) - np.exp(ref)).max() <= 5e-6
        ref_y = R.label_downsample(32)(None, R.to_frame_labels(None, lab.cpu().numpy())[1])[1]
        assert np.array_equal(y.cpu().numpy(), ref_y)


def test_drop_in_merge_on_device_matches_oracle(dev):
    """The per-sample drop-in `merge_complex_specs_apply` on device tensors (HIP path) == the oracle,
    bit for bit, and == its own op-by-op torch form on the CPU."""
    from challenge_amd import pipeline as P
    rng = np.random.default_rng(21)
    for s in range(6):
        bg = rng.standard_normal((17, 8 + s, 4)).astype(np.float32)
        voices = rng.standard_normal((4, 17, 10 + 3 * s, 4)).astype(np.float32)
        voices[1, :, 6:] = -np.abs(voices[1, :, 6:])
        labels = np.eye(5, dtype=np.float32)[rng.integers(0, 5, 4)]
        noises = rng.standard_normal((3, 17, 9 + 2 * s, 4)).astype(np.float32) if s % 2 == 0 else None
        d = P.merge_draw(8 + s, [10 + 3 * s] * 4, None if noises is None else [9 + 2 * s] * 3, n_frame=12,
                         rng=np.random.default_rng(s))
        t = lambda x: None if x is None else torch.from_numpy(x).to(dev)  # noqa: E731
        a, la = P.merge_complex_specs_apply(t(bg), t(voices), t(labels), t(noises), d, n_frame=12, n_classes=5)
        b, lb = R.merge_complex_specs_apply(bg, voices, labels, noises, d, n_frame=12, n_classes=5)
        assert np.array_equal(a.cpu().numpy(), b) and np.array_equal(la.cpu().numpy(), lb)
        c, lc = P.merge_complex_specs_apply(torch.from_numpy(bg), torch.from_numpy(voices), torch.from_numpy(labels),
                                            None if noises is None else torch.from_numpy(noises), d, n_frame=12,
                                            n_classes=5)
        assert np.array_equal(c.numpy(), b) and np.array_equal(lc.numpy(), lb)
        # seperate_noise_voice (pipeline.py:38-39, :80-81, :104-108): the same kernels over three source tables
        a, (la, ov, on) = P.merge_complex_specs_apply(t(bg), t(voices), t(labels), t(noises), d, n_frame=12,
                                                      n_classes=5, seperate_noise_voice=True)
        b, (lb, bov, bon) = R.merge_complex_specs_apply(bg, voices, labels, noises, d, n_frame=12, n_classes=5,
                                                        seperate_noise_voice=True)
        assert ov.is_cuda and np.array_equal(a.cpu().numpy(), b) and np.array_equal(la.cpu().numpy(), lb)
        assert np.array_equal(ov.cpu().numpy(), bov) and np.array_equal(on.cpu().numpy(), bon)


# ---------------------------------------------------------------------------
# round 2: value tests on the device for the rows that only had shape / CPU coverage
# ---------------------------------------------------------------------------
def test_label_helpers_on_device_match_oracle(dev):
    """R9 on the device against the oracle (data_utils.py:64-70, :85-97, :120-123; trainer.py:86-104)."""
    _, D, _ = mods()
    from challenge_amd import trainer as TR
    rng = np.random.default_rng(31)
    y4 = (rng.random((5, 4, 70, 3)) > 0.7).astype(np.float32)          # [B, voices, frames, classes]
    yd = torch.from_numpy(y4).to(dev)
    frame = D.to_frame_labels(None, yd)[1]
    assert frame.is_cuda and np.array_equal(frame.cpu().numpy(), R.to_frame_labels(None, y4)[1])
    y3 = R.to_frame_labels(None, y4)[1]                                # [B, frames, classes], ragged tail (70 = 2*32 + 6)
    for res in (32, 8, 7):
        got = D.label_downsample(res)(None, torch.from_numpy(y3).to(dev))[1]
        assert got.is_cuda and np.array_equal(got.cpu().numpy(), R.label_downsample(res)(None, y3)[1])
    ylong = (rng.random((64, 512, 3)) > 0.5).astype(np.float32)        # c3 / c4: batch 64 keeps all 64 rows
    got = D.label_downsample(32)(None, torch.from_numpy(ylong).to(dev))[1].cpu().numpy()
    assert got.shape == (64, 16, 3) and np.array_equal(got, R.label_downsample(32)(None, ylong)[1])
    for mult in (1, 10):
        got = TR.preprocess_labels(mult)(None, torch.from_numpy(y3).to(dev))[1].cpu().numpy()
        ref = R.preprocess_labels(mult)(None, y3)[1]
        assert got.shape == ref.shape == (5, 3, 3) and np.allclose(got, ref, rtol=1e-6, atol=0)
    dens = TR.to_density_labels(None, yd)[1].cpu().numpy()
    assert np.allclose(dens, R.to_density_labels(None, y4)[1], rtol=1e-6, atol=1e-9)
    zero = torch.zeros(2, 3, 8, 3, device=dev)                         # a silent voice: safe_div keeps it at 0
    assert float(TR.to_density_labels(None, zero)[1].abs().max()) == 0.0
    assert np.array_equal(D.multiply_label(3)(None, yd)[1].cpu().numpy(), R.multiply_label(3)(None, y4)[1])


def test_phase_vocoder_values_on_device(dev, kats):
    """transforms.py:137-195 on the device, values (not only shapes): fp64 against the oracle at the rates of
    transforms_test.py:98-108; the fp32 run carries the accumulated-phase noise (~1e3 rad) the reference has too."""
    T, _, _ = mods()
    k = kats["phase_vocoder_shapes"]
    rng = np.random.default_rng(41)
    spec = rng.standard_normal((k["n_freq"], k["time"], k["chan2"]))
    sd = torch.from_numpy(spec).to(dev)
    assert T.phase_vocoder(sd, 1.0) is sd
    for rate in k["rates"]:
        got = T.phase_vocoder(sd, rate=rate)
        ref = R.phase_vocoder(spec, rate)
        assert got.is_cuda and got.dtype == torch.float64
        assert list(got.shape) == list(ref.shape) == [k["n_freq"], int(np.ceil(k["time"] / rate)), k["chan2"]]
        assert np.abs(got.cpu().numpy() - ref).max() <= 1e-9 * np.abs(ref).max()
        got32 = T.phase_vocoder(sd.float(), rate=rate).cpu().numpy()
        # fp32: the accumulated phase reaches ~7e4 rad here (257 bins x ~100 steps), so fp32 rounding alone is worth
        # ~1e-2 rad per element - in the reference's fp32 TF graph too; the fp64 run above checks the algorithm
        assert got32.dtype == np.float32 and np.abs(got32 - ref).max() <= 5e-2 * np.abs(ref).max()
    # the two magnitude-phase helpers that stay torch ops: values on the device against the oracle
    mp = rng.standard_normal((5, 10, 7, 4))
    assert np.allclose(T.minmax_norm_magphase(torch.from_numpy(mp).to(dev)).cpu().numpy(), R.minmax_norm_magphase(mp),
                       rtol=1e-12, atol=1e-12)
    lm = np.abs(rng.standard_normal((6, 9, 6)))
    assert np.allclose(T.log_magphase(torch.from_numpy(lm).to(dev), n_chan=3).cpu().numpy(), R.log_magphase(lm, n_chan=3),
                       rtol=1e-12, atol=1e-12)


def test_c3_batch64_device_bands_match_oracle(dev):
    """BASELINE configs[2] at its full size: batch 64 x 130,816 samples (T = 512), SpecAugment bands drawn on the
    device (6 time + 1 frequency band per clip) plus stft_filter(3), through the fused kernel; a sample of clips is
    checked against the fp64 oracle (mel before min-max: the stated rule of oracle.mel_tolerance) and the fp32 oracle (after min-max / log)."""
    _, _, S = mods()
    b, length, n_t = 64, 130816, 512
    fe = S.WaveFrontend(1024, 256, 64, 16000, 1, b, length, dev, training=True, device_draw=True, filter_bins=3, seed=3)
    gen = torch.Generator(device=dev).manual_seed(17)
    wav = torch.randn(b, 1, length, generator=gen, device=dev) * 0.1
    tb, fb = fe.draw_bands_device(b, n_t)
    fb = torch.cat([fb, torch.tensor([[[1, 3]]], dtype=torch.int32, device=dev).expand(b, 1, 2)], dim=1)
    raw = fe.plan.wav_to_logmel(wav, minmax=False, log=False, t_bands=tb, f_bands=fb)
    out = fe.plan.wav_to_logmel(wav, t_bands=tb, f_bands=fb)
    assert tuple(out.shape) == (b, 64, n_t, 1) and torch.isfinite(out).all()
    idx = [0, 21, 42, 63]
    w, tbn, fbn = wav[idx].cpu().numpy(), tb[idx].cpu().numpy(), fb[idx].cpu().numpy()
    ref64, tol = R.mel_tolerance(w, 1024, 256, 64, 16000, t_bands=tbn, f_bands=fbn)
    got = raw[idx].cpu().numpy()
    assert R.mel_err_ratio(got, ref64, tol) <= 1.0  # the stated rule: 1e-5 |ref| + 4 eps xrms wsum (oracle.mel_tolerance)
    # masked frames / bins are exactly zero columns; the per-clip minimum is therefore 0 wherever a time band has size > 0
    for j, i in enumerate(idx):
        for off, size in tbn[j]:
            assert not got[j, :, off:off + size].any()
    ref = R.wav_to_logmel(w, 1024, 256, 64, 16000, t_bands=tbn, f_bands=fbn)
    assert np.abs(np.exp(out[idx].cpu().numpy()) - np.exp(ref)).max() <= 5e-6
    # the same call through the WaveFrontend object (its own draws): shape and range only
    x = fe(wav)
    assert tuple(x.shape) == (b, 64, n_t, 1) and float(x.max()) <= 1e-6 and float(x.min()) >= math.log(1e-8) - 1e-4


def test_sj_train_main_two_epochs(dev, tmp_path, monkeypatch):
    """sj_train.main on synthetic sources for two epochs (sj_train.py:406-525): CSV log with one row per epoch
    (:490), best-val checkpoint (:492), SWA weights (:491, :521), the run-name encoding (:416-429)."""
    import csv
    _, _, S = mods()
    monkeypatch.chdir(tmp_path)
    S.main(['--synthetic', '--epochs', '2', '--steps_per_epoch', '3', '--validation_steps', '1', '--batch_size', '8',
            '--n_frame', '128', '--v', '9', '--n_mels', '32', '--name', 'pytest'])
    stem = 'pytest_vad_v9_lr0.001_batch8_opt_adam_mel32_chan2_BCE_framelen128'
    with open(tmp_path / (stem + '.csv')) as f:
        rows = list(csv.DictReader(f))
    assert [int(r['epoch']) for r in rows] == [0, 1]
    assert all(math.isfinite(float(r['loss'])) and math.isfinite(float(r['val_loss'])) and float(r['lr']) > 0 for r in rows)
    ckpt = torch.load(tmp_path / (stem + '.pt'), map_location='cpu')
    swa = torch.load(tmp_path / (stem + '_SWA.pt'), map_location='cpu')
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '128'])
    model = S.get_model(cfg)
    model.load_state_dict(ckpt)
    model.load_state_dict(swa)
    assert set(ckpt) == set(swa) == set(model.state_dict())
    # the same entry point fed from waveform corpora mixed before the STFT (make_wave_dataset): one epoch
    S.main(['--online_stft', '--epochs', '1', '--steps_per_epoch', '2', '--validation_steps', '1', '--batch_size', '8',
            '--n_frame', '128', '--v', '9', '--n_mels', '32', '--name', 'pywave'])
    with open(tmp_path / 'pywave_vad_v9_lr0.001_batch8_opt_adam_mel32_chan2_BCE_framelen128.csv') as f:
        rows = list(csv.DictReader(f))
    assert len(rows) == 1 and math.isfinite(float(rows[0]['loss'])) and math.isfinite(float(rows[0]['val_loss']))
    # (the runs above trained on a replayed hipGraph - the default on one GPU with Adam: fit builds a GraphedTrainStep on the first
    # batch, the scheduler writes the learning rate into the optimiser's device tensor.)  The same with IRIS_GRAPH_STEP=0: eager
    assert S.GRAPH_STEP
    monkeypatch.setattr(S, 'GRAPH_STEP', False)
    S.main(['--synthetic', '--epochs', '3', '--steps_per_epoch', '4', '--validation_steps', '1', '--batch_size', '8',
            '--n_frame', '128', '--v', '9', '--n_mels', '32', '--name', 'pygraph'])
    with open(tmp_path / 'pygraph_vad_v9_lr0.001_batch8_opt_adam_mel32_chan2_BCE_framelen128.csv') as f:
        rows = list(csv.DictReader(f))
    assert [int(r['epoch']) for r in rows] == [0, 1, 2]
    assert all(math.isfinite(float(r['loss'])) and math.isfinite(float(r['val_loss'])) for r in rows)
    lrs = [float(r['lr']) for r in rows]
    assert lrs[0] > 0 and len(set(lrs)) == 3          # custom_scheduler's per-epoch rate
    assert all(0.0 < float(r['loss']) < 2.0 for r in rows)   # (rate 0.008 on 12 random batches: the loss wanders, it must not blow up)


def test_device_side_draws_match_the_oracle_restatement(dev):
    """iris_mix_draw / iris_augment_draw (the random half of a batch drawn on the device, k_draw.h): every integer the
    kernels write - sources picked from the shuffled streams, crop offsets, voice / noise counts, pads, band offsets and
    sizes - equals the oracle's NumPy restatement of the Philox generator bit for bit over several consecutive calls
    (the call counter and stream positions live on the device); gains agree to the rounding of 10^x; the batch mixed
    from those records equals the oracle's apply bit for bit, in both the spectrum and the waveform domain."""
    from challenge_amd.mixer import DeviceMixer, WaveMixer, KIND_VOICE, KIND_NOISE, KIND_UNUSED
    from challenge_amd.data_utils import DeviceAugmentDraw
    rng = np.random.default_rng(31)
    C, hop, n_fft, n_frame, n_classes, V, Nn, B, seed = 2, 64, 256, 48, 3, 4, 3, 16, 0x1234567890ABCDEF

    def clip(n, silent_from=None):
        x = (rng.standard_normal((C, n)) * 0.3).astype(np.float32)
        if silent_from is not None:
            x[:, silent_from:] = 0
        return x
    backgrounds = [clip(n) for n in (hop * 20 + 7, hop * 90, hop * 48 + 33)]
    voices = [clip(n, s) for n, s in ((hop * 30, hop * 20), (hop * 55 + 5, None), (hop * 41, hop * 5), (hop * 64, hop * 50),
                                      (hop * 25 + 60, None), (hop * 48, hop * 30), (hop * 36, None))]
    labels = np.eye(n_classes, dtype=np.float32)[rng.integers(0, n_classes, len(voices))]
    noises = [clip(n) for n in (hop * 18, hop * 90 + 9, hop * 40, hop * 52)]
    mixer = WaveMixer(backgrounds, voices, labels, noises, n_frame=n_frame, n_fft=n_fft, hop=hop, max_voices=V,
                      max_noises=Nn, n_classes=n_classes, device=dev, min_ratio=1, seed=5)
    mixer.enable_device_draw(seed)
    state = [0, 0, 0, 0]
    epochs_v = []
    for call in range(4):
        wav, lab = mixer.mix(B)
        table = mixer.last_table(B)
        want = R.mix_draw_device(mixer._bg_T, mixer._v_T, mixer._n_T, B, n_frame, V, Nn, 1.0, 0.5, -20.0, seed, state)
        draws = mixer.table_to_draws(table)
        for i, (w, d) in enumerate(zip(want, draws)):
            assert (d["bg"], d["bg_offset"], d["n_voices"], d["n_noises"]) == (w["bg"], w["bg_offset"], w["n_voices"], w["n_noises"])
            assert d["voices"] == w["voices"] and d["noises"] == w["noises"]
            assert d["v_offset"] == w["v_offset_all"][:w["n_voices"]] and d["n_offset"] == w["n_offset_all"][:w["n_noises"]]
            assert list(table[i]["off"][1:1 + V]) == w["v_offset_all"] and list(table[i]["off"][1 + V:]) == w["n_offset_all"]
            assert np.all(table[i]["pad"][1:1 + V] == w["v_pad"]) and np.all(table[i]["pad"][1 + V:] == w["n_pad"])
            assert np.allclose(table[i]["gain"][1:1 + V], w["v_gain_all"], rtol=2e-6)
            assert np.allclose(table[i]["gain"][1 + V:], w["n_gain_all"], rtol=2e-6)
            assert list(table[i]["kind"]) == [0] + [KIND_VOICE] * w["n_voices"] + [KIND_UNUSED] * (V - w["n_voices"]) + \
                [KIND_NOISE] * w["n_noises"] + [KIND_UNUSED] * (Nn - w["n_noises"])
            assert 1 <= d["n_voices"] <= V - 1 and 0 <= d["n_noises"] <= Nn - 1
        epochs_v += [k for d in draws for k in d["voices"]]
        wav_h, lab_h = wav.cpu().numpy(), lab.cpu().numpy()
        for i, d in enumerate(draws):
            ref_wav, ref_lab = R.mix_waves_apply(backgrounds[d["bg"]], [voices[k] for k in d["voices"]], labels[d["voices"]],
                                                 [noises[k] for k in d["noises"]], d, n_frame=n_frame, n_classes=n_classes,
                                                 hop=hop, n_fft=n_fft, min_ratio=1)
            assert np.array_equal(wav_h[i], ref_wav) and np.array_equal(lab_h[i], ref_lab), (call, i)
    assert state[0] == 4 and mixer._dd["state"].cpu().tolist() == state
    nv = len(voices)                                   # shuffled, repeated stream: every epoch is a permutation
    for e in range(len(epochs_v) // nv):
        assert sorted(epochs_v[e * nv:(e + 1) * nv]) == list(range(nv))
    # the spectrum-domain mixer takes the same kernel (no `len` column)
    to_spec = lambda w: R.to_ref_layout(R.stft(w, n_fft, hop))  # noqa: E731
    smix = DeviceMixer([to_spec(b) for b in backgrounds], [to_spec(v) for v in voices], labels, [to_spec(n) for n in noises],
                       n_frame=n_frame, max_voices=V, max_noises=Nn, n_classes=n_classes, device=dev, min_ratio=1, seed=5)
    smix.enable_device_draw(seed)
    spec, slab = smix.mix(B)
    st2 = [0, 0, 0, 0]
    want = R.mix_draw_device(smix._bg_T, smix._v_T, smix._n_T, B, n_frame, V, Nn, 1.0, 0.5, -20.0, seed, st2)
    d0 = smix.table_to_draws(smix.last_table(B))
    assert [d["voices"] for d in d0] == [w["voices"] for w in want] and [d["bg_offset"] for d in d0] == [w["bg_offset"] for w in want]
    F = n_fft // 2 + 1
    for i, d in enumerate(d0[:6]):
        def padded(bank, idx, length):
            out = np.zeros((len(idx), F, length, 2 * C), np.float32)
            for j, k in enumerate(idx):
                sp = to_spec(bank[k])
                out[j, :, :sp.shape[1]] = sp
            return out
        ref_spec, ref_lab = R.merge_complex_specs_apply(to_spec(backgrounds[d["bg"]]), padded(voices, d["voices"], d["v_len"]),
                                                        labels[d["voices"]], padded(noises, d["noises"], d["n_len"]), d,
                                                        n_frame=n_frame, n_classes=n_classes, min_ratio=1)
        assert np.array_equal(spec[i].cpu().numpy(), ref_spec) and np.array_equal(slab[i].cpu().numpy(), ref_lab)
    # SpecAugment bands: exact integers, and the reference's support (the last index is never masked when size > 0)
    aug = DeviceAugmentDraw(dev, seed=77, filter_bins=3)
    ast = [0]
    for _ in range(3):
        tb, fb = aug(B, 512, 257)
        wt, wf = R.augment_draw_device(B, 512, 6, 24, 257, 1, 16, 77, ast)
        assert np.array_equal(tb.cpu().numpy(), wt) and np.array_equal(fb[:, :1].cpu().numpy(), wf)
        assert np.array_equal(fb[:, 1].cpu().numpy(), np.tile(np.array([1, 3], np.int32), (B, 1)))
        t = tb.cpu().numpy()
        assert t[..., 1].max() <= 23 and np.all(t[..., 0] + t[..., 1] <= 511) and t.min() >= 0
    with pytest.raises(ValueError):
        N = __import__("challenge_amd._native", fromlist=["x"])
        N.check(N.lib().iris_augment_draw(4, 10, 6, 24, 257, 1, 16, 0, aug.state.data_ptr(), tb.data_ptr(), fb.data_ptr(), None),
                "iris_augment_draw")                   # a 24-frame mask on a 10-frame axis: the reference errors too


def test_device_drawn_datasets_run_and_are_reproducible(dev):
    """make_wave_dataset / make_device_dataset with device_draw=True: shapes as the host-drawn datasets, finite values,
    same seed -> same batches, different seed -> different batches."""
    from challenge_amd import sj_train as S
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '40', '--n_frame', '64', '--n_chan', '2', '--batch_size', '6', '--name', 'r_filter'])
    srcs = S.synthetic_wave_sources(2, 3, 256, n_bg=4, n_voice=9, n_noise=5, seed=2)

    def first(n, seed, fn, sources, **kw):
        it = iter(fn(cfg, True, sources=sources, device=dev, seed=seed, device_draw=True, **kw))
        return [next(it) for _ in range(n)]
    a, b, c = first(3, 7, S.make_wave_dataset, srcs), first(3, 7, S.make_wave_dataset, srcs), first(3, 8, S.make_wave_dataset, srcs)
    for (x, y), (x2, y2), (x3, _) in zip(a, b, c):
        assert tuple(x.shape) == (6, 40, 64, 2) and tuple(y.shape) == (6, 2, 3) and torch.isfinite(x).all()
        assert torch.equal(x, x2) and torch.equal(y, y2) and not torch.equal(x, x3)
    ssrc = S.synthetic_sources(2, 3, n_bg=4, n_voice=9, n_noise=5, seed=2)
    d = first(2, 3, S.make_device_dataset, ssrc)
    assert tuple(d[0][0].shape) == (6, 40, 64, 2) and torch.isfinite(d[0][0]).all() and not torch.equal(d[0][0], d[1][0])


def test_channel_helpers_on_device_match_oracle(dev):
    """R7 on the device against the oracle (data_utils.py:73-117): stereo_mono, mono_chan (both call forms, the
    broadcast-add quirk included), random_merge_aug through its deterministic half with the drawn factor replayed."""
    from challenge_amd import data_utils as D
    from challenge_amd import transforms as T
    rng = np.random.default_rng(21)
    x = rng.standard_normal((257, 37, 4)).astype(np.float32)
    xb = rng.standard_normal((3, 33, 20, 4)).astype(np.float32)
    for a in (x, xb):
        t = torch.from_numpy(a).to(dev)
        got = D.stereo_mono(t)
        assert got.is_cuda and np.array_equal(got.cpu().numpy(), R.stereo_mono(a))
        g2, y2 = D.stereo_mono(t, 5)
        assert y2 == 5 and torch.equal(g2, got)
        assert D.mono_chan(t) is t                                          # no labels: identity (data_utils.py:76)
        m, _ = D.mono_chan(t, 1)
        assert m.is_cuda and np.array_equal(m.cpu().numpy(), R.mono_chan(a, 1)[0]) and m.shape[-1] == 3
        m2, _ = D.mono_chan(t[..., :2], 1)                                  # 2-entry axis: a true down-mix
        assert np.array_equal(m2.cpu().numpy(), a[..., :1] + a[..., 1:2])
        for number in (3, 5):
            factor = rng.uniform(0.1, 0.9, size=(1, 1, number - 2))
            out = D.random_merge_aug_apply(t, number, factor)
            want = R.random_merge_aug_apply(a, number, factor.astype(np.float32))
            assert out.is_cuda and tuple(out.shape) == a.shape[:-1] + (2 * number,)
            assert np.abs(out.cpu().numpy() - want).max() <= 1e-6
        # the closure draws from transforms' generator: same seed, same factor as a host replay of the draw
        T.set_seed(11)
        out = D.random_merge_aug(4)(t)
        T.set_seed(11)
        factor = T.get_rng().uniform(0.1, 0.9, size=(1, 1, 2))
        assert np.abs(out.cpu().numpy() - R.random_merge_aug_apply(a, 4, factor.astype(np.float32))).max() <= 1e-6
    with pytest.raises(ValueError):
        D.random_merge_aug(5)(torch.zeros(2, 2, 6, device=dev))


def test_bias_relu_epilogues_match_torch(dev):
    """iris_bias_relu / iris_bias_relu_maxpool against the torch ops they replace (exact: one add, one max per element;
    MaxPool2d(2, 2, ceil_mode=True) = Keras 'same'), odd heights / widths included."""
    FEm = __import__("challenge_amd.frontend", fromlist=["x"])
    g = torch.Generator(device=dev).manual_seed(3)
    for b, c, h, w in [(2, 32, 64, 50), (3, 64, 5, 7), (1, 512, 2, 16), (2, 128, 1, 9)]:
        x = torch.randn(b, c, h, w, device=dev, generator=g).contiguous(memory_format=torch.channels_last)
        bias = torch.randn(c, device=dev, generator=g)
        want = torch.relu(x + bias.view(1, -1, 1, 1))
        pooled = torch.nn.functional.max_pool2d(want, 2, 2, ceil_mode=True)
        got_p = FEm.bias_relu_maxpool(x, bias)
        assert got_p.is_contiguous(memory_format=torch.channels_last) and torch.equal(got_p, pooled)
        got = FEm.bias_relu_(x.clone(memory_format=torch.channels_last), bias)
        assert torch.equal(got, want)
        xc = x.contiguous()                                  # the NCHW twins (block 1 of the inference engine)
        got_pn = FEm.bias_relu_maxpool_nchw(xc, bias)
        assert got_pn.is_contiguous(memory_format=torch.channels_last) and torch.equal(got_pn, pooled)
        if (h * w) % 4 == 0:
            assert torch.equal(FEm.bias_relu_nchw_(xc.clone(), bias), want)
    with pytest.raises(ValueError):
        FEm.bias_relu_maxpool(torch.zeros(1, 6, 4, 4, device=dev).contiguous(memory_format=torch.channels_last),
                              torch.zeros(6, device=dev))


def test_fused_bn_relu_training_matches_torch(dev, monkeypatch):
    """iris_bn_* behind _ConvBNReLU in training mode: outputs, every gradient (input, convolution weight, BatchNorm scale /
    shift) and the running statistics equal the stock torch / MIOpen ops on the same parameters (fp32, different summation
    orders only); the convolution bias gets an exact zero gradient (its true gradient is zero: BatchNorm removes the mean)."""
    import copy
    from challenge_amd import sj_train as S
    S.configure_miopen()
    torch.manual_seed(4)
    for cin, cout, b, h, w in [(1, 32, 3, 16, 40), (32, 64, 2, 9, 7), (64, 512, 2, 4, 5)]:
        blk = S._ConvBNReLU(cin, cout).to(dev).to(memory_format=torch.channels_last).train()
        with torch.no_grad():
            blk[1].weight.uniform_(0.5, 1.5)
            blk[1].bias.uniform_(-0.3, 0.3)
            blk[1].running_mean.uniform_(-0.2, 0.2)
            blk[1].running_var.uniform_(0.5, 1.5)
            blk[0].bias.uniform_(-0.5, 0.5)
        ref = copy.deepcopy(blk)
        x = torch.randn(b, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        g = torch.randn(b, cout, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        monkeypatch.setattr(S, "FUSED_BN_RELU", True)
        ya = blk(xa)
        ya.backward(g)
        monkeypatch.setattr(S, "FUSED_BN_RELU", False)
        yb = ref(xb)
        yb.backward(g)
        tol = lambda t: 2e-5 * float(t.abs().max()) + 1e-6  # noqa: E731
        assert float((ya - yb).abs().max()) <= tol(yb)
        assert float((xa.grad - xb.grad).abs().max()) <= tol(xb.grad) * 5
        assert float((blk[0].weight.grad - ref[0].weight.grad).abs().max()) <= tol(ref[0].weight.grad) * 5
        assert float((blk[1].weight.grad - ref[1].weight.grad).abs().max()) <= tol(ref[1].weight.grad) * 5
        assert float((blk[1].bias.grad - ref[1].bias.grad).abs().max()) <= tol(ref[1].bias.grad) * 5
        assert float(blk[0].bias.grad.abs().max()) == 0.0 and float(ref[0].bias.grad.abs().max()) <= 1e-4 * float(g.abs().sum())
        assert float((blk[1].running_mean - ref[1].running_mean).abs().max()) <= 1e-6
        assert float((blk[1].running_var - ref[1].running_var).abs().max()) <= 1e-6
        assert int(blk[1].num_batches_tracked) == int(ref[1].num_batches_tracked) == 1


def test_fused_bn_statistics_with_a_large_channel_offset(dev):
    """Advisor finding (round 3): E[z^2] - E[z]^2 on raw fp32 partial sums loses the variance of a channel whose |mean| is far
    above its spread (clamped to 0: rstd = 1 / sqrt(eps)).  The passes accumulate SHIFTED sums (z - z[row 0]); on
    z = offset + 0.1 randn with offsets up to 1000 the output, the gradient and the running estimates must equal an fp64
    evaluation of BatchNormalization + ReLU (+ the block's pooling) - the yardstick is fp64 because the stock fp32 op is
    itself offset-sensitive."""
    from challenge_amd import sj_train as S
    torch.manual_seed(11)
    b, c, h, w = 4, 32, 24, 40
    offs = torch.tensor([0.0, 1.0, -7.0, 100.0, -300.0, 1000.0, 31.0, 0.5] * 4, device=dev).view(1, c, 1, 1)
    z = (offs + 0.1 * torch.randn(b, c, h, w, device=dev)).contiguous(memory_format=torch.channels_last)
    gamma = torch.empty(c, device=dev).uniform_(0.5, 1.5)
    beta = torch.empty(c, device=dev).uniform_(-0.3, 0.3)
    for pool in (False, True):
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        za = z.clone().requires_grad_(True)
        ya = S._FusedBiasBNReLU.apply(za, None, gamma, beta, rm, rv, 1e-3, 0.01, pool)
        zb = z.double().requires_grad_(True)                    # fp64 reference of the same function
        mean = zb.mean(dim=(0, 2, 3), keepdim=True)
        var = zb.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        pre = (zb - mean) / torch.sqrt(var + 1e-3) * gamma.double().view(1, c, 1, 1) + beta.double().view(1, c, 1, 1)
        yb = torch.relu(pre)
        if pool:
            yb = torch.nn.functional.max_pool2d(yb, 2, 2, ceil_mode=True)
        # z itself is only known to 6e-5 at an offset of 1000 (fp32 spacing) = 6e-4 of its 0.1 spread: that is the floor
        assert float((ya.detach().double() - yb.detach()).abs().max()) <= 4e-3 * float(yb.detach().abs().max())
        if not pool:
            # gradient: wherever the ReLU decision is not within the fp32 resolution of z itself (|pre-activation| > 0.02; an
            # element on the threshold flips in ANY fp32 implementation and carries a whole gamma / sigma x g)
            g = torch.randn_like(ya)
            ya.backward(g)
            yb.backward(g.double())
            safe = pre.detach().abs() > 0.02
            assert float(safe.double().mean()) > 0.97
            assert float(((za.grad.double() - zb.grad).abs() * safe).max()) <= 2e-2 * float(zb.grad.abs().max())
        n = b * h * w
        var_u = var.flatten() * n / (n - 1)
        assert float(((rv.double() - (0.99 + 0.01 * var_u)) / (0.01 * var_u)).abs().max()) <= 2e-3   # the variance survives every offset
        assert float((rm.double() - 0.01 * mean.flatten()).abs().max()) <= 1e-6 + 1e-7 * 1000


def test_first_layer_conv_recomputed_in_bn_passes(dev, monkeypatch):
    """iris_conv0_* behind _ConvBNReLU for the model's first layer (1 or 2 input channels, no input gradient): output, weight /
    BatchNorm gradients and running statistics equal the stock conv2d + BatchNorm + ReLU; the convolution output is never
    stored (the Function saves x, not z)."""
    import copy
    from challenge_amd import sj_train as S
    S.configure_miopen()
    torch.manual_seed(6)
    for cin, cout, b, h, w in [(1, 32, 3, 16, 40), (2, 32, 2, 9, 7), (1, 64, 2, 5, 33), (2, 8, 1, 1, 3), (1, 32, 8, 64, 128)]:
        blk = S._ConvBNReLU(cin, cout).to(dev).to(memory_format=torch.channels_last).train()
        with torch.no_grad():
            blk[1].weight.uniform_(-1.5, 1.5)
            blk[1].bias.uniform_(-0.3, 0.3)
            blk[1].running_mean.uniform_(-0.2, 0.2)
            blk[1].running_var.uniform_(0.5, 1.5)
            blk[0].bias.uniform_(-0.5, 0.5)
        ref = copy.deepcopy(blk)
        x = torch.randn(b, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        g = torch.randn(b, cout, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        monkeypatch.setattr(S, "FUSED_BN_RELU", True)
        monkeypatch.setattr(S, "FUSED_CONV0", True)
        ya = blk(x)
        assert ya.grad_fn.name().startswith("_FusedConv0BNReLU") and ya.is_contiguous(memory_format=torch.channels_last)
        assert all(t.shape != ya.shape for t in ya.grad_fn.saved_tensors)      # nothing of the output's size is kept
        ya.backward(g)
        monkeypatch.setattr(S, "FUSED_BN_RELU", False)
        yb = ref(x)
        yb.backward(g)
        tol = lambda t: 2e-5 * float(t.abs().max()) + 1e-6  # noqa: E731
        assert float((ya - yb).abs().max()) <= tol(yb), (cin, cout, h, w)
        assert blk[0].weight.grad.shape == ref[0].weight.grad.shape
        assert float((blk[0].weight.grad - ref[0].weight.grad).abs().max()) <= tol(ref[0].weight.grad) * 5, (cin, cout, h, w)
        assert float((blk[1].weight.grad - ref[1].weight.grad).abs().max()) <= tol(ref[1].weight.grad) * 5
        assert float((blk[1].bias.grad - ref[1].bias.grad).abs().max()) <= tol(ref[1].bias.grad) * 5
        assert float(blk[0].bias.grad.abs().max()) == 0.0
        assert float((blk[1].running_mean - ref[1].running_mean).abs().max()) <= 1e-6
        assert float((blk[1].running_var - ref[1].running_var).abs().max()) <= 1e-6
        # with FUSED_CONV0 off the generic fused passes run on MIOpen's convolution output: the same function
        monkeypatch.setattr(S, "FUSED_BN_RELU", True)
        monkeypatch.setattr(S, "FUSED_CONV0", False)
        blk.zero_grad()
        yc = blk(x)
        assert yc.grad_fn.name().startswith("_FusedBiasBNReLU") and float((yc - ya).abs().max()) <= tol(ya)


def test_fused_dense_bn_relu_matches_torch(dev, monkeypatch):
    """FullyConnectedLayer (Dense + BatchNorm over the feature axis + ReLU, sj_train.py:204-211) through the iris_bn_* passes
    on the [B T, C] activation: output, input gradient, weight / BatchNorm gradients, running statistics equal the stock
    Linear + BatchNorm1d + ReLU."""
    import copy
    from challenge_amd import sj_train as S
    torch.manual_seed(8)
    for cin, cout, b, t in [(1024, 512, 64, 16), (256, 64, 3, 5), (128, 8, 1, 1), (64, 128, 2, 33)]:
        if b * t == 1:
            continue  # BatchNorm1d refuses a single value per channel in training mode
        fc = S.FullyConnectedLayer(cin, cout, BN=True).to(dev).train()
        with torch.no_grad():
            fc.bn.weight.uniform_(-1.5, 1.5)
            fc.bn.bias.uniform_(-0.3, 0.3)
            fc.bn.running_mean.uniform_(-0.2, 0.2)
            fc.bn.running_var.uniform_(0.5, 1.5)
        ref = copy.deepcopy(fc)
        x = torch.randn(b, t, cin, device=dev)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        g = torch.randn(b, t, cout, device=dev)
        monkeypatch.setattr(S, "FUSED_FC_BN", True)
        ya = fc(xa)
        assert ya.shape == (b, t, cout)
        ya.backward(g)
        monkeypatch.setattr(S, "FUSED_FC_BN", False)
        yb = ref(xb)
        yb.backward(g)
        tol = lambda v: 2e-5 * float(v.abs().max()) + 1e-6  # noqa: E731
        assert float((ya - yb).abs().max()) <= tol(yb), (cin, cout, b, t)
        assert float((xa.grad - xb.grad).abs().max()) <= tol(xb.grad) * 5
        assert float((fc.fc.weight.grad - ref.fc.weight.grad).abs().max()) <= tol(ref.fc.weight.grad) * 5
        assert float((fc.bn.weight.grad - ref.bn.weight.grad).abs().max()) <= tol(ref.bn.weight.grad) * 5
        assert float((fc.bn.bias.grad - ref.bn.bias.grad).abs().max()) <= tol(ref.bn.bias.grad) * 5
        assert float(fc.fc.bias.grad.abs().max()) == 0.0
        assert float((fc.bn.running_mean - ref.bn.running_mean).abs().max()) <= 1e-6
        assert float((fc.bn.running_var - ref.bn.running_var).abs().max()) <= 1e-6


def test_fused_bn_relu_pool_block_matches_torch(dev, monkeypatch):
    """A ConvMPBlock with its MaxPool folded into the last layer's BatchNorm + ReLU passes (iris_bn_relu_pool_*): outputs,
    every gradient and the running statistics equal the stock torch / MIOpen ops, for even and odd heights / widths
    (ceil_mode windows at the edges) and for blocks of one and two convolutions."""
    import copy
    from challenge_amd import sj_train as S
    S.configure_miopen()
    torch.manual_seed(5)
    for cin, cout, nconv, b, h, w in [(1, 32, 2, 3, 16, 40), (32, 64, 2, 2, 9, 7), (64, 128, 1, 2, 5, 6), (16, 512, 1, 1, 1, 3)]:
        blk = S.ConvMPBlock(cin, num_convs=nconv, fsize=cout, BN=True, MP=True).to(dev).to(memory_format=torch.channels_last).train()
        with torch.no_grad():
            for m in blk.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.weight.uniform_(-1.5, 1.5)   # negative scales too: the maximum must be taken after the affine map
                    m.bias.uniform_(-0.3, 0.3)
        ref = copy.deepcopy(blk)
        x = torch.randn(b, cin, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        g = torch.randn(b, cout, (h + 1) // 2, (w + 1) // 2, device=dev).contiguous(memory_format=torch.channels_last)
        monkeypatch.setattr(S, "FUSED_BN_RELU", True)
        monkeypatch.setattr(S, "FUSED_BN_POOL", True)
        ya = blk(xa)
        assert ya.grad_fn.name().startswith("_FusedBiasBNReLU")   # the pooling ran inside the fused passes
        ya.backward(g)
        monkeypatch.setattr(S, "FUSED_BN_RELU", False)
        yb = ref(xb)
        yb.backward(g)
        assert ya.shape == yb.shape and ya.is_contiguous(memory_format=torch.channels_last)
        tol = lambda t: 2e-5 * float(t.abs().max()) + 1e-6  # noqa: E731
        assert float((ya - yb).abs().max()) <= tol(yb)
        assert float((xa.grad - xb.grad).abs().max()) <= tol(xb.grad) * 5
        for (na, pa), (nb, pb) in zip(blk.named_parameters(), ref.named_parameters()):
            if na.endswith("0.bias"):   # convolution bias: exact zero on the fused path, rounding noise on the stock one
                assert float(pa.grad.abs().max()) == 0.0
                continue
            assert float((pa.grad - pb.grad).abs().max()) <= tol(pb.grad) * 5, (na, cin, cout, h, w)
        for ma, mb in zip(blk.modules(), ref.modules()):
            if isinstance(ma, torch.nn.BatchNorm2d):
                assert float((ma.running_mean - mb.running_mean).abs().max()) <= 1e-6
                assert float((ma.running_var - mb.running_var).abs().max()) <= 1e-6
        # the unfolded form of the fused passes (pooling as the module) is the same function as well
        monkeypatch.setattr(S, "FUSED_BN_RELU", True)
        monkeypatch.setattr(S, "FUSED_BN_POOL", False)
        blk.zero_grad()
        xc = x.clone().requires_grad_(True)
        yc = blk(xc)
        yc.backward(g)
        assert float((yc - ya).abs().max()) <= tol(ya) and float((xc.grad - xa.grad).abs().max()) <= tol(xa.grad) * 5


def test_first_conv_bias_relu_one_pass(dev):
    """iris_conv3x3_small_bias_relu_nchw (the CRNN's first layer, 1 or 2 input channels): equals relu(conv2d + bias) on
    odd heights, one-row and one-column-group images, and negative biases."""
    from challenge_amd import frontend as FE
    torch.manual_seed(13)
    for b, cin, cout, h, w in [(3, 1, 32, 16, 40), (2, 2, 32, 9, 8), (1, 1, 7, 1, 4), (2, 2, 5, 3, 132), (64, 1, 32, 64, 512)]:
        x = torch.randn(b, cin, h, w, device=dev)
        wt = torch.randn(cout, cin, 3, 3, device=dev) * 0.3
        bias = torch.randn(cout, device=dev) * 0.5
        want = torch.relu(torch.nn.functional.conv2d(x, wt, bias, padding=1))
        got = FE.conv3x3_small_bias_relu_nchw(x, wt, bias)
        assert got.shape == want.shape and got.is_contiguous()
        assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), (b, cin, cout, h, w)
        if cout % 4 == 0 and 1024 % cout == 0:   # the channels-last form (what the matrix-core convolution behind it reads)
            got = FE.conv3x3_small_bias_relu(x, wt, bias, channels_last=True)
            assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
            assert float((got - want).abs().max()) <= 2e-6 * max(1.0, float(want.abs().max())), (b, cin, cout, h, w)
    with pytest.raises(ValueError):
        FE.conv3x3_small_bias_relu_nchw(torch.zeros(1, 3, 4, 4, device=dev), torch.zeros(8, 3, 3, 3, device=dev), torch.zeros(8, device=dev))
    with pytest.raises(ValueError):
        FE.conv3x3_small_bias_relu_nchw(torch.zeros(1, 1, 4, 6, device=dev), torch.zeros(8, 1, 3, 3, device=dev), torch.zeros(8, device=dev))


def test_conv32_matrix_core_kernel(dev):
    """iris_conv3x3_c32_bias_relu (32 -> 32 channels, implicit GEMM on the fp32 MFMA, bias + ReLU (+ MaxPool) fused): equals
    relu(conv2d + bias) and its 2x2 'same' max-pool on sizes that are and are not multiples of the 4 x 64 tile, incl. odd ones."""
    from challenge_amd import frontend as FE
    torch.manual_seed(14)
    for b, h, w in [(2, 8, 64), (1, 5, 70), (3, 4, 7), (2, 1, 1), (1, 9, 129), (4, 64, 512)]:
        x = torch.randn(b, 32, h, w, device=dev).contiguous(memory_format=torch.channels_last)
        wt = torch.randn(32, 32, 3, 3, device=dev) * 0.1
        bias = torch.randn(32, device=dev) * 0.5
        want = torch.relu(torch.nn.functional.conv2d(x, wt, bias, padding=1))
        got = FE.conv3x3_c32_bias_relu(x, wt, bias)
        assert got.shape == want.shape and got.is_contiguous(memory_format=torch.channels_last)
        tol = 2e-5 * max(1.0, float(want.abs().max()))
        assert float((got - want).abs().max()) <= tol, (b, h, w, float((got - want).abs().max()))
        wantp = torch.nn.functional.max_pool2d(want, 2, 2, ceil_mode=True)
        gotp = FE.conv3x3_c32_bias_relu(x, wt, bias, pool=True)
        assert gotp.shape == wantp.shape and float((gotp - wantp).abs().max()) <= tol, (b, h, w)
        # the same values in the channel-chunked layout the Winograd layers read
        assert torch.equal(FE.conv3x3_c32_bias_relu(x, wt, bias, out_chunked=True), FE.to_chunked(got))
        assert torch.equal(FE.conv3x3_c32_bias_relu(x, wt, bias, pool=True, out_chunked=True), FE.to_chunked(gotp))
    with pytest.raises(ValueError):
        FE.conv3x3_c32_bias_relu(torch.zeros(1, 16, 4, 4, device=dev).contiguous(memory_format=torch.channels_last),
                                 torch.zeros(32, 32, 3, 3, device=dev), torch.zeros(32, device=dev))


def test_model_predict_uses_a_cached_engine(dev):
    """CustomModel.predict on the GPU: through an InferenceEngine that is reused while the weights stand and rebuilt after a
    training step; values equal the module in eval mode to 1e-4."""
    from challenge_amd import sj_train as S
    S.configure_miopen()
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '128', '--n_chan', '1', '--batch_size', '4'])
    torch.manual_seed(2)
    model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    model.compile(S.make_optimizer(cfg, model.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
    x = torch.rand(6, 64, 128, 1, device=dev)
    y = (torch.rand(6, 4, 3, device=dev) < 0.2).float()

    def reference():
        model.eval()
        with torch.no_grad():
            return model(x)

    p1 = model.predict(x, batch_size=4)
    eng1 = model._predict_engine[1]
    assert eng1.hip_convs == 14 and eng1.wino_convs == 12 and eng1.fused_lstm and float((p1 - reference()).abs().max()) <= 1e-4
    model.predict(x)
    assert model._predict_engine[1] is eng1                       # unchanged weights: the same engine
    model.train_step((x[:4], y[:4]))
    p2 = model.predict(x)
    assert model._predict_engine[1] is not eng1 and model.training   # rebuilt; the training flag is left alone
    assert float((p2 - reference()).abs().max()) <= 1e-4 and float((p2 - p1).abs().max()) > 0


@pytest.mark.parametrize("b,h,w,cin,cout", [(2, 32, 256, 32, 64), (2, 32, 256, 64, 64), (3, 16, 128, 64, 128), (2, 16, 128, 128, 128),
                                            (5, 8, 64, 256, 256), (3, 4, 32, 512, 512), (7, 2, 16, 512, 512),     # the CRNN's own shapes
                                            (2, 40, 256, 32, 64), (3, 5, 32, 256, 512), (1, 7, 9, 16, 64), (2, 3, 70, 8, 128),
                                            (1, 1, 1, 8, 64), (130, 2, 2, 24, 64)])                                   # odd sizes, edges
def test_winograd_convolution_matches_fp64(dev, b, h, w, cin, cout):
    """Blocks 2-5 of the inference engine: Conv2D 3x3 'same' + bias + ReLU (+ MaxPool 2x2 'same') as Winograd F(2x2, 3x3) on
    the fp32 matrix cores (iris_conv3x3_wino_bias_relu) against an fp64 convolution: as close as MIOpen's direct fp32
    convolution is (stated bound 2e-6 of the output's peak, i.e. 10x the typical error); pooled and unpooled, chunked and
    channels-last outputs identical; every tile-block geometry (64, 32, 16 tile columns), odd heights / widths, tile rows that
    straddle images, batches that do not fill a block."""
    from challenge_amd import frontend as FE
    g = torch.Generator(device=dev).manual_seed(b * 1000 + h)
    x = torch.randn(b, cin, h, w, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g, device=dev) * 0.1
    packed = FE.wino_pack_weights(wt)
    xc = FE.to_chunked(x)
    assert tuple(xc.shape) == (b, cin // 8, h, w, 8)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1).relu()
    for pool in (False, True):
        want = torch.nn.functional.max_pool2d(ref, 2, 2, ceil_mode=True) if pool else ref
        y_cl = FE.conv3x3_wino_bias_relu(xc, packed, bias, cout, pool=pool, out_nhwc=True)
        assert tuple(y_cl.shape) == tuple(want.shape) and y_cl.is_contiguous(memory_format=torch.channels_last)
        err = float((y_cl.double() - want).abs().max() / want.abs().max().clamp_min(1e-30))
        assert err <= 2e-6, (pool, err)
        y_ch = FE.conv3x3_wino_bias_relu(xc, packed, bias, cout, pool=pool)
        ho, wo = want.shape[2], want.shape[3]
        assert tuple(y_ch.shape) == (b, cout // 8, ho, wo, 8)
        assert torch.equal(y_ch.permute(0, 1, 4, 2, 3).reshape(b, cout, ho, wo), y_cl)
        assert torch.equal(FE.to_chunked(y_cl), y_ch)      # the next layer's input, either way
    with pytest.raises(ValueError):
        FE.conv3x3_wino_bias_relu(xc, packed, bias, cout + 8)       # cout must be a multiple of 64
    with pytest.raises(ValueError):
        FE.conv3x3_wino_bias_relu(x, packed, bias, cout)            # not the chunked layout


@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("b,h,w,cin,cout,offset", [(3, 16, 128, 64, 128, 0.0), (2, 7, 9, 32, 64, 0.0), (5, 4, 32, 256, 512, 0.0),
                                                   (2, 33, 70, 32, 32, 0.0), (4, 8, 64, 128, 256, 30.0), (3, 9, 21, 32, 32, 30.0)])
def test_batchnorm_statistics_from_the_convolution_epilogue(dev, b, h, w, cin, cout, offset, split, monkeypatch):
    """Round 6: the convolution kernels of the training step (Winograd, split-bf16 Winograd, the 32 -> 32 kernel) accumulate the
    statistics of the BatchNorm behind them in their epilogue (fp32 about a local shift per lane, fp64 about zero when flushed,
    bn_epilogue.h) and `_FusedBiasBNReLU(..., sums0)` skips its pass over z.  Against the separate pass on the same z: outputs,
    saved statistics and running statistics equal to fp32 rounding, gradients too; with an input offset (z's mean ~ 30x its
    spread) the variance stays right; odd sizes (tiles outside the image must not be counted); pooled and unpooled."""
    from challenge_amd import sj_train as S
    from challenge_amd.hip_autograd import _FusedBiasBNReLU, _WinoConv3x3
    if split and (cin % 16 or (cin, cout) == (32, 32)):
        pytest.skip("the split-bf16 kernel takes 16 | cin and 64 | cout")
    monkeypatch.setattr(S, "WINO_SPLIT_BF16", split)
    g = torch.Generator(device=dev).manual_seed(b + h + cout)
    x = (torch.randn(b, cin, h, w, generator=g, device=dev) + offset).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5 + (0.02 if offset else 0.0))
    wt = wt.contiguous(memory_format=torch.channels_last).requires_grad_(True)
    gamma, beta = torch.rand(cout, generator=g, device=dev) + 0.5, torch.randn(cout, generator=g, device=dev) * 0.1
    fwd = 'c32' if (cin, cout) == (32, 32) else True
    for pool in (False, True):
        outs = []
        for fused in (True, False):
            rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
            w_ = wt.detach().clone().requires_grad_(True)
            if fused:
                z, sums0 = _WinoConv3x3.apply(x, w_, fwd, False, False, True)
                assert sums0 is not None and sums0.dtype == torch.float64
            else:
                z, sums0 = _WinoConv3x3.apply(x, w_, fwd, False, False), None
            y = _FusedBiasBNReLU.apply(z, None, gamma, beta, rm, rv, 1e-3, 0.01, pool, sums0)
            y.backward(torch.ones_like(y))
            outs.append((y.detach(), rm, rv, w_.grad, z.detach()))
        (y1, m1, v1, g1, z1), (y0, m0, v0, g0, z0) = outs
        assert torch.equal(z1, z0)                                     # the convolution itself does not change
        zd = z0.double()
        mean64, var64 = zd.mean((0, 2, 3)), zd.var((0, 2, 3), unbiased=True)
        for got_m, got_v in ((m1, v1), (m0, v0)):                     # both forms against fp64 on the same z
            # (+ 2e-7: the running values are stored in fp32 - running_var sits near 1, one ulp = 1.2e-7)
            assert float((got_m.double() - 0.01 * mean64).abs().max()) <= 1e-6 * float(mean64.abs().max()) * 0.01 + 2e-7
            assert float(((got_v.double() - 0.99) - 0.01 * var64).abs().max()) <= 2e-6 * float(var64.max()) * 0.01 + 2e-7
        assert float((y1 - y0).abs().max()) <= 2e-6 * float(y0.abs().max()) + 1e-6, (pool, float((y1 - y0).abs().max()))
        # (with the 30-sigma input offset the weight gradient sums inputs ~30 times their spread: a 1e-6 difference in rstd shows at 2-3e-5)
        assert float((g1 - g0).abs().max()) <= (1e-4 if offset else 2e-5) * float(g0.abs().max()) + 1e-9


@pytest.mark.parametrize("b,h,w,cin,cout", [(2, 32, 256, 32, 64), (2, 32, 256, 64, 64), (3, 16, 128, 64, 128), (2, 16, 128, 128, 128),
                                            (5, 8, 64, 256, 256), (3, 4, 32, 512, 512), (7, 2, 16, 512, 512),     # the CRNN's own shapes
                                            (2, 40, 256, 32, 64), (3, 5, 32, 256, 512), (1, 7, 9, 16, 64), (2, 3, 70, 48, 128),
                                            (1, 1, 1, 16, 64), (130, 2, 2, 32, 64)])                                  # odd sizes, edges
def test_winograd_split_bf16_convolution_matches_fp64(dev, b, h, w, cin, cout):
    """The same convolution with its GEMMs on the BF16 matrix cores (iris_conv3x3_wino_b3: both operands split into three bf16
    terms, six partial products accumulated in fp32) under the UNCHANGED fp64-referenced gates of the exact-fp32 kernel: error
    <= 1.5x that kernel's on the same input (floor 3e-7: both sit at a few ulp there) and <= 2e-6 of the output's peak, that
    kernel's own bound - nothing relaxed; pooled / unpooled, chunked / channels-last inputs and outputs bit-identical; all three tile geometries, odd
    sizes, tile rows straddling images.  Measured 0.65 - 1.14x (profiles/r6/wino_b3_check_and_time.log)."""
    from challenge_amd import frontend as FE
    g = torch.Generator(device=dev).manual_seed(b * 1000 + h)
    x = torch.randn(b, cin, h, w, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5
    bias = torch.randn(cout, generator=g, device=dev) * 0.1
    packed, packed3 = FE.wino_pack_weights(wt), FE.wino_pack_weights_device(wt, split_bf16=True)
    assert packed3.numel() == 24 * cin * cout                     # 96 bytes per weight pair: 16 positions x 3 bf16 terms
    xc = FE.to_chunked(x)
    ref = torch.nn.functional.conv2d(x.double(), wt.double(), bias.double(), padding=1).relu()
    for pool in (False, True):
        want = torch.nn.functional.max_pool2d(ref, 2, 2, ceil_mode=True) if pool else ref
        peak = want.abs().max().clamp_min(1e-30)
        exact = FE.conv3x3_wino_bias_relu(xc, packed, bias, cout, pool=pool, out_nhwc=True)
        y_cl = FE.conv3x3_wino_bias_relu(xc, packed3, bias, cout, pool=pool, out_nhwc=True, split_bf16=True)
        assert tuple(y_cl.shape) == tuple(want.shape) and y_cl.is_contiguous(memory_format=torch.channels_last)
        e_exact, e_split = float((exact.double() - want).abs().max() / peak), float((y_cl.double() - want).abs().max() / peak)
        # (2e-6 = the exact-fp32 kernel's own stated bound, test_winograd_convolution_matches_fp64: on 7 x 2 x 16, 512 -> 512 that
        # kernel reads 1.22e-6 and this one 1.02e-6)
        assert e_split <= max(1.5 * e_exact, 3e-7) and e_split <= 2e-6, (pool, e_exact, e_split)
        y_ch = FE.conv3x3_wino_bias_relu(xc, packed3, bias, cout, pool=pool, split_bf16=True)
        assert torch.equal(FE.to_chunked(y_cl), y_ch)
        y_in = FE.conv3x3_wino(x, packed3, bias, cout, pool=pool, out_nhwc=True, relu=True, split_bf16=True)   # channels-last input
        assert torch.equal(y_in, y_cl)
    bare = FE.conv3x3_wino(x, packed3, None, cout, out_nhwc=True, relu=False, split_bf16=True)                  # the training form
    want = torch.nn.functional.conv2d(x.double(), wt.double(), None, padding=1)
    assert float((bare.double() - want).abs().max() / want.abs().max()) <= 2e-6
    with pytest.raises(ValueError):
        FE.wino_pack_weights_device(torch.zeros(64, 8, 3, 3, device=dev), split_bf16=True)   # 16 | cin


@pytest.mark.parametrize("b,h,w,cin,cout", [(4, 8, 64, 128, 256), (2, 4, 32, 512, 512), (3, 5, 9, 64, 128), (2, 6, 10, 32, 64)])
def test_winograd_split_bf16_training_convolution_matches_torch(dev, b, h, w, cin, cout, monkeypatch):
    """sj_train._WinoConv3x3 with IRIS_WINO_SPLIT_BF16 on: forward and backward-data through the BF16-matrix-core kernel (weights
    packed and split on the device, plain and transposed / flipped), the weight gradient by the exact-fp32 Winograd kernel as
    before - output and both gradients equal torch's fp64 convolution under the bounds of the exact-fp32 path (2e-6)."""
    from challenge_amd import sj_train as S
    monkeypatch.setattr(S, "WINO_SPLIT_BF16", True)
    g = torch.Generator(device=dev).manual_seed(cin + cout + h)
    x = torch.randn(b, cin, h, w, generator=g, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5).contiguous(memory_format=torch.channels_last)
    wt.requires_grad_(True)
    z = S._WinoConv3x3.apply(x, wt, cout % 64 == 0, cout % 8 == 0 and cin % 64 == 0, True)
    dz = torch.randn(z.shape, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    z.backward(dz)
    x2, w2 = x.detach().clone().requires_grad_(True), wt.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.conv2d(x2.double(), w2.double(), None, padding=1)
    ref.backward(dz.double())

    def rel(a, r):
        return float((a.double() - r).abs().max() / r.abs().max())
    assert rel(z, ref) <= 2e-6 and rel(x.grad, x2.grad) <= 2e-6 and rel(wt.grad, w2.grad) <= 2e-6, (rel(z, ref), rel(x.grad, x2.grad))
    # it really took the other kernel: the exact-fp32 path gives (slightly) different bits
    monkeypatch.setattr(S, "WINO_SPLIT_BF16", False)
    z32 = S._WinoConv3x3.apply(x.detach(), wt.detach(), cout % 64 == 0, False, False)
    assert not torch.equal(z32, z.detach()) and rel(z32, ref) <= 2e-6


@pytest.mark.parametrize("b,h,w,cin,cout", [(4, 8, 64, 128, 256), (3, 8, 64, 256, 256), (5, 4, 32, 256, 512), (2, 4, 32, 512, 512),
                                            (3, 5, 9, 64, 128), (2, 6, 10, 24, 64), (2, 6, 10, 32, 64), (2, 6, 10, 32, 32), (3, 9, 70, 32, 32)])
def test_winograd_training_convolution_matches_torch(dev, b, h, w, cin, cout):
    """The training step's deep convolutions (sj_train._WinoConv3x3): forward z = conv(x, W) and backward-data dx by the
    Winograd kernel on channels_last tensors with the weights packed on the device (plain and transposed / flipped), dW by
    the Winograd weight-gradient kernel where both channel counts are multiples of 32 (MIOpen otherwise) - equal to torch's
    conv2d and its autograd gradients; the device packing equals the host packing to fp32 rounding; the gradient of a
    channels_last parameter comes back in the parameter's own strides.
    (cin = 24 -> 64: the backward-data pass's shape rule fails for the swapped channel counts and MIOpen computes dx.)"""
    from challenge_amd import frontend as FE
    from challenge_amd import sj_train as S
    g = torch.Generator(device=dev).manual_seed(cin + cout + h)
    x = torch.randn(b, cin, h, w, generator=g, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wt = (torch.randn(cout, cin, 3, 3, generator=g, device=dev) * (2.0 / (9 * cin)) ** 0.5).contiguous(memory_format=torch.channels_last)
    wt.requires_grad_(True)
    # (the host packs in double precision and rounds once, the device in fp32: equal to a few ulp of the largest weight)
    wmax = float(wt.detach().abs().max())
    if cout % 64 == 0:
        assert float((FE.wino_pack_weights_device(wt.detach()) - FE.wino_pack_weights(wt.detach())).abs().max()) <= 4e-7 * wmax
    if cout % 8 == 0 and cin % 64 == 0:   # the transposed packing == the host packing of the flipped, transposed weight
        flipped = wt.detach().flip(2, 3).transpose(0, 1).contiguous()
        assert float((FE.wino_pack_weights_device(wt.detach(), transposed=True) - FE.wino_pack_weights(flipped)).abs().max()) <= 4e-7 * wmax
    wrw = cin % 32 == 0 and cout % 32 == 0
    c32 = 'c32' if (cin, cout) == (32, 32) else None   # 32 -> 32: forward / backward-data by the bare implicit-GEMM kernel
    z = S._WinoConv3x3.apply(x, wt, c32 or cout % 64 == 0, c32 or (cout % 8 == 0 and cin % 64 == 0), wrw)
    assert z.is_contiguous(memory_format=torch.channels_last)
    dz = torch.randn(z.shape, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    z.backward(dz)
    x2, w2 = x.detach().clone().requires_grad_(True), wt.detach().clone().requires_grad_(True)
    ref = torch.nn.functional.conv2d(x2.double(), w2.double(), None, padding=1)
    ref.backward(dz.double())

    def rel(a, r):
        return float((a.double() - r).abs().max() / r.abs().max())
    assert rel(z, ref) <= 2e-6 and rel(x.grad, x2.grad) <= 2e-6, (rel(z, ref), rel(x.grad, x2.grad))
    assert rel(wt.grad, w2.grad) <= (2e-6 if wrw else 2e-5)     # (MIOpen's weight gradient: fp32 atomics)
    assert wt.grad.stride() == wt.stride()


@pytest.mark.parametrize("b,h,w,cin,cout", [(2, 8, 12, 64, 64), (3, 7, 9, 64, 128), (1, 5, 33, 128, 64), (2, 16, 128, 128, 128),
                                            (2, 4, 32, 512, 512), (64, 4, 32, 256, 512), (5, 1, 1, 64, 64), (1, 2, 3, 64, 64),
                                            (300, 2, 5, 64, 64), (64, 32, 256, 64, 64),
                                            (2, 9, 14, 32, 64), (64, 32, 256, 32, 64), (3, 6, 8, 96, 128),
                                            (2, 9, 14, 32, 32), (16, 64, 512, 32, 32), (3, 6, 8, 64, 96)])
def test_winograd_weight_gradient_matches_fp64(dev, b, h, w, cin, cout):
    """iris_conv3x3_wino_wrw (csrc/k_conv_wino_wrw.h): dW of the 3x3 'same' convolution as Winograd F(2x2, 3x3) on the fp32
    MFMA, against aten's convolution_backward in float64 - odd heights and widths (tiles hanging over the right / bottom edge),
    single pixels, more tile rows than workgroups and fewer, the step's largest activation (64 x 32 x 256 x 64), 4 and 256
    splits (the two-stage sum).  Error of a direct fp32 gradient (MIOpen's beside it), the same bits on every call, the
    parameter's strides honoured."""
    from challenge_amd import frontend as FE
    g = torch.Generator(device=dev).manual_seed(b + h + w + cin)
    x = torch.randn(b, cin, h, w, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(b, cout, h, w, generator=g, device=dev).contiguous(memory_format=torch.channels_last)
    like_cl = torch.empty(cout, cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    like_nchw = torch.empty(cout, cin, 3, 3, device=dev)
    got = FE.conv3x3_wino_wrw(x, dy, like=like_cl)
    assert got.stride() == like_cl.stride()
    w0 = torch.zeros(cout, cin, 3, 3, device=dev, dtype=torch.float64)
    ref = torch.ops.aten.convolution_backward(dy.double(), x.double(), w0, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                              [False, True, False])[1]
    err = float((got.double() - ref).abs().max() / ref.abs().max())
    assert err <= 2e-6, err
    again = FE.conv3x3_wino_wrw(x, dy, like=like_nchw)
    assert again.stride() == like_nchw.stride() and torch.equal(again, got)
    with pytest.raises(ValueError):
        FE.conv3x3_wino_wrw(x.contiguous(), dy)                       # not channels_last (unless a dimension is 1)
        if 1 in (h * w, cin):
            raise ValueError("layouts coincide")
    with pytest.raises(ValueError):
        FE.conv3x3_wino_wrw(x[:, :16].contiguous(memory_format=torch.channels_last), dy)   # 16 input channels: unsupported


def test_hip_bilstm_matches_torch(dev):
    """iris_bilstm128_forward behind sj_train._HipBiLSTM: the whole bidirectional recurrence in one launch equals
    torch.nn.LSTM (MIOpen) on the same weights - odd batch sizes (a workgroup owns two rows), one step, long sequences,
    large pre-activations (saturated gates)."""
    from challenge_amd import frontend as FE
    from challenge_amd import sj_train as S
    torch.manual_seed(11)
    lstm = torch.nn.LSTM(128, 128, batch_first=True, bidirectional=True).to(dev).eval()
    with torch.no_grad():
        for p in lstm.parameters():
            p.uniform_(-0.25, 0.25)
    hip = S._HipBiLSTM(lstm).to(dev)
    # (inputs 30x larger: pre-activations ~ +-100, where the two implementations' fp32 summation orders differ by ~5e-5
    # before the gates - rounding of the inputs' GEMM, not of the gate functions)
    for b, t, scale, tol in [(64, 16, 1.0, 2e-6), (1, 1, 1.0, 2e-6), (3, 5, 1.0, 2e-6), (7, 40, 1.0, 2e-6), (5, 16, 30.0, 5e-5)]:
        x = torch.randn(b, t, 128, device=dev) * scale
        with torch.no_grad():
            want, _ = lstm(x)
            got, none = hip(x)
        assert none is None and got.shape == want.shape == (b, t, 256)
        assert float((got - want).abs().max()) <= tol, (b, t, scale, float((got - want).abs().max()))
    assert not S._HipBiLSTM.supports(torch.nn.LSTM(128, 64, batch_first=True, bidirectional=True))
    assert not S._HipBiLSTM.supports(torch.nn.LSTM(128, 128, batch_first=True))
    with pytest.raises(ValueError):
        FE.bilstm128_forward(torch.zeros(2, 3, 2, 256, device=dev), torch.zeros(2, 512, 128, device=dev))


def test_hip_bilstm_training_matches_torch(dev):
    """sj_train.bilstm128 under autograd: output, input gradient and the gradients of all eight nn.LSTM parameters equal
    torch's own (MIOpen) LSTM - back-propagation through time inside iris_bilstm128_backward, dW_hh from its dgx."""
    import copy
    from challenge_amd import sj_train as S
    torch.manual_seed(12)
    lstm = torch.nn.LSTM(128, 128, batch_first=True, bidirectional=True).to(dev)
    with torch.no_grad():
        for p in lstm.parameters():
            p.uniform_(-0.3, 0.3)
    ref = copy.deepcopy(lstm)
    for b, t in [(64, 16), (3, 1), (5, 7), (2, 33)]:
        lstm.zero_grad()
        ref.zero_grad()
        x = torch.randn(b, t, 128, device=dev)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        g = torch.randn(b, t, 256, device=dev)
        ya = S.bilstm128(lstm, xa)
        assert ya.grad_fn.name().startswith("_BiLSTM128")
        ya.backward(g)
        yb, _ = ref(xb)
        yb.backward(g)
        tol = lambda v: 2e-5 * float(v.abs().max()) + 1e-6  # noqa: E731
        assert float((ya - yb).abs().max()) <= 2e-6
        assert float((xa.grad - xb.grad).abs().max()) <= tol(xb.grad), (b, t)
        for (n, pa), (_, pb) in zip(lstm.named_parameters(), ref.named_parameters()):
            assert float((pa.grad - pb.grad).abs().max()) <= tol(pb.grad), (n, b, t)
    # no autograd: the plain launch
    with torch.no_grad():
        assert float((S.bilstm128(lstm, x) - ref(x)[0]).abs().max()) <= 2e-6


def test_training_step_is_bit_reproducible(dev):
    """With the three convolution passes on the Winograd / implicit-GEMM kernels (no atomics in any of them; the weight
    gradient's partial sums are added in a fixed order) the training step is a deterministic function of its inputs: two fresh
    models from one seed, three steps each on the same batches - every parameter and every BatchNorm buffer bit-equal.  (With
    MIOpen's convolutions the same comparison gives the discrete 1e-2 states of profiles/r5/grad_reproducibility.log.)"""
    from challenge_amd import sj_train as S
    S.configure_miopen()
    assert S.WINO_TRAIN and S.WINO_TRAIN_WRW and S.C32_TRAIN
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '128', '--n_chan', '1', '--batch_size', '8'])
    g = torch.Generator().manual_seed(9)
    batches = [(torch.rand(8, 64, 128, 1, generator=g).to(dev), (torch.rand(8, 4, 3, generator=g) < 0.2).float().to(dev)) for _ in range(3)]
    states = []
    for _ in range(2):
        torch.manual_seed(4)
        m = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
        m.compile(S.make_optimizer(cfg, m.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
        losses = [float(m.train_step(b)['loss']) for b in batches]
        torch.cuda.synchronize()
        states.append((losses, {n: t.detach().clone() for n, t in list(m.named_parameters()) + list(m.named_buffers())}))
    diff = [n for n in states[0][1] if not torch.equal(states[0][1][n], states[1][1][n])]
    if diff or states[0][0] != states[1][0]:
        # (the BatchNorm sums are fp64 atomics: a different order can, once in ~1e5 runs, round a mean differently - that is a
        # last-bit difference, not the 1e-2 states of atomically accumulated convolutions)
        worst = max(float((states[0][1][n].double() - states[1][1][n].double()).abs().max()) / (float(states[1][1][n].double().abs().max()) + 1e-12)
                    for n in diff) if diff else 0.0
        assert worst <= 1e-5 and max(abs(a - b) for a, b in zip(*[st[0] for st in states])) <= 1e-6, (worst, diff[:5])


def test_fit_on_a_hipgraph_trains_as_the_eager_fit(dev):
    """sj_train.fit with the step as one replayed hipGraph (the default on one GPU with a capturable Adam) against fit with
    graph=False from the same initial state and the same batches: the warm-up steps GraphedTrainStep needs are undone
    (preserve_state), so every batch causes exactly one update - after 2 epochs x 3 steps the two models hold the same
    parameters, BatchNorm statistics and batch counters (the step is bit-reproducible with the Winograd passes; a warm-up that
    left a trace would show as differences of the order of the learning rate), and the CSV rows carry the scheduler's rates."""
    import copy
    from challenge_amd import sj_train as S
    S.configure_miopen()
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '64', '--n_chan', '1', '--batch_size', '8'])
    g = torch.Generator().manual_seed(5)
    batches = [(torch.rand(8, 32, 64, 1, generator=g).to(dev), (torch.rand(8, 2, 3, generator=g) < 0.2).float().to(dev)) for _ in range(6)]

    def forever():
        i = 0
        while True:
            yield batches[i % len(batches)]
            i += 1
    torch.manual_seed(1)
    base = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    out = []
    for graph in (True, False):
        m = copy.deepcopy(base)
        m.compile(S.make_optimizer(cfg, m.parameters(), capturable=True), S.binary_crossentropy, clipvalue=cfg.clipvalue)
        S.fit(m, forever(), epochs=2, steps_per_epoch=3, scheduler=S.custom_scheduler(4096, 2 / 12, 2), verbose=False, graph=graph)
        torch.cuda.synchronize()
        out.append(({n: p.detach().clone() for n, p in m.named_parameters()}, {n: b.detach().clone() for n, b in m.named_buffers()}))
    (pg, bg), (pe, be) = out
    assert any(not torch.equal(pg[n], p) for n, p in base.named_parameters())          # it trained
    worst = max((float((pg[n] - pe[n]).abs().max()) / (float(pe[n].abs().max()) + 1e-12), n) for n in pe)
    assert worst[0] <= 1e-5, worst
    for n in be:
        if be[n].dtype.is_floating_point:
            assert float((bg[n] - be[n]).abs().max()) <= 1e-5 * float(be[n].abs().max()) + 1e-7, n
        else:
            assert torch.equal(bg[n], be[n]), n           # num_batches_tracked: 6 batches, not 6 + warm-up


def test_graphed_train_step_equals_eager(dev):
    """GraphedTrainStep (the whole training step as one replayed hipGraph) against the eager step from the same state.
    With the learning rate written to 0 through set_lr the two are the same deterministic function of the batch: equal
    losses, equal BatchNorm running statistics, parameters untouched.  With a learning rate both train; their losses stay
    together only loosely - Adam's normalised update turns the 1e-7 noise of atomically reduced gradients (MIOpen's split-K
    kernels) into +-lr steps on near-zero-gradient parameters, between two EAGER runs just the same."""
    import copy
    from challenge_amd import sj_train as S
    S.configure_miopen()
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '128', '--n_chan', '1', '--batch_size', '4'])
    torch.manual_seed(3)
    b = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    b.compile(S.make_optimizer(cfg, b.parameters(), capturable=True), S.binary_crossentropy, clipvalue=cfg.clipvalue)
    xs = [torch.rand(4, 64, 128, 1, device=dev) for _ in range(3)]
    ys = [(torch.rand(4, 4, 3, device=dev) < 0.2).float() for _ in range(3)]
    step = S.GraphedTrainStep(b, (xs[0], ys[0]), warmup=2)
    a = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    a.load_state_dict(b.state_dict())                    # the eager twin starts where the capture left the model
    a.compile(S.make_optimizer(cfg, a.parameters()), S.binary_crossentropy, clipvalue=cfg.clipvalue)
    before = [p.detach().clone() for p in b.parameters()]
    for g in a.optimizer.param_groups:
        g['lr'] = 0.0
    step.set_lr(0.0)
    for x, y in zip(xs, ys):
        la, lb = a.train_step((x, y))['loss'], step((x, y))['loss']
        assert abs(float(la) - float(lb)) <= 1e-5 * max(1.0, abs(float(la)))
    for p0, pb in zip(before, b.parameters()):
        assert torch.equal(p0, pb)
    for (n, ba), (_, bb) in zip(a.named_buffers(), b.named_buffers()):
        if ba.dtype.is_floating_point:
            assert float((ba - bb).abs().max()) <= 1e-5 * float(ba.abs().max()) + 1e-7, n
        else:
            assert torch.equal(ba, bb), n                # num_batches_tracked
    for g in a.optimizer.param_groups:
        g['lr'] = 1e-3
    step.set_lr(1e-3)
    first = float(step((xs[0], ys[0]))['loss'])
    a.train_step((xs[0], ys[0]))
    for _ in range(12):
        la, lb = float(a.train_step((xs[0], ys[0]))['loss']), float(step((xs[0], ys[0]))['loss'])
    assert lb < first - 0.02 and abs(la - lb) <= 0.05, (first, la, lb)
    with pytest.raises(ValueError):
        S.GraphedTrainStep(a, (xs[0], ys[0]))            # a's optimiser is not capturable
    # predict after replays: a replay moves weights and BatchNorm statistics without touching any ATen version counter,
    # so the cached InferenceEngine must be keyed on the model's generation counter (advisor finding, round 3)
    p_before = b.predict(xs[1])
    eng = b._predict_engine[1]
    step((xs[0], ys[0]))
    p_after = b.predict(xs[1])
    assert b._predict_engine[1] is not eng and float((p_after - p_before).abs().max()) > 0
    b.eval()
    with torch.no_grad():
        assert float((p_after - b(xs[1])).abs().max()) <= 1e-4
    # an EAGER step on the captured model (a ragged last batch falling back) must not rebuild the buffers the graph
    # replays from: the graph owns its FusedAGC, the eager step builds the model's own
    assert b._fused_agc is None and step._agc is not None
    table_ptr, host_ptr = step._agc._table.data_ptr(), step._agc._host_table.data_ptr()
    b.train_step((xs[2][:3], ys[2][:3]))
    assert b._fused_agc is not None and b._fused_agc is not step._agc
    assert step._agc._table.data_ptr() == table_ptr and step._agc._host_table.data_ptr() == host_ptr
    with pytest.raises(RuntimeError):
        step._agc(0.01, 1e-3, None)                      # frozen: never called (and rebuilt) eagerly
    l1 = float(step((xs[0], ys[0]))['loss'])             # replays still run on intact tables
    assert np.isfinite(l1) and all(torch.isfinite(p).all() for p in b.parameters())


def test_inference_engine_matches_module(dev):
    """InferenceEngine (BatchNorm folded, conv + HIP bias/ReLU/pool epilogue, frontend + forward as one hipGraph) is the
    same function as the training module in eval mode: <= 1e-4 on the sigmoid outputs, eager and replayed."""
    from challenge_amd import sj_train as S
    S.configure_miopen()
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '64', '--n_frame', '128', '--n_chan', '1', '--batch_size', '4'])
    torch.manual_seed(1)
    model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    with torch.no_grad():  # non-trivial BatchNorm statistics
        for mod in model.modules():
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.running_mean.uniform_(-0.2, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
                mod.weight.uniform_(0.5, 1.5)
                mod.bias.uniform_(-0.2, 0.2)
    length = 127 * 256
    fe = S.WaveFrontend(1024, 256, 64, 16000, 1, 4, length, dev, training=False)
    wav = torch.randn(4, 1, length, device=dev) * 0.1
    eng = S.InferenceEngine(model, fe, wav)
    assert eng.fused_convs == 14 and eng.fused_lstm and eng.hip_convs == 14 and eng.wino_convs == 12  # every convolution is a HIP kernel
    model.eval()
    with torch.no_grad():
        want = model(fe(wav))
    assert float((eng.eager() - want).abs().max()) <= 1e-4
    assert eng.graph_ok, eng.graph_error
    assert float((eng.replay() - want).abs().max()) <= 1e-4
    wav2 = torch.randn(4, 1, length, device=dev) * 0.1
    with torch.no_grad():
        want2 = model(fe(wav2))
    assert float((eng.replay(wav2) - want2).abs().max()) <= 1e-4
    # with SpecAugment bands drawn on the device per replay the graph still runs (values differ per draw: shape only)
    fe_t = S.WaveFrontend(1024, 256, 64, 16000, 1, 4, length, dev, training=True, device_draw=True, seed=3)
    eng_t = S.InferenceEngine(model, fe_t, wav)
    a, b = eng_t.replay().clone(), eng_t.replay().clone()
    assert tuple(a.shape) == (4, 4, 3) and torch.isfinite(a).all() and torch.isfinite(b).all()


@pytest.mark.parametrize("v,n_mels,n_chan", [(6, 64, 1), (7, 64, 2), (8, 80, 1), (1, 80, 2)])
def test_inference_engine_other_model_variants(dev, v, n_mels, n_chan):
    """The engine's HIP convolutions pick their layers by shape, whatever the variant: v6 / v7 put smoothing pools / bottlenecks
    between the blocks (only the trailing run of plain blocks becomes the Winograd stack), v8 has 48-channel-based widths (96 is no
    multiple of 64: that block stays on MIOpen, the ones behind it convert from channels-last), v1 has no LSTM; 80 mel bands give
    odd heights (5, 3).  Same outputs as the module in eval mode to 1e-4."""
    from challenge_amd import sj_train as S
    S.configure_miopen()
    cfg = S.ARGS().get(['--v', str(v), '--n_mels', str(n_mels), '--n_frame', '128', '--n_chan', str(n_chan), '--batch_size', '3'])
    torch.manual_seed(v)
    model = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    with torch.no_grad():
        for mod in model.modules():
            if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                mod.running_mean.uniform_(-0.2, 0.2)
                mod.running_var.uniform_(0.5, 1.5)
    x = torch.rand(3, n_mels, 128, n_chan, device=dev)
    eng = S.InferenceEngine(model)
    model.eval()
    with torch.no_grad():
        want = model(x)
    got = eng(x)
    assert got.shape == want.shape and float((got - want).abs().max()) <= 1e-4, float((got - want).abs().max())
    assert eng.wino_convs >= 3, eng.wino_convs


def test_bench_self_launch_two_ranks(dev):
    """`python bench.py --gpus 2` without a launcher starts its own ranks (torch.distributed.run children) before it
    touches the GPU; IRIS_BENCH_SHARE_GPU=1 lets both ranks use cuda:0 over gloo so this runs on a one-GPU box."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IRIS_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 2 and res["steps"] == 5 and res["value"] > 0 and res["config"]["global_batch"] == 64
    assert res["scaling"] == "weak" and res["roofline"]["launches_timed"] >= 50
    assert res["roofline"]["frac"] is not None or "frac_withheld" in res["roofline"]   # two ranks share one GPU here
    # the N > 1 line audits itself: who ran where, and how many ranks the collective backend saw
    assert res["rccl_world"] == 2 and [r["rank"] for r in res["ranks"]] == [0, 1]
    assert all(r["value"] > 0 and "device" in r for r in res["ranks"])
    # backend audit: only this share-one-GPU hook may run on anything but RCCL, and the line says so
    assert res["backend"] == "gloo" and res["backend_is_rccl"] is False
    # the scaling target (the training step) is named in `metric` and has its top-level slots (empty with --no-extras)
    assert "SCALING TARGET = the training step" in res["metric"]
    assert all(k in res and res[k] is None for k in ("train_step_ms", "train_step_audio_s_per_s", "train_step_form", "host_ms_per_step",
                                                       "allreduce_exposed_ms", "grad_bytes"))


def test_bench_two_ranks_lift_the_training_step_to_the_top_level(dev):
    """With the side measurements on, the N > 1 line carries the end-to-end training step (the quantity north_star's 8-vs-1
    target is about) at its top level: step time, audio-s/s, exposed all-reduce (DDP.no_sync A/B) and gradient bytes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IRIS_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--no-kernel-events", "--extra-steps", "3"], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "error" not in res["extra"], res["extra"]
    c4 = res["extra"]["c4_train_step"]
    assert res["train_step_ms"] == c4["ms_per_step"] > 0 and res["train_step_audio_s_per_s"] == c4["audio_s_per_s"] > 0
    # (gloo cannot be captured into a hipGraph: the line names the eager form, with the host time it costs per step; over RCCL the
    # replayed graph with the all-reduce inside is timed beside it and lifted when it is the faster one: tests/test_ddp_gpu.py
    # and `IRIS_FORCE_PG=1 python bench.py` exercise that at world size 1)
    assert res["train_step_form"] == "eager DDP" and c4["hipgraph"] is None and res["host_ms_per_step"] == c4["host_ms_per_step"] > 0
    assert res["allreduce_exposed_ms"] == c4["allreduce"]["exposed_allreduce_ms_per_step"]
    assert res["grad_bytes"] == c4["allreduce"]["grad_bytes"] == 4 * c4["params"]
    assert c4["n_gpus"] == 2 and c4["grad_allreduce"].startswith("DDP/")


def test_bench_line_survives_stuck_side_measurements(dev):
    """The side measurements run under a watchdog: when they do not finish within --extras-limit the headline line is
    still printed (with the reason in `extra.error`) and every rank leaves with exit code 0."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IRIS_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--extras-limit", "0.2"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["value"] > 0 and "abandoned" in res["extra"]["error"]
    assert res["roofline"]["launches_timed"] >= 50 and res["rccl_world"] == 2


def test_bench_strong_scaling_two_ranks(dev):
    """`bench.py --gpus 2 --strong`: the fixed global batch (256 c2 clips) is split over the ranks and the line says so."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["IRIS_BENCH_SHARE_GPU"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--strong", "--steps", "5", "--warmup", "2",
                        "--no-cpu-baseline", "--no-extras", "--no-precondition"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["scaling"] == "strong" and res["n_gpus"] == 2 and res["config"]["global_batch"] == 256
    assert "batch 128" in res["config"]["workload"] and res["value"] > 0 and res["rccl_world"] == 2


@pytest.mark.parametrize("orig,new,chan,length", [(44100, 16000, 2, 441000), (48000, 16000, 1, 48001), (8000, 16000, 2, 12345),
                                                  (22050, 16000, 1, 30000), (32000, 16000, 3, 37), (16000, 44100, 1, 900),
                                                  (44100, 16000, 1, 5), (11025, 16000, 2, 1)])
def test_resample_matches_oracle(dev, orig, new, chan, length):
    """iris_resample (the kaldi.resample_waveform of data_utils.py:20-21 = torchaudio.functional.resample's Hann-windowed sinc) against
    the fp64 oracle: <= 2e-6 of the peak (fp32 taps rounded once from fp64, fp32 sums over <= 2 w + o taps), the length
    ceil(n L / o), inputs shorter than one filter, up- and down-sampling, 1-3 channels, a [samples] vector, equal rates = a copy;
    invalid arguments are refused before the device is touched."""
    from challenge_amd import frontend as F
    _, D, _ = mods()
    rng = np.random.default_rng(orig + length)
    x = (rng.standard_normal((chan, length)) * 0.3).astype(np.float32)
    want = R.resample_waveform(x.astype(np.float64), orig, new)
    got = D.resample_waveform(torch.from_numpy(x).to(dev), orig, new)
    assert tuple(got.shape) == want.shape and got.dtype == torch.float32
    assert np.abs(got.cpu().numpy() - want).max() <= 2e-6 * max(np.abs(want).max(), 1e-3)
    flat = F.resample(torch.from_numpy(x[0]).to(dev), orig, new)
    assert flat.dim() == 1 and torch.equal(flat, got[0])
    same = F.resample(torch.from_numpy(x).to(dev), 16000, 16000)
    assert torch.equal(same.cpu(), torch.from_numpy(x))
    with pytest.raises(ValueError):
        F.resample(torch.from_numpy(x).to(dev), 0, 16000)
    with pytest.raises(ValueError):
        F.resample(torch.empty((1, 0), device=dev), 44100, 16000)
    with pytest.raises(RuntimeError):
        F.resample(torch.from_numpy(x), 44100, 16000)    # a CPU tensor: no fallback


def test_load_wav_resamples_a_44k_file(dev, tmp_path):
    """load_wav on a 44.1 kHz stereo PCM file: read -> resample to 16 kHz -> normalize -> STFT(512), against the oracle's chain on the
    same samples (data_utils.py:9-29; the reference reads with torchaudio.load, here scipy's reader: int16 / 32768)."""
    from scipy.io import wavfile
    _, D, _ = mods()
    rng = np.random.default_rng(5)
    pcm = (rng.standard_normal((22050, 2)) * 4000).astype(np.int16)
    path = str(tmp_path / "clip44.wav")
    wavfile.write(path, 44100, pcm)
    spec = D.load_wav(path, dev)
    x = (pcm.astype(np.float32) / 32768.0).T
    ref = R.load_wav_array(x, 512, sample_rate=44100)
    assert tuple(spec.shape) == ref.shape == (257, 1 + 8000 // 256, 4)
    assert np.abs(spec.cpu().numpy() - ref).max() <= 3e-6 * np.abs(ref).max()


def test_weight_packings_in_one_launch(dev):
    """iris_wino_pack_weights_device_multi == iris_wino_pack_weights_device per job, bit for bit (exact-fp32 and split-bf16 packing,
    forward and transposed, contiguous and channels_last weights, more jobs than one launch carries); and the training step's
    `_PackBook`: the first forward packs per layer and remembers, the next one packs everything in one launch, a packing is
    reused only at the weight version it was made at, gradients equal those of the per-layer path bit for bit."""
    from challenge_amd import frontend as F
    from challenge_amd import sj_train as S
    from challenge_amd import hip_autograd as HA
    torch.manual_seed(11)
    shapes = [(64, 32), (64, 64), (128, 64), (128, 128), (256, 128), (64, 16)]
    for split in (False, True):
        jobs, want = [], []
        for rep in range(10):   # 60-100 jobs: several launches of 48
            for co, ci in shapes:
                w = torch.randn(co, ci, 3, 3, device=dev)
                if (rep + co) % 2:
                    w = w.contiguous(memory_format=torch.channels_last)
                for transposed in (False, True):
                    cin, cout = (co, ci) if transposed else (ci, co)
                    if cout % 64 or cin % (16 if split else 8):
                        continue
                    out = torch.empty(F.wino_packed_len(cin, cout, split), device=dev)
                    jobs.append((w, transposed, out))
                    want.append(F.wino_pack_weights_device(w, transposed=transposed, split_bf16=split))
        assert len(jobs) > 48
        F.wino_pack_weights_device_multi(jobs, split_bf16=split)
        for (w, t, out), ref in zip(jobs, want):
            assert torch.equal(out.view(torch.int32), ref.view(torch.int32))
    with pytest.raises(ValueError):
        F.wino_pack_weights_device_multi([(torch.randn(64, 64, 3, 3, device=dev), False, torch.empty(8, device=dev))])
    # the book, on a two-layer block
    S.configure_miopen()
    blk = S.ConvMPBlock(64, num_convs=2, fsize=64, BN=True, MP=True).to(dev).to(memory_format=torch.channels_last).train()
    x = torch.randn(2, 64, 8, 40, device=dev).contiguous(memory_format=torch.channels_last)

    start = [p.detach().clone() for p in blk.parameters()]
    bufs = [b.detach().clone() for b in blk.buffers()]

    def grads(fused):
        S.FUSED_PACK = fused
        HA._PACKS.clear()
        with torch.no_grad():
            for p, q in zip(blk.parameters(), start):
                p.copy_(q)
            for b, q in zip(blk.buffers(), bufs):
                b.copy_(q)
        out = []
        for it in range(3):
            for p in blk.parameters():
                p.grad = None
            HA._PACKS.prepack(dev)
            xx = x.clone().requires_grad_(True)
            blk(xx).square().sum().backward()
            out.append([p.grad.clone() for p in blk.parameters()] + [xx.grad.clone()])
            with torch.no_grad():   # an in-place update: every packing is stale now
                for p in blk.parameters():
                    p.mul_(1.01)
        return out
    try:
        a = grads(True)
        n_entries = len(HA._PACKS.entries)
        b = grads(False)
    finally:
        S.FUSED_PACK = True
        HA._PACKS.clear()
    assert n_entries == 4   # two layers x (forward, backward-data)
    for ga, gb in zip(a, b):
        for u, v in zip(ga, gb):
            assert torch.equal(u, v)


@pytest.mark.parametrize("capturable", [False, True])
def test_agc_clip_adam_in_one_launch_matches_the_two_steps(dev, capturable):
    """iris_agc_clip_adam (AGC + clipvalue + Adam in one launch, sj_train.py:145-155, :434-435) against iris_agc_clip followed by
    torch.optim.Adam's fused update: four training steps of the v9 CRNN from the same state on the same batches - the clipped
    gradients left in p.grad equal (bit for bit behind the last MIOpen pass: the same clip arithmetic), parameters and both moments to fp32 rounding of one
    update (1e-6 of each tensor's peak; Adam's first steps are +-lr whatever the gradient's size, so a wrong moment or bias
    correction shows at 1e-1), the step counters equal; the optimiser states are interchangeable (state_dict of one loaded into
    the other continues identically); a learning-rate change is seen (device tensor for a capturable optimiser, float otherwise)."""
    from challenge_amd import sj_train as S
    S.configure_miopen()
    cfg = S.ARGS().get(['--v', '9', '--n_mels', '32', '--n_frame', '64', '--n_chan', '1', '--batch_size', '4'])
    torch.manual_seed(3)
    base = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
    gen = torch.Generator(device=dev).manual_seed(8)
    batches = [(torch.rand(4, 32, 64, 1, generator=gen, device=dev), (torch.rand(4, 2, 3, generator=gen, device=dev) < 0.3).float())
               for _ in range(5)]

    def make(fused):
        m = S.get_model(cfg).to(dev).to(memory_format=torch.channels_last)
        m.load_state_dict(base.state_dict())
        m.compile(S.make_optimizer(cfg, m.parameters(), capturable=capturable), S.binary_crossentropy, clipvalue=cfg.clipvalue)
        return m

    def set_lr(m, lr):
        for g in m.optimizer.param_groups:
            if torch.is_tensor(g['lr']):
                g['lr'].fill_(lr)
            else:
                g['lr'] = lr

    def rel(a, b):
        return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)

    try:
        # (1) the launch itself, on identical inputs: the same synthetic gradients (some units far above their clip norm, some
        # elements above clipvalue, some tiny) handed to both forms for four steps
        from challenge_amd.hip_autograd import FusedAGC
        S.FUSED_ADAM = True
        a, b = make(True), make(False)
        fa, fb = FusedAGC(list(a.parameters())), FusedAGC(list(b.parameters()))
        assert fa.attach_adam(a.optimizer)
        for k in range(4):
            for p, q in zip(a.parameters(), b.parameters()):
                gr = torch.randn(p.shape, generator=gen, device=dev).contiguous(memory_format=torch.channels_last if p.dim() == 4 else torch.contiguous_format)
                gr = gr * float(10.0 ** float(torch.randint(-6, 1, (1,), generator=gen, device=dev)))
                p.grad, q.grad = gr.clone(memory_format=torch.preserve_format), gr.clone(memory_format=torch.preserve_format)
            if k == 2:
                set_lr(a, 3e-4), set_lr(b, 3e-4)
            assert fa.adam_step(0.01, 1e-3, cfg.clipvalue)
            fb(0.01, 1e-3, cfg.clipvalue)
            b.optimizer.step()
            for (n, p), q in zip(a.named_parameters(), b.parameters()):
                sa, sb = a.optimizer.state[p], b.optimizer.state[q]
                assert rel(p.grad, q.grad) <= 3e-7, (k, n)   # the clipped gradient (two compilations of the norm sums: one ulp of the clip factor)
                assert rel(p, q) <= 1e-6, (k, n, rel(p, q))
                assert rel(sa['exp_avg'], sb['exp_avg']) <= 1e-6 and rel(sa['exp_avg_sq'], sb['exp_avg_sq']) <= 1e-6, (k, n)
                assert float(sa['step']) == float(sb['step']) == k + 1
        # (2) inside train_step (the raw gradients of the two models differ in last bits behind this geometry's one MIOpen pass)
        S.FUSED_ADAM = True
        a = make(True)
        S.FUSED_ADAM = False
        b = make(False)
        for k, batch in enumerate(batches[:4]):
            if k == 2:
                set_lr(a, 3e-4), set_lr(b, 3e-4)
            S.FUSED_ADAM = True
            la = a.train_step(batch)['loss']
            assert a._fused_agc._adam is a.optimizer          # the one-launch path ran
            S.FUSED_ADAM = False
            lb = b.train_step(batch)['loss']
            assert b._fused_agc._adam is None
            if k == 0:   # same state, same batch: the clipped gradients left in p.grad are the same (bit for bit wherever the
                # raw gradients are - the layers behind the one MIOpen pass of this geometry, whose atomics move last bits)
                assert abs(float(la) - float(lb)) <= 1e-6
                same = 0
                for (n, p), q in zip(a.named_parameters(), b.parameters()):
                    assert rel(p.grad, q.grad) <= 1e-4, (n, rel(p.grad, q.grad))
                    same += bool(torch.equal(p.grad, q.grad))
                assert same >= 40, same
            # the two models' raw gradients differ in last bits (MIOpen's atomics), Adam's first steps turn that into a few per cent
            # of a step on near-zero elements and the trajectories drift apart from there: a sanity bound, (1) is the real check
            assert abs(float(la) - float(lb)) <= 2e-4 * (k + 1), (k, float(la), float(lb))
            for (n, p), q in zip(a.named_parameters(), b.parameters()):
                sa, sb = a.optimizer.state[p], b.optimizer.state[q]
                assert float((p - q).abs().max()) <= 0.25 * 1e-3 * (k + 1), (k, n)
                assert float(sa['step']) == float(sb['step']) == k + 1
        # interchangeable state: torch's optimiser continues from the fused one's state and vice versa
        S.FUSED_ADAM = False
        c = make(False)
        c.load_state_dict(a.state_dict())
        c.optimizer.load_state_dict(a.optimizer.state_dict())
        S.FUSED_ADAM = True
        d = make(True)
        d.load_state_dict(a.state_dict())
        d.optimizer.load_state_dict(a.optimizer.state_dict())
        S.FUSED_ADAM = False
        c.train_step(batches[4])
        S.FUSED_ADAM = True
        d.train_step(batches[4])
        a.train_step(batches[4])
        for (n, p), q, r in zip(a.named_parameters(), c.parameters(), d.parameters()):
            assert float((q - p).abs().max()) <= 2.5e-4 and float((r - p).abs().max()) <= 2.5e-4, n
    finally:
        S.FUSED_ADAM = True
