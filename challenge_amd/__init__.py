"""challenge_amd -- MI355X-native feature frontend + data-parallel training step
for the IRIS-AUDIO/challenge hot path (STFT -> mel -> min-max -> log, SpecAugment,
CRNN step).  The frontend is hand-written HIP behind a C ABI
(include/iris_frontend.h); this package is the Python host side that mirrors the
reference's transforms.py / data_utils.py / pipeline.py / sj_train.py / trainer.py
callables on torch tensors.  There is no CPU fallback: hot-path ops raise if the
HIP library or a ROCm device tensor is missing."""

__version__ = "0.1.0"
