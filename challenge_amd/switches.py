"""Every switch of the CRNN side of the path, in ONE place.

Each switch is a module attribute read AT CALL TIME by the code that honours it (`from . import switches as SW`;
`SW.FUSED_BN_RELU`), initialised once from its `IRIS_*` environment variable.  `challenge_amd.sj_train` forwards reads and
writes of these names (`sj_train.WINO_TRAIN = False`, `monkeypatch.setattr(sj_train, "FUSED_BN_RELU", False)`) to this module,
so a test or an A/B script flips one attribute and every module sees it.  The table in INTEGRATION.md section 6 is this file.

Every HIP pass has an off switch that falls back to the STOCK torch / MIOpen operator of the same layer - never to a CPU path -
and every pass is compared against that stock operator in tests/ (outputs and gradients).

Switches read elsewhere (listed in the same INTEGRATION.md table): IRIS_LIB (`_native.py`: path of the shared library),
IRIS_EPILOGUE / IRIS_STREAMS / IRIS_CHUNK_FRAMES / IRIS_MAGMEL_GENERIC / IRIS_ABLATE (`csrc/host_plan.h`: read once per plan),
IRIS_ADAM_FUSED and IRIS_MIOPEN_DB (`fit.py`), IRIS_FORCE_PG (`distributed.py`), IRIS_BENCH_SHARE_GPU (`bench.py` test hook).
"""
import os


def _on(name: str, default: str = "1") -> bool:
    return os.environ.get(name, default) != "0"


# --- training-mode passes around the convolutions (hip_autograd.py; model.py decides per layer) -----------------------------------
# Conv2D bias + BatchNorm + ReLU through the HIP kernels iris_bn_* (two passes over the activation each way instead of seven
# forward / nine backward); IRIS_FUSED_BN=0 keeps the stock torch / MIOpen ops.
FUSED_BN_RELU = _on("IRIS_FUSED_BN")
FUSED_FC_BN = _on("IRIS_FUSED_FC_BN")        # Dense + BatchNorm1d + ReLU through the same passes
FUSED_BN_POOL = _on("IRIS_FUSED_BN_POOL")    # a block's MaxPool inside its last layer's passes
# round 6: the forward statistics (sum z, sum z^2) come out of the convolution kernel's epilogue wherever a HIP convolution produces z
# (Winograd, split-bf16 Winograd, the 32 -> 32 kernel): the iris_bn_stats pass over z disappears; 0: the separate pass
FUSED_BN_STATS = _on("IRIS_FUSED_BN_STATS")
FUSED_CONV0 = _on("IRIS_FUSED_CONV0")        # the first layer (1-2 input channels): convolution recomputed inside every pass
FUSED_LSTM = _on("IRIS_FUSED_LSTM")          # Bidirectional(LSTM(128)): each pass through time in one launch (k_lstm.h)
# zero-initialised scratch (BatchNorm sums, first-layer dW copies, zero bias gradients) from one pool with one fill per step
ZERO_POOL = _on("IRIS_ZERO_POOL")

# round 6: every Winograd weight packing of a training step (12 forward + 11 backward-data layers) in ONE launch at the top of the
# forward pass (iris_wino_pack_weights_device_multi) instead of one launch per layer and pass; 0: per layer
FUSED_PACK = _on("IRIS_FUSED_PACK")

# round 6: AGC + clipvalue + the Adam update in ONE launch (iris_agc_clip_adam) for the optimiser make_optimizer builds (plain Adam,
# fused or capturable); 0: iris_agc_clip, then torch's fused Adam (three more launches, 143 us for the CRNN)
FUSED_ADAM = _on("IRIS_FUSED_ADAM_AGC")

# --- the convolutions themselves in the training step -------------------------------------------------------------------------------
# The bare 3x3 convolutions of blocks 2-5 - forward and backward-data - as Winograd F(2x2, 3x3) on the fp32 matrix cores
# (iris_conv3x3_wino) instead of MIOpen's implicit GEMMs, reading and writing channels_last where the weight-gradient kernel and the
# BatchNorm passes read it (that layout costs the kernel 14 % against its own chunked one).  Wherever the kernel's shape rule holds
# (8 | input channels, 64 | output channels - per direction, the backward-data pass swaps them) and the wider side has >= 64
# channels: every layer of blocks 2-5 forward, all but block 2's first backward.  Thresholds per direction by environment (the
# measured step is flat within noise between 64 and 128: profiles/r5/c4_wino_train_ab.log); IRIS_WINO_TRAIN=0 keeps MIOpen everywhere.
WINO_TRAIN = _on("IRIS_WINO_TRAIN")
WINO_TRAIN_MIN_C_FWD = int(os.environ.get("IRIS_WINO_TRAIN_MIN_C_FWD", "64"))
WINO_TRAIN_MIN_C_BWD = int(os.environ.get("IRIS_WINO_TRAIN_MIN_C_BWD", "64"))
# the weight gradient of the same layers as Winograd on the fp32 MFMA too (k_conv_wino_wrw.h; channel counts multiples of 32):
# 1.6-2.0x MIOpen's weight-gradient kernels on the step's shapes, deterministic; IRIS_WINO_TRAIN_WRW=0 keeps MIOpen's
WINO_TRAIN_WRW = WINO_TRAIN and _on("IRIS_WINO_TRAIN_WRW")
# block 1's 32 -> 32 layer: forward and backward-data by the inference engine's implicit-GEMM kernel (k_conv_c32.h) without
# bias / ReLU instead of CK's / MIOpen's kernels (431 + ~470 us per step); IRIS_C32_TRAIN=0 keeps those
C32_TRAIN = WINO_TRAIN and _on("IRIS_C32_TRAIN")

# Round 6, opt-in: the same Winograd convolutions (forward, backward-data, inference) with their GEMMs on the BF16 matrix cores at fp32
# accuracy - both operands split into three bf16 terms, six partial products accumulated in fp32 (k_conv_wino_b3.h: error against
# fp64 0.65 - 1.14x the exact-fp32 kernel's, 1.13 - 1.57x faster per layer).  Layers with 16 | input channels; default OFF: the headline
# numbers stay on the exact-fp32 kernels.
WINO_SPLIT_BF16 = _on("IRIS_WINO_SPLIT_BF16", "0")

# --- inference ------------------------------------------------------------------------------------------------------------------------
WINO_CONVS = _on("IRIS_WINO")   # blocks 2-5 of the InferenceEngine as Winograd F(2x2, 3x3) on the fp32 MFMA (0: MIOpen + HIP epilogue)

# --- the step as a whole ------------------------------------------------------------------------------------------------------------
# fit / main run the training step as ONE replayed hipGraph wherever that is possible (a GPU, Adam, batches of one shape; under
# DistributedDataParallel: over RCCL, the gradient all-reduce inside the graph): the step then costs what its kernels cost however
# slow the host is at launching ~260 of them - with eight ranks on one host that is the scaling risk, not xGMI.  0: eager.
GRAPH_STEP = _on("IRIS_GRAPH_STEP")
# Gradient buckets of the v9 CRNN (39.5 MB, filled in reverse layer order): with 25 MB the last bucket is 22 MB (block 4's first two
# convolutions and everything below) and its all-reduce starts only when backward has finished, fully exposed; with 12 MB the
# buckets are 0.9 / 7.2 / 9.4 / 9.4 / 11.8 / 0.7 MB - the 11.8 MB one goes out while blocks 2 and 1 (40 % of backward) still
# compute, and what is left after backward is 0.7 MB.  Used by wrap_ddp AND by GraphedTrainStep's own exchange.
DDP_BUCKET_MB = int(os.environ.get("IRIS_DDP_BUCKET_MB", "12"))

# --- test hook ------------------------------------------------------------------------------------------------------------------------
_PLAN_CHECK_ON_CPU = False  # tests/test_ddp_gloo.py: consult the frontend plans' status for a CPU-resident loss too

NAMES = ("FUSED_BN_RELU", "FUSED_FC_BN", "FUSED_BN_POOL", "FUSED_BN_STATS", "FUSED_PACK", "FUSED_ADAM", "FUSED_CONV0", "FUSED_LSTM", "ZERO_POOL", "WINO_TRAIN",
         "WINO_TRAIN_MIN_C_FWD", "WINO_TRAIN_MIN_C_BWD", "WINO_TRAIN_WRW", "C32_TRAIN", "WINO_SPLIT_BF16", "WINO_CONVS", "GRAPH_STEP", "DDP_BUCKET_MB",
         "_PLAN_CHECK_ON_CPU")
