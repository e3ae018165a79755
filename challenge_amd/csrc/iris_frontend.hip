// iris_frontend.hip -- HIP kernels + C ABI of the MI355X audio feature frontend.
// Written for gfx950 (CDNA4) only: 64-lane wavefronts, 160 KiB LDS per CU,
// 8 XCDs with private L2s.  See include/iris_frontend.h for the contract and
// DESIGN.md for the data layout and the roofline of each kernel.
//
// One translation unit, split by subject:
//   common.h         includes, diagnostics switches, error plumbing, the plan, small device helpers
//   iris_fft.h       the wave-per-frame FFT core (registers + private LDS exchanges)
//   spectrum.h       frame loads, untangle (+ magnitude), the per-lane constant block
//   k_fused.h        K1: waveform -> mel magnitudes (the hot path)
//   k_fused_mfma.h   K1m: the same with the mel contraction on the matrix cores (fp16 MFMA variant)
//   k_stft.h         STFT in the reference layout
//   k_magmel.h       spectrum -> mel
//   k_elementwise.h  min-max / log, normalize, magnitude-phase, mask, adaptive gradient clipping
//   k_mix.h          batched sample synthesis (merge_complex_specs)
//   k_draw.h         the random half of a batch drawn on the device (source table, SpecAugment bands)
//   k_lstm.h         the CRNN's bidirectional LSTM: forward and backward through time, one launch each
//   k_conv_small.h   the CRNN's first convolution (1-2 input channels) with bias + ReLU, one pass (inference)
//   k_conv0_bn.h     the same layer in training mode: convolution recomputed inside the BatchNorm + ReLU passes
//   k_conv_c32.h     the 32 -> 32 convolution of block 1 on the fp32 matrix cores, bias + ReLU (+ MaxPool) fused (inference)
//   k_conv_wino.h    blocks 2-5 (64 ... 512 channels) as Winograd F(2x2, 3x3) on the fp32 matrix cores (inference; the training
//                    step's forward and backward-data passes)
//   k_conv_wino_b3.h   the same convolution on the BF16 matrix cores at fp32 accuracy: three-term split of both operands (opt-in)
//   k_conv_wino_wrw.h  the same layers' weight gradient as Winograd F(2x2, 3x3) on the fp32 matrix cores (training)
//   host_plan.h      mel matrix, constant tables, plan create / destroy
//   host_ops.h       the operators' C-ABI entry points
#include "common.h"
#include "spectrum.h"
#include "k_fused.h"
#include "k_fused_mfma.h"
#include "k_stft.h"
#include "k_magmel.h"
#include "k_elementwise.h"
#include "host_plan.h"
#include "host_ops.h"
#include "k_mix.h"
#include "k_draw.h"
#include "k_lstm.h"
#include "k_conv_small.h"
#include "k_conv0_bn.h"
#include "k_conv_c32.h"
#include "k_conv_wino.h"
#include "k_conv_wino_b3.h"
#include "k_conv_wino_wrw.h"
#include "k_resample.h"
#include "k_agc_adam.h"
