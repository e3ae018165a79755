// bn_epilogue.h -- the copies of the per-channel BatchNorm sums, and how a convolution kernel's epilogue contributes to them.
// Included by k_elementwise.h (the BatchNorm passes) and by the convolution kernels (k_conv_wino.h, k_conv_wino_b3.h, k_conv_c32.h),
// also when those are built alone (scripts/microbench/wino_conv.hip).
#pragma once
#ifndef IRIS_BN_SLOTS
#define IRIS_BN_SLOTS 8
#endif
// The per-channel sums are accumulated by fp64 atomics of every block.  A 32-channel layer has 64 addresses and 2048
// blocks: the same-address atomics queue up behind one another.  Blocks therefore add into one of kBnSlots copies
// (sums[slot][2][C], slot = block % bn_slots(C)) and the consumers add the copies up while they form their per-channel
// coefficients (their blocks are fat - at most 2048 per launch - so that this setup is amortised).  A last-block fold behind an
// arrival counter was measured too: the counter is one more hot address (reductions 0.88 -> 0.99 ms per step).
constexpr int kBnSlots = IRIS_BN_SLOTS;  // at most; bn_slots(channels) of them are used (wide layers have few blocks per address)
__host__ __device__ constexpr int bn_slots(int channels) {
    return channels <= 64 ? kBnSlots : channels <= 128 ? kBnSlots / 2 : channels <= 256 ? kBnSlots / 4 : 1;
}

// ---- BatchNorm statistics from a convolution's EPILOGUE (round 6): the kernel that produces z adds each lane's share of
// sum z and sum z^2 of its output channel to `sums` itself, so that the separate read of z by k_bn_reduce<false> disappears.
// A lane accumulates in fp32 about a LOCAL shift K (one of its own values: the partial sums are then of the size of the
// channel's spread, whatever its mean), and converts to sums about ZERO in fp64 when it flushes:
//   sum z = S1 + n K,   sum z^2 = S2 + 2 K S1 + n K^2      (fp64: E[z^2] - E[z]^2 then cancels to 2^-53 (mean / sigma)^2)
// The consumers are told (`sums_about_zero`) that their K is 0.  Same [slot][2][C] copies as k_bn_reduce.
struct BnEpilogue {
    float k, s1, s2, n;
};
__device__ __forceinline__ void bn_epilogue_add(BnEpilogue& a, float v, float valid /*1 or 0*/) {
    const float d = (v - a.k) * valid;
    a.s1 += d;
    a.s2 = fmaf(d, d, a.s2);
    a.n += valid;
}
__device__ __forceinline__ void bn_epilogue_flush(const BnEpilogue& a, double* sums, int C, int channel, int slot) {
    if (a.n > 0.f) {
        const double K = (double)a.k, n = (double)a.n, S1 = (double)a.s1, S2 = (double)a.s2;
        atomicAdd(sums + (size_t)slot * 2 * C + channel, S1 + n * K);
        atomicAdd(sums + (size_t)slot * 2 * C + C + channel, S2 + 2.0 * K * S1 + n * K * K);
    }
}

