// k_conv_c32.h -- Conv2D(32 -> 32, 3x3 'same') + bias + ReLU (+ MaxPool 2x2 'same') on the fp32 matrix cores: the second
// layer of the CRNN's first block at full resolution (sj_train.py:191-201, 244), for inference.
// Part of the single translation unit iris_frontend.hip.
#pragma once
#include "bn_epilogue.h"
// ---------------------------------------------------------------------------
// 38.7 GFLOP at batch 64 x 64 x 512 - the largest single layer of the forward pass, and the one MIOpen's kernels like least
// (32 channels: 0.39-0.46 ms = 85-100 TFLOP/s, then a separate bias / ReLU / pooling pass over the 268 MB output).
// Here: an implicit GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles per SIMD each):
//   * a wave owns a 2-row x 16-column patch of output pixels (M = 32) x all 32 output channels (N = 32) and walks
//     K = 9 taps x 32 input channels in 144 MFMAs; the whole weight tensor (9216 floats) sits in its registers as the 144 B
//     operands (lane l: W[out l & 31][in 2 cp + (l >> 5)][tap]);
//   * the A operand (lane l: pixel l & 31, input channel 2 cp + (l >> 5)) is one ds_read_b32 with an immediate offset from a
//     halo tile of the input in LDS ([rows][cols][36]: 16-byte aligned pixels staged with ds_write_b128);
//   * a workgroup of 4 waves covers 4 rows x 64 columns (each wave two patches), halo 6 x 66 pixels staged once (57 KB);
//     workgroups are persistent (two per CU walk the tiles), so the weights are fetched once per workgroup, not per tile;
//   * output pixel -> MFMA row i = 4 (col >> 1) + 2 row + (col & 1): the four pixels of a pooling window are the four
//     consecutive accumulator registers reg & 3 of one lane, so bias + ReLU + MaxPool is an in-lane max and the pooled tensor
//     (a quarter of the output) is all that is written.
// x [B, H, W, 32] channels-last, w [32][32][3][3], bias [32]; y [B, H, W, 32] or, pooled, [B, ceil(H/2), ceil(W/2), 32].
// ---------------------------------------------------------------------------
constexpr int kC32 = 32;
constexpr int kC32TileH = 4, kC32TileW = 64;
constexpr int kC32PixPitch = 36;                                                  // floats per staged pixel: 16-byte aligned
constexpr int kC32RowPitch = (kC32TileW + 2) * kC32PixPitch;
constexpr int kC32HaloRows = kC32TileH + 2;
constexpr size_t kC32LdsBytes = (size_t)kC32HaloRows * kC32RowPitch * sizeof(float);

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) float lds_float;

// byte offset of the A operand of MFMA j of group g (tap g >> 1, input-channel pair 8 (g & 1) + j) from the lane's base
constexpr int c32_off(int g, int j) {
    return 4 * (((g >> 1) / 3) * kC32RowPitch + ((g >> 1) % 3) * kC32PixPitch + 2 * ((g & 1) * 8 + j));
}
// Left to itself the compiler reads each A operand right in front of its MFMA and waits for it (one live register;
// volatile reads did not change that).  So a group's eight reads are ONE asm statement, and the wait is explicit: LDS returns
// in order, so `lgkmcnt(8)` behind the NEXT group's eight reads means this group's have landed.
template <int G>
__device__ __forceinline__ void c32_fetch(float (&dst)[8], unsigned addr) {
    asm volatile("ds_read_b32 %0, %8 offset:%9\n\tds_read_b32 %1, %8 offset:%10\n\tds_read_b32 %2, %8 offset:%11\n\t"
                 "ds_read_b32 %3, %8 offset:%12\n\tds_read_b32 %4, %8 offset:%13\n\tds_read_b32 %5, %8 offset:%14\n\t"
                 "ds_read_b32 %6, %8 offset:%15\n\tds_read_b32 %7, %8 offset:%16"
                 : "=&v"(dst[0]), "=&v"(dst[1]), "=&v"(dst[2]), "=&v"(dst[3]), "=&v"(dst[4]), "=&v"(dst[5]), "=&v"(dst[6]),
                   "=&v"(dst[7])
                 : "v"(addr), "n"(c32_off(G, 0)), "n"(c32_off(G, 1)), "n"(c32_off(G, 2)), "n"(c32_off(G, 3)), "n"(c32_off(G, 4)),
                   "n"(c32_off(G, 5)), "n"(c32_off(G, 6)), "n"(c32_off(G, 7)));
}
template <int G>
__device__ __forceinline__ void c32_groups(float (&abuf)[2][8], const float (&breg)[9][16], f32x16& acc, unsigned addr) {
    if constexpr (G < 18) {
        float(&cur)[8] = abuf[G & 1];
        if constexpr (G + 1 < 18) {
            c32_fetch<G + 1>(abuf[(G + 1) & 1], addr);
            asm volatile("s_waitcnt lgkmcnt(8)"
                         : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)"
                         : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]), "+v"(cur[3]), "+v"(cur[4]), "+v"(cur[5]), "+v"(cur[6]), "+v"(cur[7]));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[j], breg[G >> 1][(G & 1) * 8 + j], acc, 0, 0, 0);
        c32_groups<G + 1>(abuf, breg, acc, addr);
    }
}

// CHUNKED: y in the channel-chunked layout [B][4][Ho][Wo][8] the Winograd layers read (k_conv_wino.h) instead of channels-last
// RAW: the bare convolution of a TRAINING pass (BatchNorm follows: no bias, no ReLU), channels-last out, the weight read with its
// own element strides (so / si / sh / sw over [cout][cin][3][3]: a channels_last parameter as it is) and, with `transposed`, as
// the backward-data pass needs it: W'[ci][co][ky][kx] = W[co][ci][2 - ky][2 - kx]
// BN (RAW only): also accumulate the statistics of the BatchNorm behind the convolution - a template parameter (see k_conv_wino_b3.h)
template <bool POOL, bool CHUNKED = false, bool RAW = false, bool BN = false>
__global__ __launch_bounds__(256, 2) void k_conv3x3_c32(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                        int W, long so = 0, long si = 0, long sh = 0, long sw = 0, int transposed = 0,
                                                        double* __restrict__ bn_sums = nullptr) {
    // bn_sums (RAW only): sum z / sum z^2 per output channel for the BatchNorm behind the convolution (bn_epilogue.h).  A lane's
    // channel is the same for every tile of this persistent workgroup: fp32 partial sums per patch, fp64 running totals in
    // registers, ONE pair of atomics per lane when the workgroup has walked its tiles
    [[maybe_unused]] double bn_t1 = 0.0, bn_t2 = 0.0;
    extern __shared__ float halo[];  // [6][66][33 (+ row padding)]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int hl = lane >> 5, i = lane & 31;
    // B operands: the whole weight tensor, 144 registers per lane, fetched once per (persistent) workgroup
    float breg[9][16];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int cp = 0; cp < 16; ++cp) {
            if constexpr (RAW) {   // output channel i, input channel 2 cp + hl, tap (ky, kx) of the convolution being computed
                const int c = 2 * cp + hl, ky = tap / 3, kx = tap % 3;
                breg[tap][cp] = transposed ? w[(long)c * so + (long)i * si + (long)(2 - ky) * sh + (long)(2 - kx) * sw]
                                           : w[(long)i * so + (long)c * si + (long)ky * sh + (long)kx * sw];
            } else {
                breg[tap][cp] = w[((size_t)i * kC32 + 2 * cp + hl) * 9 + tap];
            }
        }
    // this lane's pixel inside a 2 x 16 patch: MFMA row i = 4 (col >> 1) + 2 row + (col & 1)
    const int pr = (i >> 1) & 1, pc = 2 * (i >> 2) + (i & 1);
    const float bj = RAW ? 0.f : bias[i];  // output channel j = lane & 31 of every accumulator register
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const int tiles_w = (W + kC32TileW - 1) / kC32TileW, tiles_h = (H + kC32TileH - 1) / kC32TileH;
    const int n_tiles = B * tiles_h * tiles_w;

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int tw = tile % tiles_w, th = (tile / tiles_w) % tiles_h, b = tile / (tiles_w * tiles_h);
        const int h0 = th * kC32TileH, w0 = tw * kC32TileW;
        __syncthreads();  // the previous tile's halo is consumed
        // stage the halo: rows h0 - 1 .. h0 + 4, columns w0 - 1 .. w0 + 64, zero outside the image
        // (3168 float4 = 12.4 per thread, fetched in two batches of 7 so that the loads of a batch are all in flight
        // together: one load - one store per iteration cost a memory latency each)
        constexpr int kElems = kC32HaloRows * (kC32TileW + 2) * 8, kBatch = 7;
#pragma unroll
        for (int e0 = 0; e0 < kElems; e0 += 256 * kBatch) {
            float4 v[kBatch];
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int e = e0 + u * 256 + (int)threadIdx.x;
                const int q = e & 7, pix = e >> 3;
                const int rr = pix / (kC32TileW + 2), cc = pix - rr * (kC32TileW + 2);
                const int hh = h0 - 1 + rr, ww = w0 - 1 + cc;
                v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < kElems && hh >= 0 && hh < H && ww >= 0 && ww < W) v[u] = x4[(((size_t)b * H + hh) * W + ww) * 8 + q];
            }
#pragma unroll
            for (int u = 0; u < kBatch; ++u) {
                const int e = e0 + u * 256 + (int)threadIdx.x;
                const int q = e & 7, pix = e >> 3;
                const int rr = pix / (kC32TileW + 2), cc = pix - rr * (kC32TileW + 2);
                if (e < kElems) *reinterpret_cast<float4*>(halo + rr * kC32RowPitch + cc * kC32PixPitch + 4 * q) = v[u];
            }
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t) {  // the wave's two patches: rows 2 t, 2 t + 1 of the tile
            const float* a0 = halo + (2 * t + pr) * kC32RowPitch + (16 * wv + pc) * kC32PixPitch + hl;
            f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            // 18 groups of 8 MFMAs (half a tap each), the A operands of group g + 1 read from LDS while group g runs
            float abuf[2][8];
            const unsigned a_addr = (unsigned)(uintptr_t)(lds_float*)a0;
            c32_fetch<0>(abuf[0], a_addr);
            c32_groups<0>(abuf, breg, acc, a_addr);
            // accumulator register r of this lane: MFMA row (r & 3) + 8 (r >> 2) + 4 hl, column = output channel lane & 31
            if constexpr (!POOL) {
                [[maybe_unused]] BnEpilogue bn = {acc[0], 0.f, 0.f, 0.f};
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row_i = (r & 3) + 8 * (r >> 2) + 4 * hl;
                    const int orow = h0 + 2 * t + ((row_i >> 1) & 1), ocol = w0 + 16 * wv + 2 * (row_i >> 2) + (row_i & 1);
                    if constexpr (RAW && BN) bn_epilogue_add(bn, acc[r], (orow < H && ocol < W) ? 1.f : 0.f);
                    if (orow < H && ocol < W)
                        y[CHUNKED ? ((((size_t)b * 4 + (i >> 3)) * H + orow) * W + ocol) * 8 + (i & 7)
                                  : (((size_t)b * H + orow) * W + ocol) * kC32 + i] = RAW ? acc[r] : fmaxf(acc[r] + bj, 0.f);
                }
                if constexpr (RAW && BN) {   // this patch's share, about zero, in fp64
                    const double K = (double)bn.k, n = (double)bn.n, S1 = (double)bn.s1, S2 = (double)bn.s2;
                    bn_t1 += S1 + n * K;
                    bn_t2 += S2 + 2.0 * K * S1 + n * K * K;
                }
            } else {
                const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
                const int prow = (h0 >> 1) + t;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    // the window's four pixels are registers 4 g .. 4 g + 3; pixels outside the image hold convolution values
                    // of the zero padding and must not win: compare only the ones inside
                    const int q = 2 * g + hl;  // pooled column inside the wave's patch
                    const int oc0 = w0 + 16 * wv + 2 * q, or0 = h0 + 2 * t;
                    float m = acc[4 * g];      // (or0, oc0) is inside whenever the pooled pixel exists
                    if (oc0 + 1 < W) m = fmaxf(m, acc[4 * g + 1]);
                    if (or0 + 1 < H) {
                        m = fmaxf(m, acc[4 * g + 2]);
                        if (oc0 + 1 < W) m = fmaxf(m, acc[4 * g + 3]);
                    }
                    const int pcol = (w0 >> 1) + 8 * wv + q;
                    if (prow < Ho && pcol < Wo && or0 < H && oc0 < W)
                        y[CHUNKED ? ((((size_t)b * 4 + (i >> 3)) * Ho + prow) * Wo + pcol) * 8 + (i & 7)
                                  : (((size_t)b * Ho + prow) * Wo + pcol) * kC32 + i] = fmaxf(m + bj, 0.f);
                }
            }
        }
    }
    if constexpr (RAW && BN) {
        if (bn_t2 > 0.0) {
            const int slot = (int)(blockIdx.x % (unsigned)bn_slots(kC32));
            atomicAdd(bn_sums + (size_t)slot * 2 * kC32 + i, bn_t1);
            atomicAdd(bn_sums + (size_t)slot * 2 * kC32 + kC32 + i, bn_t2);
        }
    }
}

// The bare 32 -> 32 convolution for the training step (forward: transposed = 0; backward-data on dz: transposed = 1), y and x
// channels-last [B, H, W, 32], the weight [32, 32, 3, 3] with element strides
static int conv3x3_c32_impl(const float* x, const float* weight, long stride_o, long stride_i, long stride_h, long stride_w,
                            int transposed, float* y, int batch, int height, int width, double* bn_sums, void* stream) {
    if (!x || !weight || !y) return fail(IRIS_E_INVALID, "iris_conv3x3_c32: NULL argument");
    if (batch <= 0 || height <= 0 || width <= 0) return fail(IRIS_E_INVALID, "iris_conv3x3_c32: empty tensor");
    if ((reinterpret_cast<uintptr_t>(x) & 15)) return fail(IRIS_E_INVALID, "iris_conv3x3_c32: x must be 16-byte aligned");
    int dev = 0, n_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    static std::atomic<unsigned> attr_set[64];
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c32<false, false, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kC32LdsBytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c32<false, false, true, true>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kC32LdsBytes));
        if (dev >= 0 && dev < 64) attr_set[dev].store(1u, std::memory_order_release);
    }
    const long long n_tiles = (long long)((width + kC32TileW - 1) / kC32TileW) * ((height + kC32TileH - 1) / kC32TileH) * batch;
    if (n_tiles >= 2147483647LL) return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_c32: too many tiles");
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
    const unsigned grid = (unsigned)std::min<long long>(n_tiles, 2LL * n_cu);
    if (bn_sums)
        k_conv3x3_c32<false, false, true, true><<<grid, 256, kC32LdsBytes, (hipStream_t)stream>>>(x, weight, nullptr, y, batch, height, width,
                                                                                                   stride_o, stride_i, stride_h, stride_w, transposed, bn_sums);
    else
        k_conv3x3_c32<false, false, true><<<grid, 256, kC32LdsBytes, (hipStream_t)stream>>>(x, weight, nullptr, y, batch, height, width, stride_o,
                                                                                             stride_i, stride_h, stride_w, transposed, nullptr);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_conv3x3_c32(const float* x, const float* weight, long stride_o, long stride_i, long stride_h, long stride_w,
                                int transposed, float* y, int batch, int height, int width, void* stream) {
    return conv3x3_c32_impl(x, weight, stride_o, stride_i, stride_h, stride_w, transposed, y, batch, height, width, nullptr, stream);
}
// the same + the statistics of the BatchNorm behind it (bn_sums_zeroed: DEVICE double [iris_bn_sums_len(32)], zero on entry; consumed
// by iris_bn_relu_apply_sums0 / iris_bn_relu_pool_apply_sums0)
extern "C" int iris_conv3x3_c32_bn(const float* x, const float* weight, long stride_o, long stride_i, long stride_h, long stride_w,
                                   float* y, int batch, int height, int width, double* bn_sums_zeroed, void* stream) {
    if (!bn_sums_zeroed) return fail(IRIS_E_INVALID, "iris_conv3x3_c32_bn: NULL argument");
    return conv3x3_c32_impl(x, weight, stride_o, stride_i, stride_h, stride_w, 0, y, batch, height, width, bn_sums_zeroed, stream);
}

extern "C" int iris_conv3x3_c32_bias_relu(const float* x, const float* weight, const float* bias, float* y, int batch, int height,
                                          int width, int pool, int out_chunked, void* stream) {
    if (!x || !weight || !bias || !y) return fail(IRIS_E_INVALID, "iris_conv3x3_c32_bias_relu: NULL argument");
    if (batch <= 0 || height <= 0 || width <= 0) return fail(IRIS_E_INVALID, "iris_conv3x3_c32_bias_relu: empty tensor");
    if ((reinterpret_cast<uintptr_t>(x) & 15)) return fail(IRIS_E_INVALID, "iris_conv3x3_c32_bias_relu: x must be 16-byte aligned");
    // per DEVICE (the attribute belongs to the function object of the current device) and safe from several threads: one
    // atomic flag per device ordinal; setting the attribute twice is harmless, skipping it on a second GPU is not
    int dev = 0, n_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    static std::atomic<unsigned> attr_set[64];
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c32<false>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kC32LdsBytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c32<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kC32LdsBytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c32<false, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kC32LdsBytes));
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(k_conv3x3_c32<true, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)kC32LdsBytes));
        if (dev >= 0 && dev < 64) attr_set[dev].store(1u, std::memory_order_release);
    }
    const long long n_tiles = (long long)((width + kC32TileW - 1) / kC32TileW) * ((height + kC32TileH - 1) / kC32TileH) * batch;
    if (n_tiles >= 2147483647LL) return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_c32_bias_relu: too many tiles");
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
    const unsigned grid = (unsigned)std::min<long long>(n_tiles, 2LL * n_cu);  // persistent: two workgroups per CU walk the tiles
    hipStream_t s = (hipStream_t)stream;
    if (pool && out_chunked) k_conv3x3_c32<true, true><<<grid, 256, kC32LdsBytes, s>>>(x, weight, bias, y, batch, height, width);
    else if (pool) k_conv3x3_c32<true><<<grid, 256, kC32LdsBytes, s>>>(x, weight, bias, y, batch, height, width);
    else if (out_chunked) k_conv3x3_c32<false, true><<<grid, 256, kC32LdsBytes, s>>>(x, weight, bias, y, batch, height, width);
    else k_conv3x3_c32<false><<<grid, 256, kC32LdsBytes, s>>>(x, weight, bias, y, batch, height, width);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
