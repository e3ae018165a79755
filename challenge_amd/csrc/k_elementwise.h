// k_elementwise.h -- min-max / log, normalize, magnitude-phase, mask apply, adaptive gradient clipping.
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// K4/K5: min-max (+ log): partial reduce, then apply
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax_partial(const float* x, float* partial, size_t row_len,
                                                        int n_part) {
    __shared__ float red[32];
    const int row = blockIdx.y, part = blockIdx.x;
    const float* p = x + (size_t)row * row_len;
    const size_t beg = (size_t)part * kChunk, end = min(beg + (size_t)kChunk, row_len);
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
        const float v = p[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    block_minmax(mn, mx, red);
    if (threadIdx.x == 0) {
        partial[((size_t)row * n_part + part) * 2 + 0] = mn;
        partial[((size_t)row * n_part + part) * 2 + 1] = mx;
    }
}

// One float4 per thread (kApply elements per block): at a few MB per launch the kernel is bound
// by instruction latency, not bandwidth, so it wants many short waves.  The element load is
// issued before the partials are folded (two independent round trips overlap).
constexpr int kApply = 1024;
__global__ __launch_bounds__(256) void k_minmax_log_apply(float* x, const float* partial, int n_part,
                                                          size_t row_len, int do_minmax, int do_log,
                                                          float eps_div, float eps_log) {
    __shared__ float red[32];
    const int row = blockIdx.y;
    float* p = x + (size_t)row * row_len;
    const size_t beg = (size_t)blockIdx.x * kApply, end = min(beg + (size_t)kApply, row_len);
    const bool vec = ((row_len & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);  // uniform
    const size_t iv = beg + 4 * (size_t)threadIdx.x;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec && iv < end) v = *reinterpret_cast<const float4*>(p + iv);

    float mn = 0.f, inv = 1.f;
    if (do_minmax) {
        float lo = INFINITY, hi = -INFINITY;
        if (n_part <= 256) {  // few partials: every wave folds them itself (no barrier)
            const float2* pr = reinterpret_cast<const float2*>(partial) + (size_t)row * n_part;
            for (int i = threadIdx.x & 63; i < n_part; i += 64) {
                const float2 q = pr[i];
                lo = fminf(lo, q.x);
                hi = fmaxf(hi, q.y);
            }
            lo = wave_min(lo);
            hi = wave_max(hi);
        } else {
            for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
                lo = fminf(lo, partial[((size_t)row * n_part + i) * 2 + 0]);
                hi = fmaxf(hi, partial[((size_t)row * n_part + i) * 2 + 1]);
            }
            block_minmax(lo, hi, red);
        }
        mn = lo;
        inv = 1.0f / fmaxf(hi - lo, eps_div);
    }
    if (vec) {
        if (iv < end) {
            float* e = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int j = 0; j < 4; ++j) e[j] = minmax_log_value(e[j], mn, inv, do_minmax, do_log, eps_log);
            *reinterpret_cast<float4*>(p + iv) = v;
        }
    } else {
        for (size_t i = beg + threadIdx.x; i < end; i += blockDim.x) p[i] = minmax_log_value(p[i], mn, inv, do_minmax, do_log, eps_log);
    }
}

// ---------------------------------------------------------------------------
// normalize: partial sums of squares, then scale
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sumsq_partial(const float* x, float* partial, size_t row_len, int n_part) {
    __shared__ float red[4];
    const int row = blockIdx.y, part = blockIdx.x;
    const float* p = x + (size_t)row * row_len;
    const size_t beg = (size_t)part * kChunk, end = min(beg + (size_t)kChunk, row_len);
    float s = 0.f;
    for (size_t i = beg + threadIdx.x; i < end; i += blockDim.x) s = fmaf(p[i], p[i], s);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)row * n_part + part] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void k_normalize_apply(const float* x, float* out, const float* partial,
                                                         int n_part, size_t row_len) {
    __shared__ float red[4];
    const int row = blockIdx.y;
    float s = 0.f;
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) s += partial[(size_t)row * n_part + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    s = (red[0] + red[1]) + (red[2] + red[3]);
    const float rms10 = sqrtf(s / (float)row_len) * 10.0f;
    const float* p = x + (size_t)row * row_len;
    float* o = out + (size_t)row * row_len;
    const size_t beg = (size_t)blockIdx.x * kChunk, end = min(beg + (size_t)kChunk, row_len);
    for (size_t i = beg + threadIdx.x; i < end; i += blockDim.x) o[i] = p[i] / rms10;
}

// ---------------------------------------------------------------------------
// elementwise: magnitude/phase, mask apply
// ---------------------------------------------------------------------------
__global__ void k_complex_to_magphase(const float* in, float* out, size_t n_outer, int C) {
    const size_t total = n_outer * (size_t)C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i / C;
        const int c = (int)(i - o * C);
        const float re = in[o * 2 * C + c], im = in[o * 2 * C + C + c];
        out[o * 2 * C + c] = sqrtf(re * re + im * im);
        out[o * 2 * C + C + c] = atan2f(im, re);
    }
}

__global__ void k_magphase_to_complex(const float* in, float* out, size_t n_outer, int C) {
    const size_t total = n_outer * (size_t)C;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t o = i / C;
        const int c = (int)(i - o * C);
        const float mag = in[o * 2 * C + c], ph = in[o * 2 * C + C + c];
        float s, co;
        sincosf(ph, &s, &co);
        out[o * 2 * C + c] = mag * co;
        out[o * 2 * C + C + c] = mag * s;
    }
}

template <typename T>
__global__ void k_mask_apply(T* x, size_t n_outer, size_t axis_len, size_t n_inner, const int* bands, int n_bands,
                             size_t outer_per_group) {
    const size_t total = n_outer * axis_len * n_inner;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total;
         i += (size_t)gridDim.x * blockDim.x) {
        const size_t oa = i / n_inner;
        const size_t o = oa / axis_len;
        const int ax = (int)(oa - o * axis_len);
        const int* bd = bands + (o / outer_per_group) * (size_t)n_bands * 2;
        if (in_bands(bd, n_bands, ax)) x[i] = T(0);
    }
}

// ---------------------------------------------------------------------------
// adaptive gradient clipping + clipvalue, one wave per output unit (row)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_agc_clip(const iris_agc_row* rows, size_t n_rows, float clip_factor,
                                                  float eps, float clipvalue) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * 4;
    for (size_t r = wave; r < n_rows; r += n_waves) {
        const float* p = rows[r].param;
        float* g = rows[r].grad;
        const long len = rows[r].len;
        float sp = 0.f, sg = 0.f;
        const bool vec = ((len & 3) == 0) && (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g)) & 15) == 0);
        if (vec) {
            for (long i = 4 * lane; i < len; i += 4 * kWave) {
                const float4 a = *reinterpret_cast<const float4*>(p + i);
                const float4 b = *reinterpret_cast<const float4*>(g + i);
                sp += a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w;
                sg += b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w;
            }
        } else {
            for (long i = lane; i < len; i += kWave) {
                sp += p[i] * p[i];
                sg += g[i] * g[i];
            }
        }
        const float p_norm = sqrtf(wave_sum(sp)), g_norm = sqrtf(wave_sum(sg));
        const float max_norm = fmaxf(p_norm, eps) * clip_factor;
        const float scale = g_norm < max_norm ? 1.0f : max_norm / fmaxf(g_norm, 1e-6f);
        const bool clamp = clipvalue > 0.f;
        if (scale == 1.0f && !clamp) continue;  // wave-uniform
        if (vec) {
            for (long i = 4 * lane; i < len; i += 4 * kWave) {
                float4 b = *reinterpret_cast<float4*>(g + i);
                b.x *= scale; b.y *= scale; b.z *= scale; b.w *= scale;
                if (clamp) {
                    b.x = fminf(fmaxf(b.x, -clipvalue), clipvalue);
                    b.y = fminf(fmaxf(b.y, -clipvalue), clipvalue);
                    b.z = fminf(fmaxf(b.z, -clipvalue), clipvalue);
                    b.w = fminf(fmaxf(b.w, -clipvalue), clipvalue);
                }
                *reinterpret_cast<float4*>(g + i) = b;
            }
        } else {
            for (long i = lane; i < len; i += kWave) {
                float v = g[i] * scale;
                if (clamp) v = fminf(fmaxf(v, -clipvalue), clipvalue);
                g[i] = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// inference epilogue of a folded Conv2D + BatchNorm + ReLU (sj_train.py:191-201): y = max(x + bias[c], 0) in place on a
// channels-last tensor [n_outer, C]; POOL: the 2x2 / stride-2 'same' max-pool of the block fused behind it
// ---------------------------------------------------------------------------
// Column of a grid-stride walk over [rows][C4] without a 64-bit modulo per element: c(i + stride) = c(i) + stride mod C4.
struct ColumnWalk {
    int c, step, C4;
    __device__ __forceinline__ ColumnWalk(size_t first, size_t stride, int n) : c((int)(first % n)), step((int)(stride % n)), C4(n) {}
    __device__ __forceinline__ void next() {
        c += step;
        if (c >= C4) c -= C4;
    }
};

// One 2x2 / stride-2 'same' pooling window of a channels-last tensor [B, H, W, C4 float4].
struct PoolWindow {
    size_t base;     // float4 index of the window's (h0, w0) element
    size_t down;     // float4 offset of the row below
    bool w1, h1;     // the window has a second column / row
};
// pooled row r = (b Ho + ho) Wo + wo (32-bit: the host checks the sizes), float4 column c
__device__ __forceinline__ PoolWindow pool_window(unsigned r, int c, int H, int W, int Ho, int Wo, int C4) {
    const unsigned t = r / (unsigned)Wo, wo = r - t * (unsigned)Wo;
    const unsigned b = t / (unsigned)Ho, ho = t - b * (unsigned)Ho;
    PoolWindow w;
    w.base = (((size_t)b * H + 2 * ho) * W + 2 * wo) * C4 + c;
    w.down = (size_t)W * C4;
    w.w1 = 2 * wo + 1 < (unsigned)W;
    w.h1 = 2 * ho + 1 < (unsigned)H;
    return w;
}
__global__ __launch_bounds__(256) void k_bias_relu(float* x, const float* bias, size_t n_vec4, int C4) {
    const float4* b4 = reinterpret_cast<const float4*>(bias);
    float4* x4 = reinterpret_cast<float4*>(x);
    ColumnWalk col(blockIdx.x * (size_t)blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x, C4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_vec4; i += (size_t)gridDim.x * blockDim.x, col.next()) {
        float4 v = x4[i];
        const float4 b = b4[col.c];
        v.x = fmaxf(v.x + b.x, 0.f);
        v.y = fmaxf(v.y + b.y, 0.f);
        v.z = fmaxf(v.z + b.z, 0.f);
        v.w = fmaxf(v.w + b.w, 0.f);
        x4[i] = v;
    }
}

// x [B, H, W, C] -> y [B, ceil(H/2), ceil(W/2), C]: y = maxpool2x2(relu(x + bias)) = relu(max over the window of x + bias)
__global__ __launch_bounds__(256) void k_bias_relu_pool(const float* x, const float* bias, float* y, int B, int H, int W, int C4) {
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const unsigned total = (unsigned)B * Ho * Wo * C4;  // < 2^31 (host check)
    const float4* b4 = reinterpret_cast<const float4*>(bias);
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i / (unsigned)C4;
        const int c = (int)(i - r * (unsigned)C4);
        const PoolWindow w = pool_window(r, c, H, W, Ho, Wo, C4);
        float4 m = x4[w.base];
        auto take = [&](size_t off) {
            const float4 v = x4[w.base + off];
            m.x = fmaxf(m.x, v.x);
            m.y = fmaxf(m.y, v.y);
            m.z = fmaxf(m.z, v.z);
            m.w = fmaxf(m.w, v.w);
        };
        if (w.w1) take(C4);
        if (w.h1) take(w.down);
        if (w.h1 && w.w1) take(w.down + C4);
        const float4 bb = b4[c];
        m.x = fmaxf(m.x + bb.x, 0.f);
        m.y = fmaxf(m.y + bb.y, 0.f);
        m.z = fmaxf(m.z + bb.z, 0.f);
        m.w = fmaxf(m.w + bb.w, 0.f);
        y4[i] = m;
    }
}

// The same epilogues for a CONTIGUOUS (NCHW) activation - block 1 of the CRNN runs its 32 -> 32 convolution in that
// layout (MIOpen's NCHW solvers take 415 us for it against 683 us for the NHWC implicit GEMM; every other layer is
// faster in NHWC): x [B, C, inner] in place, and the pooling variant reads NCHW and writes the pooled tensor
// channels-last [B, Ho, Wo, C] through an LDS transpose (reads coalesced along w, writes along c).
__global__ __launch_bounds__(256) void k_bias_relu_nchw(float* x, const float* bias, size_t n_vec4, size_t inner4, int C) {
    float4* x4 = reinterpret_cast<float4*>(x);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_vec4; i += (size_t)gridDim.x * blockDim.x) {
        // (32-bit quotients when they fit: a 64-bit division costs more than the rest of the loop body)
        const float b = n_vec4 <= 0xffffffffull ? bias[((unsigned)i / (unsigned)inner4) % (unsigned)C] : bias[(i / inner4) % C];
        float4 v = x4[i];
        v.x = fmaxf(v.x + b, 0.f);
        v.y = fmaxf(v.y + b, 0.f);
        v.z = fmaxf(v.z + b, 0.f);
        v.w = fmaxf(v.w + b, 0.f);
        x4[i] = v;
    }
}

// tile_w pooled columns per block (<= 64: one lane each), sized by the host so that the tile fits 48 KB of LDS
__global__ __launch_bounds__(256) void k_bias_relu_pool_nchw(const float* x, const float* bias, float* y, int H, int W, int C,
                                                             int tile_w) {
    extern __shared__ float tile[];  // [tile_w][C + 1]
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const int wo0 = blockIdx.x * tile_w, ho = blockIdx.y, b = blockIdx.z;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wo = wo0 + lane, h0 = 2 * ho, w0 = 2 * wo;
    const bool h1 = h0 + 1 < H, w1 = w0 + 1 < W, in = wo < Wo && lane < tile_w;
    // even width and an 8-byte aligned tensor: a window row is ONE 8-byte load, a wave reads 512 contiguous bytes per row
    const bool pairs = ((W & 1) == 0) && ((reinterpret_cast<uintptr_t>(x) & 7) == 0);
    for (int c = wv; c < C; c += 4) {
        const float* p = x + (((size_t)b * C + c) * H + h0) * W + w0;
        float m = -INFINITY;
        if (in) {
            if (pairs) {
                const float2 r0 = *reinterpret_cast<const float2*>(p);
                m = fmaxf(r0.x, r0.y);
                if (h1) {
                    const float2 r1 = *reinterpret_cast<const float2*>(p + W);
                    m = fmaxf(m, fmaxf(r1.x, r1.y));
                }
            } else {
                m = p[0];
                if (w1) m = fmaxf(m, p[1]);
                if (h1) {
                    m = fmaxf(m, p[W]);
                    if (w1) m = fmaxf(m, p[W + 1]);
                }
            }
            m = fmaxf(m + bias[c], 0.f);
        }
        if (lane < tile_w) tile[lane * (C + 1) + c] = m;
    }
    __syncthreads();
    const int n_w = min(tile_w, Wo - wo0);
    float* o = y + (((size_t)b * Ho + ho) * Wo + wo0) * C;  // [n_w][C] contiguous
    for (int i = threadIdx.x; i < n_w * C; i += blockDim.x) o[i] = tile[(i / C) * (C + 1) + (i % C)];
}

// ---------------------------------------------------------------------------
// Training-mode Conv2D bias + BatchNormalization + ReLU of ConvMPBlock (sj_train.py:191-201) in two passes each way, on the
// channels-last convolution output z [rows = N H W, C] (the convolution itself stays MIOpen; its bias is NOT added by a
// separate pass: batch normalisation subtracts the batch mean, so y does not depend on the bias - it only shifts the
// running mean, which is accounted for here - and its gradient is identically zero):
//   forward   k_bn_stats        per-channel sum, sum of squares of z - K, K = row 0 of the channel (fp32 per thread over <= 32 rows, fp64 atomics per block)
//             k_bn_relu_apply   y = max(gamma (z - mean) rstd + beta, 0); block 0 updates the running statistics
//   backward  k_bn_reduce<true>      g = dy [y > 0]; sum g, sum g xhat per channel (same reduction scheme)
//             k_bn_relu_bwd_dx       dz = gamma rstd (g - sum_g / M - xhat sum_gx / M); block 0 writes dgamma, dbeta
// Instead of bias add, mean / variance, normalise, ReLU (7 passes over the activation) and ReLU', dscale / dbias, dx, bias
// gradient (9 passes): 3 forward (read, read + write) and 5 backward (z and dy read twice, dz written; y is never read).  Thread layout: a block of 256 threads covers kBnRows rows x (C / 4) float4 columns.
// ---------------------------------------------------------------------------
#ifndef IRIS_BN_ROWS
#define IRIS_BN_ROWS 32
#endif
#ifndef IRIS_BN_POOL_ROWS
#define IRIS_BN_POOL_ROWS 8
#endif
#include "bn_epilogue.h"   // kBnSlots, bn_slots(), and the statistics a convolution's epilogue accumulates
__device__ __forceinline__ double bn_sum(const double* sums, int C, int i) {
    const int n = bn_slots(C);
    double v = 0.0;
    for (int s = 0; s < n; ++s) v += sums[(size_t)s * 2 * C + i];
    return v;
}
constexpr int kBnRows = IRIS_BN_ROWS;           // rows per thread-row pass
constexpr int kBnPoolRows = IRIS_BN_POOL_ROWS;  // pooled rows (windows of four) per thread-row pass: 4x the blocks of the same tensor

// (backward: the ReLU mask [y > 0] is recomputed from z with the forward's own scale / shift - bit-identical to testing
// the stored y - so that y is not read at all)
template <bool BWD>
__global__ __launch_bounds__(256) void k_bn_reduce(const float* z, const float* dy, size_t rows, int C4, const float* mean,
                                                   const float* rstd, const float* gamma, const float* beta,
                                                   double* sums /*[2][4 C4]*/) {
    // threads: tx = threadIdx.x % C4g walks the float4 columns (C4g = min(C4, 256) columns per pass), ty the rows
    extern __shared__ float red[];  // [ty][2][4 * cols]
    const int cols = min(C4, 256), tys = 256 / cols;
    const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
    const float4* z4 = reinterpret_cast<const float4*>(z);
    const float4* d4 = reinterpret_cast<const float4*>(dy);
    for (int c0 = 0; c0 < C4; c0 += cols) {
        const int c = c0 + tx;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        float4 mu = s0, rs = s0, sc = s0, sh = s0;
        // forward statistics are SHIFTED sums, sum (z - K) and sum (z - K)^2 with K = the channel's value in row 0 (the same K
        // in every block; the consumers read it back from z): E[z^2] - E[z]^2 on raw fp32 partial sums loses the variance of
        // a channel whose |mean| is far above its spread (clamped to 0: rstd = 1 / sqrt(eps)); shifted, the partial sums are
        // of the size of the spread itself
        float4 kz = s0;
        if (!BWD && c < C4) kz = z4[c];
        if (BWD && c < C4) {
            mu = reinterpret_cast<const float4*>(mean)[c];
            rs = reinterpret_cast<const float4*>(rstd)[c];
            const float4 ga = reinterpret_cast<const float4*>(gamma)[c], be = reinterpret_cast<const float4*>(beta)[c];
            sc = make_float4(ga.x * rs.x, ga.y * rs.y, ga.z * rs.z, ga.w * rs.w);  // as k_bn_relu_apply forms them
            sh = make_float4(be.x - mu.x * sc.x, be.y - mu.y * sc.y, be.z - mu.z * sc.z, be.w - mu.w * sc.w);
        }
        const size_t r_begin = (size_t)blockIdx.x * kBnRows * tys, r_end = min(r_begin + (size_t)kBnRows * tys, rows);
        if (c < C4 && ty < tys) {
#pragma unroll 4
            for (size_t r = r_begin + ty; r < r_end; r += tys) {
                const float4 v = z4[r * C4 + c];
                if constexpr (!BWD) {
                    const float ux = v.x - kz.x, uy = v.y - kz.y, uz = v.z - kz.z, uw = v.w - kz.w;
                    s0.x += ux; s0.y += uy; s0.z += uz; s0.w += uw;
                    s1.x = fmaf(ux, ux, s1.x); s1.y = fmaf(uy, uy, s1.y); s1.z = fmaf(uz, uz, s1.z); s1.w = fmaf(uw, uw, s1.w);
                } else {
                    const float4 d = d4[r * C4 + c];
                    const float gx = fmaf(v.x, sc.x, sh.x) > 0.f ? d.x : 0.f, gy = fmaf(v.y, sc.y, sh.y) > 0.f ? d.y : 0.f;
                    const float gz = fmaf(v.z, sc.z, sh.z) > 0.f ? d.z : 0.f, gw = fmaf(v.w, sc.w, sh.w) > 0.f ? d.w : 0.f;
                    s0.x += gx; s0.y += gy; s0.z += gz; s0.w += gw;
                    s1.x = fmaf(gx, (v.x - mu.x) * rs.x, s1.x); s1.y = fmaf(gy, (v.y - mu.y) * rs.y, s1.y);
                    s1.z = fmaf(gz, (v.z - mu.z) * rs.z, s1.z); s1.w = fmaf(gw, (v.w - mu.w) * rs.w, s1.w);
                }
            }
        }
        __syncthreads();
        if (ty < tys) {
            float* p = red + ((size_t)ty * 2 * cols + tx) * 4;
            p[0] = s0.x; p[1] = s0.y; p[2] = s0.z; p[3] = s0.w;
            float* q = p + (size_t)cols * 4;
            q[0] = s1.x; q[1] = s1.y; q[2] = s1.z; q[3] = s1.w;
        }
        __syncthreads();
        // fold the tys row-threads: 2 * 4 * cols values, one per thread (cols <= 256 -> up to 2048 values: loop)
        for (int i = threadIdx.x; i < 2 * 4 * cols; i += blockDim.x) {
            const int which = i / (4 * cols), j = i - which * 4 * cols;  // j = tx * 4 + component
            double acc = 0.0;
            for (int t = 0; t < tys; ++t) acc += (double)red[((size_t)t * 2 * cols) * 4 + (size_t)which * cols * 4 + j];
            const int cc = c0 * 4 + j;
            if (cc < 4 * C4) atomicAdd(sums + (size_t)(blockIdx.x % bn_slots(4 * C4)) * 8 * C4 + (size_t)which * 4 * C4 + cc, acc);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_bn_relu_apply(const float* z, float* y, size_t n_vec4, int C4, double inv_m, double unbias,
                                                       const double* sums, const float* gamma, const float* beta,
                                                       const float* conv_bias, float eps, float momentum, float* running_mean,
                                                       float* running_var, float* save_mean, float* save_rstd, int sums_about_zero) {
    extern __shared__ float coef[];  // [2][C]: scale = gamma rstd, shift = beta - mean scale  (y = max(z scale + shift, 0))
    const int C = 4 * C4;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        // sums are of (z - K), K = z[row 0][c] (k_bn_reduce) - or K = 0 when a convolution's epilogue accumulated them in fp64
        // (bn_epilogue_flush): mean = K + E[z - K], var = E[(z - K)^2] - E[z - K]^2
        const double ms = bn_sum(sums, C, c) * inv_m, var = fmax(bn_sum(sums, C, C + c) * inv_m - ms * ms, 0.0);
        const double m = (sums_about_zero ? 0.0 : (double)z[c]) + ms;
        const float mu = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
        coef[c] = gamma[c] * rs;
        coef[C + c] = beta[c] - mu * (gamma[c] * rs);
        if (blockIdx.x == 0) {  // statistics for backward and the running estimates (unbiased variance, Keras / torch rule)
            save_mean[c] = mu;
            save_rstd[c] = rs;
            const float bias = conv_bias ? conv_bias[c] : 0.f;  // the convolution's bias only moves the mean
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (mu + bias);
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
        }
    }
    __syncthreads();
    const float4* z4 = reinterpret_cast<const float4*>(z);
    float4* y4 = reinterpret_cast<float4*>(y);
    const float4* sc4 = reinterpret_cast<const float4*>(coef);
    const float4* sh4 = reinterpret_cast<const float4*>(coef + C);
    ColumnWalk col(blockIdx.x * (size_t)blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x, C4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_vec4; i += (size_t)gridDim.x * blockDim.x, col.next()) {
        const int c = col.c;
        const float4 v = z4[i], sc = sc4[c], sh = sh4[c];
        float4 r;
        r.x = fmaxf(fmaf(v.x, sc.x, sh.x), 0.f);
        r.y = fmaxf(fmaf(v.y, sc.y, sh.y), 0.f);
        r.z = fmaxf(fmaf(v.z, sc.z, sh.z), 0.f);
        r.w = fmaxf(fmaf(v.w, sc.w, sh.w), 0.f);
        y4[i] = r;
    }
}

__global__ __launch_bounds__(256) void k_bn_relu_bwd_dx(const float* z, const float* dy, float* dz, size_t n_vec4, int C4,
                                                        float inv_m, const float* mean, const float* rstd, const float* gamma,
                                                        const float* beta, const double* sums, float* dgamma, float* dbeta) {
    // dz = a g + b z + d  with  a = gamma rstd,  b = -a rstd sum_gx / M,  d = -a sum_g / M - b mean   (per channel);
    // g = dy where z a + (beta - mean a) > 0 (the forward's expression: the stored y is not read)
    extern __shared__ float coef[];  // [4][C]
    const int C = 4 * C4;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float sg = (float)bn_sum(sums, C, c), sgx = (float)bn_sum(sums, C, C + c);
        const float a = gamma[c] * rstd[c], b = -a * rstd[c] * sgx * inv_m;
        coef[c] = a;
        coef[C + c] = b;
        coef[2 * C + c] = -a * sg * inv_m - b * mean[c];
        coef[3 * C + c] = beta[c] - mean[c] * a;
        if (blockIdx.x == 0) {
            dbeta[c] = sg;
            dgamma[c] = sgx;
        }
    }
    __syncthreads();
    const float4* z4 = reinterpret_cast<const float4*>(z);
    const float4* d4 = reinterpret_cast<const float4*>(dy);
    float4* o4 = reinterpret_cast<float4*>(dz);
    const float4* a4 = reinterpret_cast<const float4*>(coef);
    const float4* b4 = reinterpret_cast<const float4*>(coef + C);
    const float4* c4 = reinterpret_cast<const float4*>(coef + 2 * C);
    const float4* h4 = reinterpret_cast<const float4*>(coef + 3 * C);
    ColumnWalk col(blockIdx.x * (size_t)blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x, C4);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n_vec4; i += (size_t)gridDim.x * blockDim.x, col.next()) {
        const int c = col.c;
        const float4 v = z4[i], d = d4[i], a = a4[c], b = b4[c], k = c4[c], h = h4[c];
        float4 r;
        r.x = fmaf(a.x, fmaf(v.x, a.x, h.x) > 0.f ? d.x : 0.f, fmaf(b.x, v.x, k.x));
        r.y = fmaf(a.y, fmaf(v.y, a.y, h.y) > 0.f ? d.y : 0.f, fmaf(b.y, v.y, k.y));
        r.z = fmaf(a.z, fmaf(v.z, a.z, h.z) > 0.f ? d.z : 0.f, fmaf(b.z, v.z, k.z));
        r.w = fmaf(a.w, fmaf(v.w, a.w, h.w) > 0.f ? d.w : 0.f, fmaf(b.w, v.w, k.w));
        o4[i] = r;
    }
}

// ---------------------------------------------------------------------------
// The last convolution of a ConvMPBlock is followed by MaxPool 2x2 / stride 2 / 'same' (sj_train.py:191-201).  With the
// pooling inside the BatchNorm + ReLU passes the full-size y is never written and the full-size dy never exists:
//   forward   k_bn_relu_pool_apply      p = max over the window of max(z scale + shift, 0)         (reads z, writes p = 1/4)
//   backward  k_bn_pool_bwd_reduce      g = dp at the window's first maximum if that is > 0; sum g, sum g xhat
//             k_bn_relu_pool_bwd_dx     dz = a g + b z + d over every element of the window     (reads z + dp, writes dz)
// The winner of a window is its first maximum in (h, w) scan order, which is what max_pool2d's index rule picks; windows
// whose maximum is 0 pass no gradient (ReLU' there is 0 whichever element is chosen).
// z [B, H, W, C] channels-last, p / dp [B, ceil(H/2), ceil(W/2), C]; a thread owns one float4 of one window.
// ---------------------------------------------------------------------------
// y of one element, and the running first maximum (value, its z, its slot)
__device__ __forceinline__ void pool_take(float z, float sc, float sh, int slot, float& best, float& zbest, int& sel) {
    const float y = fmaxf(fmaf(z, sc, sh), 0.f);
    if (slot == 0 || y > best) {
        best = y;
        zbest = z;
        sel = slot;
    }
}

__global__ __launch_bounds__(256) void k_bn_relu_pool_apply(const float* z, float* p, int B, int H, int W, int C4, double inv_m,
                                                            double unbias, const double* sums, const float* gamma, const float* beta,
                                                            const float* conv_bias, float eps, float momentum, float* running_mean,
                                                            float* running_var, float* save_mean, float* save_rstd, int sums_about_zero) {
    extern __shared__ float coef[];  // [2][C], formed exactly as k_bn_relu_apply forms them
    const int C = 4 * C4;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        // sums are of (z - K), K = z[row 0][c] (k_bn_reduce) or 0 (a convolution's epilogue): mean = K + E[z - K], var = E[(z - K)^2] - E[z - K]^2
        const double ms = bn_sum(sums, C, c) * inv_m, var = fmax(bn_sum(sums, C, C + c) * inv_m - ms * ms, 0.0);
        const double m = (sums_about_zero ? 0.0 : (double)z[c]) + ms;
        const float mu = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
        coef[c] = gamma[c] * rs;
        coef[C + c] = beta[c] - mu * (gamma[c] * rs);
        if (blockIdx.x == 0) {
            save_mean[c] = mu;
            save_rstd[c] = rs;
            const float bias = conv_bias ? conv_bias[c] : 0.f;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (mu + bias);
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
        }
    }
    __syncthreads();
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const unsigned total = (unsigned)B * Ho * Wo * C4;  // < 2^31 (host check)
    const float4* z4 = reinterpret_cast<const float4*>(z);
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* sc4 = reinterpret_cast<const float4*>(coef);
    const float4* sh4 = reinterpret_cast<const float4*>(coef + C);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i / (unsigned)C4;
        const int c = (int)(i - r * (unsigned)C4);
        const PoolWindow w = pool_window(r, c, H, W, Ho, Wo, C4);
        const float4 sc = sc4[c], sh = sh4[c];
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);  // y >= 0
        auto take = [&](size_t off) {
            const float4 v = z4[w.base + off];
            m.x = fmaxf(m.x, fmaf(v.x, sc.x, sh.x));
            m.y = fmaxf(m.y, fmaf(v.y, sc.y, sh.y));
            m.z = fmaxf(m.z, fmaf(v.z, sc.z, sh.z));
            m.w = fmaxf(m.w, fmaf(v.w, sc.w, sh.w));
        };
        take(0);
        if (w.w1) take(C4);
        if (w.h1) take(w.down);
        if (w.h1 && w.w1) take(w.down + C4);
        p4[i] = m;
    }
}

// the four elements of a window as zz[slot][component] (missing ones repeat element 0 and can never win: `>` is strict);
// plain float arrays with constant indices only: they stay in registers (indexing the float4s through a pointer put them in
// scratch)
__device__ __forceinline__ void pool_load(const float4* z4, const PoolWindow& w, int C4, float (&zz)[4][4]) {
    const float4 v0 = z4[w.base];
    const float4 v1 = w.w1 ? z4[w.base + C4] : v0;
    const float4 v2 = w.h1 ? z4[w.base + w.down] : v0;
    const float4 v3 = (w.h1 && w.w1) ? z4[w.base + w.down + C4] : v0;
    zz[0][0] = v0.x; zz[0][1] = v0.y; zz[0][2] = v0.z; zz[0][3] = v0.w;
    zz[1][0] = v1.x; zz[1][1] = v1.y; zz[1][2] = v1.z; zz[1][3] = v1.w;
    zz[2][0] = v2.x; zz[2][1] = v2.y; zz[2][2] = v2.z; zz[2][3] = v2.w;
    zz[3][0] = v3.x; zz[3][1] = v3.y; zz[3][2] = v3.z; zz[3][3] = v3.w;
}

__global__ __launch_bounds__(256) void k_bn_pool_bwd_reduce(const float* z, const float* dp, int B, int H, int W, int C4,
                                                            const float* mean, const float* rstd, const float* gamma,
                                                            const float* beta, double* sums /*[2][4 C4]*/) {
    // same thread layout and reduction as k_bn_reduce, over the POOLED rows
    extern __shared__ float red[];
    const int cols = min(C4, 256), tys = 256 / cols;
    const int tx = threadIdx.x % cols, ty = threadIdx.x / cols;
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const size_t rows = (size_t)B * Ho * Wo;
    const float4* z4 = reinterpret_cast<const float4*>(z);
    const float4* d4 = reinterpret_cast<const float4*>(dp);
    for (int c0 = 0; c0 < C4; c0 += cols) {
        const int c = c0 + tx;
        float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
        float mu[4] = {0.f, 0.f, 0.f, 0.f}, rs[4] = {0.f, 0.f, 0.f, 0.f}, sc[4] = {0.f, 0.f, 0.f, 0.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (c < C4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                mu[k] = mean[4 * c + k];
                rs[k] = rstd[4 * c + k];
                sc[k] = gamma[4 * c + k] * rs[k];
                sh[k] = beta[4 * c + k] - mu[k] * sc[k];
            }
        }
        const size_t r_begin = (size_t)blockIdx.x * kBnPoolRows * tys, r_end = min(r_begin + (size_t)kBnPoolRows * tys, rows);
        if (c < C4 && ty < tys) {
#pragma unroll 2
            for (size_t r = r_begin + ty; r < r_end; r += tys) {
                const PoolWindow w = pool_window((unsigned)r, c, H, W, Ho, Wo, C4);
                float v[4][4];
                pool_load(z4, w, C4, v);
                const float4 d = d4[r * C4 + c];
                const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float best = 0.f, zb = 0.f;
                    int sel = 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pool_take(v[j][k], sc[k], sh[k], j, best, zb, sel);
                    const float g = best > 0.f ? dd[k] : 0.f;
                    s0[k] += g;
                    s1[k] = fmaf(g, (zb - mu[k]) * rs[k], s1[k]);
                }
            }
        }
        __syncthreads();
        if (ty < tys) {
            float* pp = red + ((size_t)ty * 2 * cols + tx) * 4;
            float* q = pp + (size_t)cols * 4;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                pp[k] = s0[k];
                q[k] = s1[k];
            }
        }
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * 4 * cols; i += blockDim.x) {
            const int which = i / (4 * cols), j = i - which * 4 * cols;
            double acc = 0.0;
            for (int t = 0; t < tys; ++t) acc += (double)red[((size_t)t * 2 * cols) * 4 + (size_t)which * cols * 4 + j];
            const int cc = c0 * 4 + j;
            if (cc < 4 * C4) atomicAdd(sums + (size_t)(blockIdx.x % bn_slots(4 * C4)) * 8 * C4 + (size_t)which * 4 * C4 + cc, acc);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_bn_relu_pool_bwd_dx(const float* z, const float* dp, float* dz, int B, int H, int W, int C4,
                                                             float inv_m, const float* mean, const float* rstd, const float* gamma,
                                                             const float* beta, const double* sums, float* dgamma, float* dbeta) {
    extern __shared__ float coef[];  // [4][C], as k_bn_relu_bwd_dx
    const int C = 4 * C4;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float sg = (float)bn_sum(sums, C, c), sgx = (float)bn_sum(sums, C, C + c);
        const float a = gamma[c] * rstd[c], b = -a * rstd[c] * sgx * inv_m;
        coef[c] = a;
        coef[C + c] = b;
        coef[2 * C + c] = -a * sg * inv_m - b * mean[c];
        coef[3 * C + c] = beta[c] - mean[c] * a;
        if (blockIdx.x == 0) {
            dbeta[c] = sg;
            dgamma[c] = sgx;
        }
    }
    __syncthreads();
    const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
    const unsigned total = (unsigned)B * Ho * Wo * C4;  // < 2^31 (host check)
    const float4* z4 = reinterpret_cast<const float4*>(z);
    const float4* d4 = reinterpret_cast<const float4*>(dp);
    float4* o4 = reinterpret_cast<float4*>(dz);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned r = i / (unsigned)C4;
        const int c = (int)(i - r * (unsigned)C4);
        const PoolWindow w = pool_window(r, c, H, W, Ho, Wo, C4);
        float v[4][4], o[4][4];
        pool_load(z4, w, C4, v);
        const float4 d = d4[i];
        const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float a = coef[4 * c + k], b = coef[C + 4 * c + k], kk = coef[2 * C + 4 * c + k], h = coef[3 * C + 4 * c + k];
            float best = 0.f, zb = 0.f;
            int sel = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) pool_take(v[j][k], a, h, j, best, zb, sel);
            const float g = best > 0.f ? dd[k] : 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                o[j][k] = fmaf(a, sel == j ? g : 0.f, fmaf(b, v[j][k], kk));
        }
        o4[w.base] = make_float4(o[0][0], o[0][1], o[0][2], o[0][3]);
        if (w.w1) o4[w.base + C4] = make_float4(o[1][0], o[1][1], o[1][2], o[1][3]);
        if (w.h1) o4[w.base + w.down] = make_float4(o[2][0], o[2][1], o[2][2], o[2][3]);
        if (w.h1 && w.w1) o4[w.base + w.down + C4] = make_float4(o[3][0], o[3][1], o[3][2], o[3][3]);
    }
}
