// iris_frontend.hip -- HIP kernels + C ABI of the MI355X audio feature frontend.
// Written for gfx950 (CDNA4) only: 64-lane wavefronts, 160 KiB LDS per CU,
// 8 XCDs with private L2s.  See include/iris_frontend.h for the contract and
// DESIGN.md for the data layout and the roofline of each kernel.
#include "../../include/iris_frontend.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "iris_fft.h"

using namespace iris;

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                     \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                        __FILE__, __LINE__);                                              \
    } while (0)

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
struct iris_plan {
    int device;
    int n_fft, log2n, hop, n_mel, n_bins, channels, max_batch, max_len;
    float sample_rate, lower_hz, upper_hz;
    // host copies
    std::vector<float> mel;  // [F][M]
    int max_band_len, k_need;
    // device tables
    float2* d_tw;
    float2* d_post;
    float2* d_win;
    int* d_band_lo;
    float* d_wband;  // [max_band_len][M]
    float* d_mel;    // [F][M] dense (for magmel)
    int* d_band_len;
    float* d_ws;  // workspace
    size_t ws_floats;
    int tile_frames;  // frames per workgroup tile of the fused kernel
    // timing
    bool timing;
    std::vector<hipEvent_t> ev;  // pairs
    int ev_used;
};

constexpr int kChunk = 4096;        // elements per partial-reduction block
constexpr int kMaxTimedLaunches = 4096;

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int reflect_idx(int i, int len) {
    i = i < 0 ? -i : i;
    return i >= len ? 2 * (len - 1) - i : i;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Block-wide (256 threads) min/max; result valid in every thread.
__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* red /*[16]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    mn = wave_min(mn);
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) {
        red[w] = mn;
        red[8 + w] = mx;
    }
    __syncthreads();
    mn = red[0];
    mx = red[8];
    for (int i = 1; i < nw; ++i) {
        mn = fminf(mn, red[i]);
        mx = fmaxf(mx, red[8 + i]);
    }
}

// Consecutive logical workgroup ids land on the same XCD (blocks b and b+8 share
// one; bijective for any grid size).  Placement only affects speed.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, local = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + local;
}

__device__ __forceinline__ bool in_bands(const int* bands, int n, int idx) {
    bool hit = false;
    for (int i = 0; i < n; ++i) {
        const int off = bands[2 * i], size = bands[2 * i + 1];
        hit |= (idx >= off) & (idx < off + size);
    }
    return hit;
}

// ---------------------------------------------------------------------------
// frame -> spectrum pieces shared by the fused and the STFT kernels
// ---------------------------------------------------------------------------
template <int LOG2N>
__device__ __forceinline__ void load_frame(float2 (&x)[FftCfg<LOG2N>::P], const float* clip, int len, int start,
                                           int lane) {
    constexpr int N = 1 << LOG2N, P = FftCfg<LOG2N>::P;
    const bool interior = (start >= 0) && (start + N <= len) &&
                          ((reinterpret_cast<uintptr_t>(clip + start) & 7) == 0);
    if (interior) {  // wave-uniform
        const float2* p = reinterpret_cast<const float2*>(clip + start);
#pragma unroll
        for (int q = 0; q < P; ++q) x[q] = p[lane + kWave * q];
    } else {
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int n = start + 2 * (lane + kWave * q);
            x[q].x = clip[reflect_idx(n, len)];
            x[q].y = clip[reflect_idx(n + 1, len)];
        }
    }
}

// x[q] = Z[lane + 64 q] -> Xlo[q] = X[k], Xhi[q] = X[NC - k], k = lane + 64 q, q < P/2.
// Uses the wave's LDS buffer; ends with the buffer free for reuse.
template <int LOG2N, bool HI>
__device__ __forceinline__ void untangle(const float2 (&x)[FftCfg<LOG2N>::P], const float2* post, float2* lds,
                                         int lane, float2 (&xlo)[FftCfg<LOG2N>::P / 2],
                                         float2 (&xhi)[FftCfg<LOG2N>::P / 2]) {
    constexpr int NC = (1 << LOG2N) / 2, P = FftCfg<LOG2N>::P;
#pragma unroll
    for (int q = 0; q < P; ++q) lds[lds_pad(lane + kWave * q)] = x[q];
    wave_sync_lds();
#pragma unroll
    for (int q = 0; q < P / 2; ++q) {
        const int k = lane + kWave * q;
        const float2 zk = x[q];
        const float2 zp = lds[lds_pad((NC - k) & (NC - 1))];
        const float2 e = make_float2(0.5f * (zk.x + zp.x), 0.5f * (zk.y - zp.y));
        const float2 o = make_float2(0.5f * (zk.y + zp.y), -0.5f * (zk.x - zp.x));
        const float2 wo = cmul(post[q], o);
        xlo[q] = cadd(e, wo);
        if constexpr (HI) {
            const float2 d = csub(e, wo);
            xhi[q] = make_float2(d.x, -d.y);
        }
    }
    wave_sync_lds();
}

__device__ __forceinline__ float cabs_rn(float2 v) { return __builtin_amdgcn_sqrtf(fmaf(v.x, v.x, v.y * v.y)); }

// ---------------------------------------------------------------------------
// K1: fused wav -> mel magnitudes (+ per-tile min/max partials)
//   grid  = B * tiles_per_clip workgroups of 256 threads (4 waves)
//   tile  = `tile_frames` consecutive frames of one clip, all C channels
//   LDS   = 4 wave buffers (FFT exchange / magnitudes) + [M][tile_frames*C+1] out tile
// ---------------------------------------------------------------------------
struct FusedArgs {
    const float* wav;    // [B, C, L]
    float* out;          // [B, M, T, C]
    float* partial;      // [B, tiles, 2] min, max
    const float* sumsq;  // nullable [B, n_sq] partial sums of squares (normalize)
    int n_sq;
    FftTables tab;
    const int* band_lo;    // [M]
    const float* wband;    // [max_len][M]
    int max_len, k_need;
    const int* t_bands;  // nullable [B, n_tb, 2]
    int n_tb;
    const int* f_bands;  // nullable [B, n_fb, 2]
    int n_fb;
    int B, C, L, T, hop, M, tile_frames, tiles_per_clip;
};

template <int LOG2N>
__global__ __launch_bounds__(256) void k_wav_to_mel(const FusedArgs a) {
    constexpr int N = 1 << LOG2N, NC = N / 2, P = FftCfg<LOG2N>::P, NTW = FftCfg<LOG2N>::NTW;
    constexpr int F = NC + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wg / a.tiles_per_clip, tile = wg - b * a.tiles_per_clip;
    const int t0 = tile * a.tile_frames;
    const int nt = min(a.tile_frames, a.T - t0);
    const int tile_stride = a.tile_frames * a.C + 1;

    constexpr int kWaveBufBytes = lds_padded(NC) * 8;
    float2* lds = reinterpret_cast<float2*>(smem + wv * kWaveBufBytes);
    float* magbuf = reinterpret_cast<float*>(lds);
    float* tile_out = reinterpret_cast<float*>(smem + 4 * kWaveBufBytes);
    float* red = tile_out + a.M * tile_stride;  // [16]

    // per-lane constants
    float2 tw[NTW], post[P / 2], win[P];
#pragma unroll
    for (int i = 0; i < NTW; ++i) tw[i] = a.tab.tw[i * kWave + lane];
#pragma unroll
    for (int i = 0; i < P / 2; ++i) post[i] = a.tab.post[i * kWave + lane];
#pragma unroll
    for (int i = 0; i < P; ++i) win[i] = a.tab.win[i * kWave + lane];

    if (a.sumsq != nullptr) {  // normalize: fold 1 / (10 rms) into the window
        float s = 0.f;
        for (int i = lane; i < a.n_sq; i += kWave) s += a.sumsq[(size_t)b * a.n_sq + i];
        s = wave_sum(s);
        const float rms10 = sqrtf(s / ((float)a.C * (float)a.L)) * 10.0f;
        const float inv = 1.0f / rms10;
#pragma unroll
        for (int i = 0; i < P; ++i) {
            win[i].x *= inv;
            win[i].y *= inv;
        }
    }

    const int* tb = a.t_bands ? a.t_bands + (size_t)b * a.n_tb * 2 : nullptr;
    const int* fb = a.f_bands ? a.f_bands + (size_t)b * a.n_fb * 2 : nullptr;
    const bool need_hi = a.k_need > NC / 2;

    const int nwf = nt * a.C;  // wave-frames in this tile
    for (int f = wv; f < nwf; f += 4) {
        const int tl = f / a.C, c = f - tl * a.C;
        const int t = t0 + tl;
        bool masked = false;
        if (tb) masked = in_bands(tb, a.n_tb, t);
        if (masked) {  // wave-uniform
            for (int m = lane; m < a.M; m += kWave) tile_out[m * tile_stride + f] = 0.f;
            continue;
        }
        const float* clip = a.wav + ((size_t)b * a.C + c) * a.L;
        float2 x[P];
        load_frame<LOG2N>(x, clip, a.L, t * a.hop - N / 2, lane);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            x[q].x *= win[q].x;
            x[q].y *= win[q].y;
        }
        fft_frame<LOG2N>(x, tw, lds, lane);
        float2 xlo[P / 2], xhi[P / 2];
        if (need_hi) {
            untangle<LOG2N, true>(x, post, lds, lane, xlo, xhi);
#pragma unroll
            for (int q = 0; q < P / 2; ++q) {
                const int k = lane + kWave * q;
                magbuf[k] = cabs_rn(xlo[q]);
                magbuf[NC - k] = cabs_rn(xhi[q]);
            }
            if (lane == 0) magbuf[NC / 2] = cabs_rn(x[P / 2]);
        } else {
            untangle<LOG2N, false>(x, post, lds, lane, xlo, xhi);
#pragma unroll
            for (int q = 0; q < P / 2; ++q) magbuf[lane + kWave * q] = cabs_rn(xlo[q]);
        }
        wave_sync_lds();
        if (fb) {
            for (int i = 0; i < a.n_fb; ++i) {
                const int off = fb[2 * i], end = min(off + fb[2 * i + 1], F);
                for (int k = off + lane; k < end; k += kWave) magbuf[k] = 0.f;
            }
            wave_sync_lds();
        }
        for (int m = lane; m < a.M; m += kWave) {
            const int lo = a.band_lo[m];
            float acc = 0.f;
            for (int i = 0; i < a.max_len; ++i) {
                const float w = a.wband[i * a.M + m];
                acc = fmaf(w, magbuf[min(lo + i, F - 1)], acc);
            }
            tile_out[m * tile_stride + f] = acc;
        }
        wave_sync_lds();
    }
    __syncthreads();

    // write the tile: for each m a contiguous run of nt*C floats
    const int run = nt * a.C;
    float mn = INFINITY, mx = -INFINITY;
    for (int idx = threadIdx.x; idx < a.M * run; idx += blockDim.x) {
        const int m = idx / run, r = idx - m * run;
        const float v = tile_out[m * tile_stride + r];
        a.out[(((size_t)b * a.M + m) * a.T + t0) * a.C + r] = v;
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
    }
    block_minmax(mn, mx, red);
    if (threadIdx.x == 0) {
        a.partial[((size_t)b * a.tiles_per_clip + tile) * 2 + 0] = mn;
        a.partial[((size_t)b * a.tiles_per_clip + tile) * 2 + 1] = mx;
    }
}

// ---------------------------------------------------------------------------
// K2: STFT only, reference layout [B, F, T, 2C]
// ---------------------------------------------------------------------------
struct StftArgs {
    const float* wav;
    float* spec;
    FftTables tab;
    int B, C, L, T, hop, tile_frames, tiles_per_clip;
};

template <int LOG2N>
__global__ __launch_bounds__(256) void k_stft(const StftArgs a) {
    constexpr int N = 1 << LOG2N, NC = N / 2, P = FftCfg<LOG2N>::P, NTW = FftCfg<LOG2N>::NTW;
    constexpr int F = NC + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wg / a.tiles_per_clip, tile = wg - b * a.tiles_per_clip;
    const int t0 = tile * a.tile_frames;
    const int nt = min(a.tile_frames, a.T - t0);
    const int C2 = 2 * a.C;
    const int row = a.tile_frames * C2 + 1;  // odd stride: conflict-free column writes

    constexpr int kWaveBufBytes = lds_padded(NC) * 8;
    float2* lds = reinterpret_cast<float2*>(smem + wv * kWaveBufBytes);
    float* tile_out = reinterpret_cast<float*>(smem + 4 * kWaveBufBytes);  // [F][row]

    float2 tw[NTW], post[P / 2], win[P];
#pragma unroll
    for (int i = 0; i < NTW; ++i) tw[i] = a.tab.tw[i * kWave + lane];
#pragma unroll
    for (int i = 0; i < P / 2; ++i) post[i] = a.tab.post[i * kWave + lane];
#pragma unroll
    for (int i = 0; i < P; ++i) win[i] = a.tab.win[i * kWave + lane];

    const int nwf = nt * a.C;
    for (int f = wv; f < nwf; f += 4) {
        const int tl = f / a.C, c = f - tl * a.C;
        const float* clip = a.wav + ((size_t)b * a.C + c) * a.L;
        float2 x[P];
        load_frame<LOG2N>(x, clip, a.L, (t0 + tl) * a.hop - N / 2, lane);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            x[q].x *= win[q].x;
            x[q].y *= win[q].y;
        }
        fft_frame<LOG2N>(x, tw, lds, lane);
        float2 xlo[P / 2], xhi[P / 2];
        untangle<LOG2N, true>(x, post, lds, lane, xlo, xhi);
        const int col = tl * C2 + c;
#pragma unroll
        for (int q = 0; q < P / 2; ++q) {
            const int k = lane + kWave * q;
            tile_out[k * row + col] = xlo[q].x;
            tile_out[k * row + col + a.C] = xlo[q].y;
            tile_out[(NC - k) * row + col] = xhi[q].x;  // k = 0 -> Nyquist bin NC
            tile_out[(NC - k) * row + col + a.C] = xhi[q].y;
        }
        if (lane == 0) {  // X[NC/2] = conj(Z[NC/2])
            tile_out[(NC / 2) * row + col] = x[P / 2].x;
            tile_out[(NC / 2) * row + col + a.C] = -x[P / 2].y;
        }
    }
    __syncthreads();
    const int run = nt * C2;
    for (int idx = threadIdx.x; idx < F * run; idx += blockDim.x) {
        const int k = idx / run, r = idx - k * run;
        a.spec[(((size_t)b * F + k) * a.T + t0) * C2 + r] = tile_out[k * row + r];
    }
}

// ---------------------------------------------------------------------------
// K3: spectrum -> mel (complex_to_magphase + magphase_to_mel fused)
//   block = 256 threads: 64 consecutive (t, c) columns x 4 waves over mel bands
// ---------------------------------------------------------------------------
struct MagmelArgs {
    const float* spec;  // [B, F, T, 2C]
    float* mel;         // [B, M, T, C]
    const float* w;     // dense [F][M]
    const int* band_lo;
    const int* band_len;
    const int* t_bands;
    int n_tb;
    const int* f_bands;
    int n_fb;
    int B, C, F, T, M, is_magphase;
};

__global__ __launch_bounds__(256) void k_magmel(const MagmelArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int tc = blockIdx.x * 64 + lane;
    const int TC = a.T * a.C;
    const bool valid = tc < TC;
    const int t = valid ? tc / a.C : 0, c = valid ? tc - t * a.C : 0;
    const int C2 = 2 * a.C;
    const int* tb = a.t_bands ? a.t_bands + (size_t)b * a.n_tb * 2 : nullptr;
    const int* fb = a.f_bands ? a.f_bands + (size_t)b * a.n_fb * 2 : nullptr;
    const bool tmask = tb ? in_bands(tb, a.n_tb, t) : false;
    const float* sp = a.spec + (size_t)b * a.F * a.T * C2 + (size_t)t * C2 + c;
    for (int m = wv; m < a.M; m += 4) {
        const int lo = a.band_lo[m], len = a.band_len[m];
        float acc = 0.f;
        for (int i = 0; i < len; ++i) {
            const int f = lo + i;
            if (fb && in_bands(fb, a.n_fb, f)) continue;  // uniform
            const float w = a.w[f * a.M + m];
            float mag = 0.f;
            if (valid) {
                const float re = sp[(size_t)f * a.T * C2];
                if (a.is_magphase) {
                    mag = re;
                } else {
                    const float im = sp[(size_t)f * a.T * C2 + a.C];
                    mag = __builtin_amdgcn_sqrtf(fmaf(re, re, im * im));
                }
            }
            acc = fmaf(w, mag, acc);
        }
        if (valid) a.mel[(((size_t)b * a.M + m) * a.T + t) * a.C + c] = tmask ? 0.f : acc;
    }
}

// ---------------------------------------------------------------------------
// K4/K5: min-max (+ log): partial reduce, then apply
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax_partial(const float* x, float* partial, size_t row_len,
                                                        int n_part) {
    __shared__ float red[16];
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

__global__ __launch_bounds__(256) void k_minmax_log_apply(float* x, const float* partial, int n_part,
                                                          size_t row_len, int do_minmax, int do_log,
                                                          float eps_div, float eps_log) {
    __shared__ float red[16];
    const int row = blockIdx.y;
    float mn = 0.f, den = 1.f;
    if (do_minmax) {
        float lo = INFINITY, hi = -INFINITY;
        for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
            lo = fminf(lo, partial[((size_t)row * n_part + i) * 2 + 0]);
            hi = fmaxf(hi, partial[((size_t)row * n_part + i) * 2 + 1]);
        }
        block_minmax(lo, hi, red);
        mn = lo;
        den = fmaxf(hi - lo, eps_div);
    }
    float* p = x + (size_t)row * row_len;
    const size_t beg = (size_t)blockIdx.x * kChunk, end = min(beg + (size_t)kChunk, row_len);
    const bool vec = ((row_len & 3) == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    if (vec) {
        for (size_t i = beg + 4 * (size_t)threadIdx.x; i < end; i += 4 * (size_t)blockDim.x) {
            float4 v = *reinterpret_cast<float4*>(p + i);
            float* e = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = e[j];
                if (do_minmax) y = (y - mn) / den;
                if (do_log) y = logf(y + eps_log);
                e[j] = y;
            }
            *reinterpret_cast<float4*>(p + i) = v;
        }
    } else {
        for (size_t i = beg + threadIdx.x; i < end; i += blockDim.x) {
            float y = p[i];
            if (do_minmax) y = (y - mn) / den;
            if (do_log) y = logf(y + eps_log);
            p[i] = y;
        }
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
// host: mel matrix (fp32 recipe of tf.signal.linear_to_mel_weight_matrix)
// ---------------------------------------------------------------------------
// All fp32, one rounding per operation (no FMA contraction); the logarithm is the
// correctly rounded fp32 one (evaluated in double, rounded once).
#pragma clang fp contract(off)
static inline float hz_to_mel(float hz) {
    const float arg = 1.0f + hz / 700.0f;
    const float ln = (float)log((double)arg);
    return 1127.0f * ln;
}

static void linspace_f32(float start, float stop, int num, std::vector<float>& out) {
    out.resize(num);
    if (num == 1) {
        out[0] = start;
        return;
    }
    const float step = (stop - start) / (float)(num - 1);
    for (int i = 0; i < num; ++i) out[i] = start + step * (float)i;
    out[num - 1] = stop;
}

extern "C" int iris_mel_weight_matrix(int n_mel, int n_bins, float sample_rate, float lower_hz, float upper_hz,
                                      float* out) {
    if (!out) return fail(IRIS_E_INVALID, "iris_mel_weight_matrix: out is NULL");
    if (n_mel <= 0) return fail(IRIS_E_INVALID, "num_mel_bins must be positive");
    if (n_bins < 2) return fail(IRIS_E_INVALID, "num_spectrogram_bins must be >= 2");
    if (!(sample_rate > 0.f)) return fail(IRIS_E_INVALID, "sample_rate must be positive");
    if (lower_hz < 0.f) return fail(IRIS_E_INVALID, "lower_edge_hertz must be non-negative");
    if (!(lower_hz < upper_hz)) return fail(IRIS_E_INVALID, "lower_edge_hertz must be < upper_edge_hertz");
    if (upper_hz > sample_rate / 2.f) return fail(IRIS_E_INVALID, "upper_edge_hertz must not exceed Nyquist");
    std::vector<float> lin, edges;
    linspace_f32(0.f, sample_rate / 2.0f, n_bins, lin);
    linspace_f32(hz_to_mel(lower_hz), hz_to_mel(upper_hz), n_mel + 2, edges);
    for (int m = 0; m < n_mel; ++m) out[m] = 0.f;  // DC bin
    for (int f = 1; f < n_bins; ++f) {
        const float mel = hz_to_mel(lin[f]);
        for (int m = 0; m < n_mel; ++m) {
            const float lo = edges[m], ctr = edges[m + 1], hi = edges[m + 2];
            const float up = (mel - lo) / (ctr - lo);
            const float dn = (hi - mel) / (hi - ctr);
            out[(size_t)f * n_mel + m] = fmaxf(0.f, fminf(up, dn));
        }
    }
    return IRIS_OK;
}

// ---------------------------------------------------------------------------
// host: plan
// ---------------------------------------------------------------------------
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

static int fft_ntw(int log2n) {
    switch (log2n) {
        case 11: return FftCfg<11>::NTW;
        case 10: return FftCfg<10>::NTW;
        case 9: return FftCfg<9>::NTW;
        default: return FftCfg<8>::NTW;
    }
}
static int fft_p(int log2n) { return (1 << log2n) / 2 / 64; }

static void build_tables(int log2n, std::vector<float2>& tw, std::vector<float2>& post, std::vector<float2>& win) {
    const int N = 1 << log2n, NC = N / 2, P = fft_p(log2n);
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<int> radices;
    if (log2n == 11) radices = {16, 16, 4};
    else if (log2n == 10) radices = {8, 8, 8};
    else if (log2n == 9) radices = {4, 4, 4, 4};
    else radices = {2, 2, 2, 2, 2, 2, 2};
    tw.clear();
    int ns = 1;
    for (size_t s = 0; s < radices.size(); ++s) {
        const int R = radices[s], U = P / R;
        if (s > 0) {
            for (int u = 0; u < U; ++u)
                for (int t = 1; t < R; ++t)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int bfly = lane + 64 * u;
                        const double ang = -two_pi * (double)((bfly % ns) * t) / (double)(ns * R);
                        tw.push_back(make_float2((float)cos(ang), (float)sin(ang)));
                    }
        }
        ns *= R;
    }
    post.clear();
    for (int q = 0; q < P / 2; ++q)
        for (int lane = 0; lane < 64; ++lane) {
            const double ang = -two_pi * (double)(lane + 64 * q) / (double)N;
            post.push_back(make_float2((float)cos(ang), (float)sin(ang)));
        }
    win.clear();
    for (int q = 0; q < P; ++q)
        for (int lane = 0; lane < 64; ++lane) {
            const int n = 2 * (lane + 64 * q);
            const double w0 = 0.5 - 0.5 * cos(two_pi * (double)n / (double)N);
            const double w1 = 0.5 - 0.5 * cos(two_pi * (double)(n + 1) / (double)N);
            win.push_back(make_float2((float)w0, (float)w1));
        }
    (void)NC;
}

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
    HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(src.size(), 1) * sizeof(T)));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return IRIS_OK;
}

static size_t fused_lds_bytes(const iris_plan* p, int tile_frames) {
    const int NC = p->n_fft / 2;
    return 4 * (size_t)lds_padded(NC) * 8 + ((size_t)p->n_mel * (tile_frames * p->channels + 1) + 16) * 4;
}

// Dynamic LDS above the 64 KiB default must be opted into once per kernel.
template <int LOG2N>
static hipError_t allow_big_lds() {
    constexpr int kMaxLds = 160 * 1024;
    hipError_t e = hipFuncSetAttribute((const void*)k_wav_to_mel<LOG2N>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)k_stft<LOG2N>, hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
}

extern "C" int iris_abi_version(void) { return IRIS_ABI_VERSION; }
extern "C" const char* iris_last_error(void) { return g_err; }

extern "C" int iris_plan_create(iris_plan** out, int device, int n_fft, int hop, int n_mel, int n_bins,
                                float sample_rate, float lower_hz, float upper_hz, int channels, int max_batch,
                                int max_len, const float* mel_host) {
    if (!out) return fail(IRIS_E_INVALID, "iris_plan_create: out is NULL");
    *out = nullptr;
    const int log2n = ilog2_exact(n_fft);
    if (log2n < 8 || log2n > 11)
        return fail(IRIS_E_UNSUPPORTED, "n_fft=%d: must be a power of two in [256, 2048]", n_fft);
    if (hop <= 0) return fail(IRIS_E_INVALID, "hop=%d must be positive", hop);
    if (n_mel <= 0) return fail(IRIS_E_INVALID, "n_mel=%d must be positive", n_mel);
    if (n_bins != n_fft / 2 + 1)
        return fail(IRIS_E_INVALID, "n_bins=%d must equal n_fft/2+1=%d", n_bins, n_fft / 2 + 1);
    if (channels <= 0 || max_batch <= 0) return fail(IRIS_E_INVALID, "channels and max_batch must be positive");
    if (max_len <= n_fft / 2)
        return fail(IRIS_E_INVALID, "max_len=%d must exceed n_fft/2 (reflect padding)", max_len);

    iris_plan* p = new (std::nothrow) iris_plan();
    if (!p) return fail(IRIS_E_NOMEM, "out of host memory");
    p->device = device;
    p->n_fft = n_fft;
    p->log2n = log2n;
    p->hop = hop;
    p->n_mel = n_mel;
    p->n_bins = n_bins;
    p->channels = channels;
    p->max_batch = max_batch;
    p->max_len = max_len;
    p->sample_rate = sample_rate;
    p->lower_hz = lower_hz;
    p->upper_hz = upper_hz;
    p->d_tw = p->d_post = p->d_win = nullptr;
    p->d_band_lo = p->d_band_len = nullptr;
    p->d_wband = p->d_mel = p->d_ws = nullptr;
    p->timing = false;
    p->ev_used = 0;

    p->mel.resize((size_t)n_bins * n_mel);
    if (mel_host) {
        memcpy(p->mel.data(), mel_host, p->mel.size() * sizeof(float));
    } else {
        int rc = iris_mel_weight_matrix(n_mel, n_bins, sample_rate, lower_hz, upper_hz, p->mel.data());
        if (rc != IRIS_OK) {
            delete p;
            return rc;
        }
    }
    // band structure: per mel column the contiguous bin range holding its non-zeros
    std::vector<int> lo(n_mel, 0), len(n_mel, 0);
    p->max_band_len = 0;
    p->k_need = 0;
    for (int m = 0; m < n_mel; ++m) {
        int first = -1, last = -1;
        for (int f = 0; f < n_bins; ++f)
            if (p->mel[(size_t)f * n_mel + m] != 0.f) {
                if (first < 0) first = f;
                last = f;
            }
        if (first >= 0) {
            lo[m] = first;
            len[m] = last - first + 1;
        }
        p->max_band_len = std::max(p->max_band_len, len[m]);
        p->k_need = std::max(p->k_need, lo[m] + len[m]);
    }
    std::vector<float> wband((size_t)std::max(p->max_band_len, 1) * n_mel, 0.f);
    for (int m = 0; m < n_mel; ++m)
        for (int i = 0; i < len[m]; ++i) wband[(size_t)i * n_mel + m] = p->mel[(size_t)(lo[m] + i) * n_mel + m];

    DeviceGuard guard(device);
    if (!guard.ok) {
        delete p;
        return fail(IRIS_E_INVALID, "cannot select HIP device %d", device);
    }
    std::vector<float2> tw, post, win;
    build_tables(log2n, tw, post, win);
    int rc;
    if ((rc = upload(&p->d_tw, tw)) || (rc = upload(&p->d_post, post)) || (rc = upload(&p->d_win, win)) ||
        (rc = upload(&p->d_band_lo, lo)) || (rc = upload(&p->d_band_len, len)) ||
        (rc = upload(&p->d_wband, wband)) || (rc = upload(&p->d_mel, p->mel))) {
        iris_plan_destroy(p);
        return rc;
    }

    {
        hipError_t e;
        switch (log2n) {
            case 11: e = allow_big_lds<11>(); break;
            case 10: e = allow_big_lds<10>(); break;
            case 9: e = allow_big_lds<9>(); break;
            default: e = allow_big_lds<8>(); break;
        }
        if (e != hipSuccess) {
            iris_plan_destroy(p);
            return fail((int)e, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
        }
    }
    // tile size of the fused kernel: 16 frames unless LDS says otherwise
    p->tile_frames = 16;
    if (const char* e = getenv("IRIS_TILE_FRAMES")) p->tile_frames = std::max(1, atoi(e));
    while (p->tile_frames > 1 && fused_lds_bytes(p, p->tile_frames) > 64 * 1024) p->tile_frames /= 2;
    if (fused_lds_bytes(p, p->tile_frames) > 160 * 1024) {
        iris_plan_destroy(p);
        return fail(IRIS_E_UNSUPPORTED, "n_mel=%d x channels=%d does not fit the LDS out tile", n_mel, channels);
    }

    // workspace of the fused path: [B, tiles, 2] min/max partials (worst case one
    // frame per tile) + [B, chunks] sums of squares for IRIS_F_NORMALIZE
    const int t_max = 1 + max_len / hop;
    const size_t wav_row = (size_t)channels * max_len;
    p->ws_floats = 2 * (size_t)max_batch * t_max + (size_t)max_batch * ((wav_row + kChunk - 1) / kChunk) + 64;
    hipError_t e = hipMalloc((void**)&p->d_ws, p->ws_floats * sizeof(float));
    if (e != hipSuccess) {
        iris_plan_destroy(p);
        return fail((int)e, "hipMalloc(workspace %zu floats) failed: %s", p->ws_floats, hipGetErrorString(e));
    }
    *out = p;
    return IRIS_OK;
}

extern "C" int iris_plan_destroy(iris_plan* p) {
    if (!p) return IRIS_OK;
    DeviceGuard guard(p->device);
    for (hipEvent_t ev : p->ev) (void)hipEventDestroy(ev);
    (void)hipFree(p->d_tw);
    (void)hipFree(p->d_post);
    (void)hipFree(p->d_win);
    (void)hipFree(p->d_band_lo);
    (void)hipFree(p->d_band_len);
    (void)hipFree(p->d_wband);
    (void)hipFree(p->d_mel);
    (void)hipFree(p->d_ws);
    delete p;
    return IRIS_OK;
}

extern "C" int iris_plan_get_mel(const iris_plan* p, float* out) {
    if (!p || !out) return fail(IRIS_E_INVALID, "iris_plan_get_mel: NULL argument");
    memcpy(out, p->mel.data(), p->mel.size() * sizeof(float));
    return IRIS_OK;
}

extern "C" int iris_plan_num_frames(const iris_plan* p, int len) {
    if (!p || len < 0) return fail(IRIS_E_INVALID, "iris_plan_num_frames: bad argument");
    return 1 + len / p->hop;
}

static int check_wav_args(const iris_plan* p, const void* a, const void* b, int batch, int len, const char* who) {
    if (!p || !a || !b) return fail(IRIS_E_INVALID, "%s: NULL argument", who);
    if (batch <= 0 || len <= 0) return fail(IRIS_E_INVALID, "%s: batch=%d len=%d must be positive", who, batch, len);
    if (batch > p->max_batch || len > p->max_len)
        return fail(IRIS_E_CAPACITY, "%s: batch=%d len=%d exceed plan capacity (%d, %d)", who, batch, len,
                    p->max_batch, p->max_len);
    if (len <= p->n_fft / 2)
        return fail(IRIS_E_INVALID, "%s: len=%d must exceed n_fft/2=%d (reflect padding)", who, len, p->n_fft / 2);
    return IRIS_OK;
}

static int check_bands(const int32_t* bands, int n, const char* who) {
    if (n < 0 || (n > 0 && !bands)) return fail(IRIS_E_INVALID, "%s: bands pointer/count mismatch", who);
    return IRIS_OK;
}

// ---------------------------------------------------------------------------
// host: ops
// ---------------------------------------------------------------------------
static size_t n_chunks_of(size_t row_len) { return (row_len + kChunk - 1) / kChunk; }

extern "C" size_t iris_normalize_workspace(int n_rows, size_t row_len) {
    return n_rows > 0 ? (size_t)n_rows * n_chunks_of(row_len) : 0;
}

extern "C" int iris_normalize(const float* wav, float* out, int n_rows, size_t row_len, float* workspace,
                              size_t workspace_floats, void* stream) {
    if (!wav || !out || !workspace) return fail(IRIS_E_INVALID, "iris_normalize: NULL argument");
    if (n_rows <= 0 || row_len == 0) return fail(IRIS_E_INVALID, "iris_normalize: empty tensor");
    if (n_rows > 65535) return fail(IRIS_E_UNSUPPORTED, "iris_normalize: n_rows %d > 65535", n_rows);
    const size_t n_part = n_chunks_of(row_len);
    if (workspace_floats < iris_normalize_workspace(n_rows, row_len))
        return fail(IRIS_E_CAPACITY, "iris_normalize: workspace %zu floats < %zu", workspace_floats,
                    iris_normalize_workspace(n_rows, row_len));
    hipStream_t s = (hipStream_t)stream;
    k_sumsq_partial<<<dim3((unsigned)n_part, n_rows), 256, 0, s>>>(wav, workspace, row_len, (int)n_part);
    k_normalize_apply<<<dim3((unsigned)n_part, n_rows), 256, 0, s>>>(wav, out, workspace, (int)n_part, row_len);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

template <int LOG2N>
static hipError_t launch_stft(const StftArgs& a, int grid, size_t lds, hipStream_t s) {
    k_stft<LOG2N><<<grid, 256, lds, s>>>(a);
    return hipGetLastError();
}

extern "C" int iris_stft(iris_plan* p, const float* wav, float* spec, int batch, int len, void* stream) {
    int rc = check_wav_args(p, wav, spec, batch, len, "iris_stft");
    if (rc) return rc;
    DeviceGuard guard(p->device);
    StftArgs a;
    a.wav = wav;
    a.spec = spec;
    a.tab = FftTables{p->d_tw, p->d_post, p->d_win};
    a.B = batch;
    a.C = p->channels;
    a.L = len;
    a.T = 1 + len / p->hop;
    a.hop = p->hop;
    const int NC = p->n_fft / 2, F = NC + 1;
    int tf = 16;
    auto lds_of = [&](int t) { return 4 * (size_t)lds_padded(NC) * 8 + (size_t)F * (t * 2 * p->channels + 1) * 4; };
    while (tf > 1 && lds_of(tf) > 64 * 1024) tf /= 2;
    if (lds_of(tf) > 160 * 1024) return fail(IRIS_E_UNSUPPORTED, "iris_stft: channels=%d too large", p->channels);
    a.tile_frames = tf;
    a.tiles_per_clip = (a.T + tf - 1) / tf;
    const int grid = batch * a.tiles_per_clip;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e;
    switch (p->log2n) {
        case 11: e = launch_stft<11>(a, grid, lds_of(tf), s); break;
        case 10: e = launch_stft<10>(a, grid, lds_of(tf), s); break;
        case 9: e = launch_stft<9>(a, grid, lds_of(tf), s); break;
        default: e = launch_stft<8>(a, grid, lds_of(tf), s); break;
    }
    HIP_TRY(e);
    return IRIS_OK;
}

static int grid_for(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 2048 * 4); }

extern "C" int iris_complex_to_magphase(const float* in, float* out, size_t n_outer, int channels, void* stream) {
    if (!in || !out || channels <= 0) return fail(IRIS_E_INVALID, "iris_complex_to_magphase: bad argument");
    if (n_outer == 0) return IRIS_OK;
    k_complex_to_magphase<<<grid_for(n_outer * channels), 256, 0, (hipStream_t)stream>>>(in, out, n_outer, channels);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_magphase_to_complex(const float* in, float* out, size_t n_outer, int channels, void* stream) {
    if (!in || !out || channels <= 0) return fail(IRIS_E_INVALID, "iris_magphase_to_complex: bad argument");
    if (n_outer == 0) return IRIS_OK;
    k_magphase_to_complex<<<grid_for(n_outer * channels), 256, 0, (hipStream_t)stream>>>(in, out, n_outer, channels);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_magmel(iris_plan* p, const float* spec, float* mel, int batch, int n_frames, int is_magphase,
                           const int32_t* t_bands, int n_tb, const int32_t* f_bands, int n_fb, void* stream) {
    if (!p || !spec || !mel) return fail(IRIS_E_INVALID, "iris_magmel: NULL argument");
    if (batch <= 0 || n_frames <= 0) return fail(IRIS_E_INVALID, "iris_magmel: batch=%d n_frames=%d", batch, n_frames);
    if (batch > 65535) return fail(IRIS_E_UNSUPPORTED, "iris_magmel: batch %d > 65535", batch);
    int rc;
    if ((rc = check_bands(t_bands, n_tb, "iris_magmel")) || (rc = check_bands(f_bands, n_fb, "iris_magmel"))) return rc;
    DeviceGuard guard(p->device);
    MagmelArgs a;
    a.spec = spec;
    a.mel = mel;
    a.w = p->d_mel;
    a.band_lo = p->d_band_lo;
    a.band_len = p->d_band_len;
    a.t_bands = n_tb ? t_bands : nullptr;
    a.n_tb = n_tb;
    a.f_bands = n_fb ? f_bands : nullptr;
    a.n_fb = n_fb;
    a.B = batch;
    a.C = p->channels;
    a.F = p->n_bins;
    a.T = n_frames;
    a.M = p->n_mel;
    a.is_magphase = is_magphase;
    const int tc = n_frames * p->channels;
    k_magmel<<<dim3((tc + 63) / 64, batch), 256, 0, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" size_t iris_minmax_log_workspace(int n_rows, size_t row_len) {
    return n_rows > 0 ? 2 * (size_t)n_rows * n_chunks_of(row_len) : 0;
}

extern "C" int iris_minmax_log(float* x, int n_rows, size_t row_len, int do_minmax, int do_log, float eps_div,
                               float eps_log, float* workspace, size_t workspace_floats, void* stream) {
    if (!x) return fail(IRIS_E_INVALID, "iris_minmax_log: x is NULL");
    if (n_rows <= 0 || row_len == 0) return fail(IRIS_E_INVALID, "iris_minmax_log: empty tensor");
    if (n_rows > 65535) return fail(IRIS_E_UNSUPPORTED, "iris_minmax_log: n_rows %d > 65535", n_rows);
    hipStream_t s = (hipStream_t)stream;
    const size_t n_part = n_chunks_of(row_len);
    if (do_minmax) {
        if (!workspace || workspace_floats < iris_minmax_log_workspace(n_rows, row_len))
            return fail(IRIS_E_CAPACITY, "iris_minmax_log: workspace %zu floats < %zu", workspace_floats,
                        iris_minmax_log_workspace(n_rows, row_len));
        k_minmax_partial<<<dim3((unsigned)n_part, n_rows), 256, 0, s>>>(x, workspace, row_len, (int)n_part);
    }
    k_minmax_log_apply<<<dim3((unsigned)n_part, n_rows), 256, 0, s>>>(x, workspace, (int)n_part, row_len, do_minmax,
                                                                     do_log, eps_div, eps_log);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

template <int LOG2N>
static hipError_t launch_fused(const FusedArgs& a, int grid, size_t lds, hipStream_t s) {
    k_wav_to_mel<LOG2N><<<grid, 256, lds, s>>>(a);
    return hipGetLastError();
}

extern "C" int iris_wav_to_logmel(iris_plan* p, const float* wav, float* out, int batch, int len, int flags,
                                  const int32_t* t_bands, int n_tb, const int32_t* f_bands, int n_fb,
                                  void* stream) {
    int rc = check_wav_args(p, wav, out, batch, len, "iris_wav_to_logmel");
    if (rc) return rc;
    if ((rc = check_bands(t_bands, n_tb, "iris_wav_to_logmel")) ||
        (rc = check_bands(f_bands, n_fb, "iris_wav_to_logmel")))
        return rc;
    DeviceGuard guard(p->device);
    hipStream_t s = (hipStream_t)stream;
    FusedArgs a;
    a.wav = wav;
    a.out = out;
    a.tab = FftTables{p->d_tw, p->d_post, p->d_win};
    a.band_lo = p->d_band_lo;
    a.wband = p->d_wband;
    a.max_len = p->max_band_len;
    a.k_need = p->k_need;
    a.t_bands = n_tb ? t_bands : nullptr;
    a.n_tb = n_tb;
    a.f_bands = n_fb ? f_bands : nullptr;
    a.n_fb = n_fb;
    a.B = batch;
    a.C = p->channels;
    a.L = len;
    a.T = 1 + len / p->hop;
    a.hop = p->hop;
    a.M = p->n_mel;
    a.tile_frames = p->tile_frames;
    a.tiles_per_clip = (a.T + a.tile_frames - 1) / a.tile_frames;
    const size_t n_partial = 2 * (size_t)batch * a.tiles_per_clip;
    a.partial = p->d_ws;
    a.sumsq = nullptr;
    a.n_sq = 0;
    if (flags & IRIS_F_NORMALIZE) {
        const size_t row = (size_t)p->channels * len;
        a.n_sq = (int)((row + kChunk - 1) / kChunk);
        float* sq = p->d_ws + n_partial;
        if (n_partial + (size_t)batch * a.n_sq > p->ws_floats)
            return fail(IRIS_E_CAPACITY, "iris_wav_to_logmel: workspace too small");
        k_sumsq_partial<<<dim3(a.n_sq, batch), 256, 0, s>>>(wav, sq, row, a.n_sq);
        a.sumsq = sq;
    }
    if (n_partial > p->ws_floats) return fail(IRIS_E_CAPACITY, "iris_wav_to_logmel: workspace too small");
    const int grid = batch * a.tiles_per_clip;
    const size_t lds = fused_lds_bytes(p, a.tile_frames);

    const bool timed = p->timing && p->ev_used < kMaxTimedLaunches;
    if (timed) {
        while ((int)p->ev.size() < 2 * (p->ev_used + 1)) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreate(&ev));
            p->ev.push_back(ev);
        }
        HIP_TRY(hipEventRecord(p->ev[2 * p->ev_used], s));
    }
    hipError_t e;
    switch (p->log2n) {
        case 11: e = launch_fused<11>(a, grid, lds, s); break;
        case 10: e = launch_fused<10>(a, grid, lds, s); break;
        case 9: e = launch_fused<9>(a, grid, lds, s); break;
        default: e = launch_fused<8>(a, grid, lds, s); break;
    }
    HIP_TRY(e);
    if (timed) {
        HIP_TRY(hipEventRecord(p->ev[2 * p->ev_used + 1], s));
        p->ev_used++;
    }
    const int do_minmax = (flags & IRIS_F_MINMAX) ? 1 : 0, do_log = (flags & IRIS_F_LOG) ? 1 : 0;
    if (do_minmax || do_log) {
        const size_t row_len = (size_t)p->n_mel * a.T * p->channels;
        const unsigned n_chunks = (unsigned)((row_len + kChunk - 1) / kChunk);
        k_minmax_log_apply<<<dim3(n_chunks, batch), 256, 0, s>>>(out, p->d_ws, a.tiles_per_clip, row_len, do_minmax,
                                                               do_log, 1e-8f, 1e-8f);
        HIP_TRY(hipGetLastError());
    }
    return IRIS_OK;
}

extern "C" int iris_mask_apply(void* x, size_t n_outer, size_t axis_len, size_t n_inner, int elem_size,
                               const int32_t* bands, int n_bands, size_t outer_per_group, void* stream) {
    if (!x) return fail(IRIS_E_INVALID, "iris_mask_apply: x is NULL");
    if (elem_size != 4 && elem_size != 8) return fail(IRIS_E_UNSUPPORTED, "iris_mask_apply: elem_size %d", elem_size);
    int rc = check_bands(bands, n_bands, "iris_mask_apply");
    if (rc) return rc;
    if (outer_per_group == 0) return fail(IRIS_E_INVALID, "iris_mask_apply: outer_per_group must be > 0");
    const size_t total = n_outer * axis_len * n_inner;
    if (total == 0 || n_bands == 0) return IRIS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (elem_size == 4)
        k_mask_apply<uint32_t><<<grid_for(total), 256, 0, s>>>((uint32_t*)x, n_outer, axis_len, n_inner, bands,
                                                              n_bands, outer_per_group);
    else
        k_mask_apply<uint64_t><<<grid_for(total), 256, 0, s>>>((uint64_t*)x, n_outer, axis_len, n_inner, bands,
                                                              n_bands, outer_per_group);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_timing_enable(iris_plan* p, int enable) {
    if (!p) return fail(IRIS_E_INVALID, "iris_timing_enable: NULL plan");
    p->timing = enable != 0;
    p->ev_used = 0;
    return IRIS_OK;
}

extern "C" int iris_timing_read(iris_plan* p, int* n_launches, float* mean_ms) {
    if (!p || !n_launches || !mean_ms) return fail(IRIS_E_INVALID, "iris_timing_read: NULL argument");
    DeviceGuard guard(p->device);
    double total = 0.0;
    for (int i = 0; i < p->ev_used; ++i) {
        HIP_TRY(hipEventSynchronize(p->ev[2 * i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, p->ev[2 * i], p->ev[2 * i + 1]));
        total += ms;
    }
    *n_launches = p->ev_used;
    *mean_ms = p->ev_used ? (float)(total / p->ev_used) : 0.f;
    p->ev_used = 0;
    return IRIS_OK;
}
