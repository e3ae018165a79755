// iris_fft.h -- wave-per-frame real FFT core for gfx950 (CDNA4).
//
// One 64-lane wavefront transforms one n_fft-point real frame:
//   * the frame is packed as NC = n_fft/2 complex points z[n] = x[2n] + i x[2n+1];
//   * every lane keeps P = NC/64 points in registers; lane j holds z[j + 64 q];
//   * the complex FFT is a Stockham autosort sequence of radix-R stages
//     (R <= P, done entirely in registers); between stages the wave exchanges
//     its points through a private, padded LDS region with 8-byte accesses;
//     after the last stage lane j holds Z[j + 64 q] again (natural order);
//   * the real spectrum follows from X[k] = E + w^k O, X[NC-k] = conj(E - w^k O)
//     with E = (Z[k] + conj Z[NC-k]) / 2, O = (Z[k] - conj Z[NC-k]) / 2i.
// No barrier is needed: a wave's DS instructions execute in issue order.
#pragma once
#include <hip/hip_runtime.h>

namespace iris {

constexpr int kWave = 64;

__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// multiply by -i
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// cos/sin(2 pi k / 16), k = 0..7
__device__ constexpr float kC16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                                      0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
__device__ constexpr float kS16[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                                      1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};

// In-place forward DFT of R points, natural-order output.
template <int R>
__device__ __forceinline__ void dft(float2 (&v)[R]) {
    if constexpr (R == 2) {
        float2 a = v[0], b = v[1];
        v[0] = cadd(a, b);
        v[1] = csub(a, b);
    } else if constexpr (R == 4) {
        float2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
        float2 t2 = cadd(v[1], v[3]), t3 = mul_mi(csub(v[1], v[3]));
        v[0] = cadd(t0, t2);
        v[1] = cadd(t1, t3);
        v[2] = csub(t0, t2);
        v[3] = csub(t1, t3);
    } else {
        float2 e[R / 2], o[R / 2];
#pragma unroll
        for (int i = 0; i < R / 2; ++i) {
            e[i] = v[2 * i];
            o[i] = v[2 * i + 1];
        }
        dft<R / 2>(e);
        dft<R / 2>(o);
#pragma unroll
        for (int k = 0; k < R / 2; ++k) {
            constexpr int step = 16 / R;
            // W_R^k = cos - i sin
            float2 w = make_float2(kC16[k * step], -kS16[k * step]);
            float2 t = (k == 0) ? o[k] : cmul(o[k], w);
            v[k] = cadd(e[k], t);
            v[k + R / 2] = csub(e[k], t);
        }
    }
}

// LDS padding: one complex slot per 8 keeps the stride-R writes of a stage and
// the unit-stride reads of the next on distinct banks.
__device__ __forceinline__ int lds_pad(int i) { return i + (i >> 3); }
constexpr int lds_padded(int n) { return n + (n >> 3); }

__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One Stockham stage.  x[q] = data[lane + 64 q] on entry and on exit.
// NS = product of the radices of the earlier stages.  tw: (P/R)*(R-1) per-lane
// twiddles exp(-2 pi i ((lane + 64u) mod NS) t / (NS R)), index u*(R-1) + t-1.
template <int P, int R, int NS, bool LAST>
__device__ __forceinline__ void fft_stage(float2 (&x)[P], const float2* tw, float2* lds, int lane) {
    constexpr int U = P / R;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        float2 v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = x[u + t * U];
        if constexpr (NS > 1) {
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = cmul(v[t], tw[u * (R - 1) + t - 1]);
        }
        dft<R>(v);
        if constexpr (LAST) {
#pragma unroll
            for (int t = 0; t < R; ++t) x[u + t * U] = v[t];
        } else {
            const int b = lane + kWave * u;
            const int base = (b / NS) * (NS * R) + (b % NS);
#pragma unroll
            for (int t = 0; t < R; ++t) lds[lds_pad(base + t * NS)] = v[t];
        }
    }
    if constexpr (!LAST) {
        wave_sync_lds();
#pragma unroll
        for (int q = 0; q < P; ++q) x[q] = lds[lds_pad(lane + kWave * q)];
        wave_sync_lds();
    }
}

// Per-size configuration: radix sequence and table sizes.
template <int LOG2N>
struct FftCfg;

template <>
struct FftCfg<11> {  // n_fft 2048: NC 1024 = 16 * 16 * 4
    static constexpr int P = 16, NSTAGE = 3, NTW = 15 + 4 * 3;
    static constexpr int radix(int s) { return s < 2 ? 16 : 4; }
};
template <>
struct FftCfg<10> {  // n_fft 1024: NC 512 = 8 * 8 * 8
    static constexpr int P = 8, NSTAGE = 3, NTW = 7 + 7;
    static constexpr int radix(int) { return 8; }
};
template <>
struct FftCfg<9> {  // n_fft 512: NC 256 = 4 * 4 * 4 * 4
    static constexpr int P = 4, NSTAGE = 4, NTW = 3 * 3;
    static constexpr int radix(int) { return 4; }
};
template <>
struct FftCfg<8> {  // n_fft 256: NC 128 = 2^7
    static constexpr int P = 2, NSTAGE = 7, NTW = 6;
    static constexpr int radix(int) { return 2; }
};

template <int LOG2N>
__device__ __forceinline__ void fft_frame(float2 (&x)[FftCfg<LOG2N>::P], const float2* tw, float2* lds, int lane) {
    if constexpr (LOG2N == 11) {
        fft_stage<16, 16, 1, false>(x, nullptr, lds, lane);
        fft_stage<16, 16, 16, false>(x, tw, lds, lane);
        fft_stage<16, 4, 256, true>(x, tw + 15, lds, lane);
    } else if constexpr (LOG2N == 10) {
        fft_stage<8, 8, 1, false>(x, nullptr, lds, lane);
        fft_stage<8, 8, 8, false>(x, tw, lds, lane);
        fft_stage<8, 8, 64, true>(x, tw + 7, lds, lane);
    } else if constexpr (LOG2N == 9) {
        fft_stage<4, 4, 1, false>(x, nullptr, lds, lane);
        fft_stage<4, 4, 4, false>(x, tw, lds, lane);
        fft_stage<4, 4, 16, false>(x, tw + 3, lds, lane);
        fft_stage<4, 4, 64, true>(x, tw + 6, lds, lane);
    } else {
        fft_stage<2, 2, 1, false>(x, nullptr, lds, lane);
        fft_stage<2, 2, 2, false>(x, tw + 0, lds, lane);
        fft_stage<2, 2, 4, false>(x, tw + 1, lds, lane);
        fft_stage<2, 2, 8, false>(x, tw + 2, lds, lane);
        fft_stage<2, 2, 16, false>(x, tw + 3, lds, lane);
        fft_stage<2, 2, 32, false>(x, tw + 4, lds, lane);
        fft_stage<2, 2, 64, true>(x, tw + 5, lds, lane);
    }
}

// Device tables of a plan, all [count][64] float2, lane-minor.
struct FftTables {
    const float2* tw;    // [NTW][64]   stage twiddles
    const float2* post;  // [P/2][64]   exp(-2 pi i (lane + 64 q) / n_fft)
    const float2* win;   // [P][64]     (hann[2n], hann[2n+1]), n = lane + 64 q
};

}  // namespace iris
