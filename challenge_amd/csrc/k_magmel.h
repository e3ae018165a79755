// k_magmel.h -- spectrum -> mel (complex_to_magphase + magphase_to_mel fused).
// Part of the single translation unit iris_frontend.hip.
#pragma once
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

// K3b: streaming variant for triangular filterbanks (every bin feeds at most two adjacent
// bands, which is what linear_to_mel_weight_matrix produces): one thread per frame t walks the
// bins once with two open accumulators per channel; a band is written as soon as the walk
// has passed its last bin.  Every spectrum element is read exactly once, with one 8/16-byte
// load per bin (all 2C components), coalesced along t.
struct MagmelTriArgs {
    const float* spec;   // [B, F, T, 2C]
    float* mel;          // [B, M, T, C]
    const int* bin_band; // [F] first band fed by bin f (-1: none)
    const float* bin_w;  // [F][2] weights for bands bin_band[f] and bin_band[f] + 1
    const int* t_bands;
    int n_tb;
    const int* f_bands;
    int n_fb;
    int B, F, T, M, is_magphase, f_lo, f_hi;  // bins outside [f_lo, f_hi) feed nothing
};

template <int C>
__global__ __launch_bounds__(512) void k_magmel_tri(const MagmelTriArgs a) {
    typedef float vecT __attribute__((ext_vector_type(2 * C)));
    constexpr int U = 8;                // bins in flight per wave
    extern __shared__ float sm_mel[];   // [M][64][C] band sums of this block's 64 frames
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int nslice = blockDim.x >> 6;
    const int t0 = blockIdx.x * 64;
    const int t = t0 + lane;
    const bool valid = t < a.T;
    const int* tb = a.t_bands ? a.t_bands + (size_t)b * a.n_tb * 2 : nullptr;
    const int* fb = a.f_bands ? a.f_bands + (size_t)b * a.n_fb * 2 : nullptr;
    for (int i = threadIdx.x; i < a.M * 64 * C; i += blockDim.x) sm_mel[i] = 0.f;
    __syncthreads();

    // this wave's slice of the bins that feed anything
    const int nb = a.f_hi - a.f_lo;
    const int per = (nb + nslice - 1) / nslice;
    const int f0 = a.f_lo + wave * per;
    const int f1 = min(f0 + per, a.f_hi);
    const vecT* sp = reinterpret_cast<const vecT*>(a.spec) + (size_t)b * a.F * a.T + (valid ? t : 0);
    float acc0[C], acc1[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc0[c] = acc1[c] = 0.f;
    int cur = -1;  // band held in acc0 (acc1 holds cur + 1); wave-uniform
    auto retire = [&]() {  // add acc0 into the block sums, shift the window up by one band
#pragma unroll
        for (int c = 0; c < C; ++c) {
            atomicAdd(&sm_mel[((size_t)cur * 64 + lane) * C + c], acc0[c]);
            acc0[c] = acc1[c];
            acc1[c] = 0.f;
        }
        ++cur;
    };
    for (int fc = f0; fc < f1; fc += U) {
        vecT v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = min(fc + u, f1 - 1);
            v[u] = valid ? __builtin_nontemporal_load(&sp[(size_t)f * a.T]) : vecT(0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int f = fc + u;
            if (f >= f1) break;
            const int m = a.bin_band[f];  // uniform
            if (m < 0) continue;
            if (cur < 0) cur = m;
            while (cur < m) {
                if (cur + 1 < m && cur + 1 < a.M) {  // gap of more than one band: acc1 is retired too
                    retire();
                    retire();
                    cur = m;
                } else {
                    retire();
                }
            }
            float w0 = a.bin_w[2 * f], w1 = a.bin_w[2 * f + 1];
            if (fb && in_bands(fb, a.n_fb, f)) w0 = w1 = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float re = v[u][c];
                const float im = v[u][C + c];
                const float mag = a.is_magphase ? re : __builtin_amdgcn_sqrtf(fmaf(re, re, im * im));
                acc0[c] = fmaf(w0, mag, acc0[c]);
                acc1[c] = fmaf(w1, mag, acc1[c]);
            }
        }
    }
    if (cur >= 0) {
        retire();
        if (cur < a.M) retire();
    }
    __syncthreads();

    // write the block's [M][64][C] sums, coalesced along t
    float* out = a.mel + (size_t)b * a.M * a.T * C;
    const int row = 64 * C;
    for (int i = threadIdx.x; i < a.M * row; i += blockDim.x) {
        const int m = i / row, r = i - m * row;
        const int tt = t0 + r / C;
        if (tt >= a.T) continue;
        const bool tm = tb ? in_bands(tb, a.n_tb, tt) : false;
        out[(size_t)m * a.T * C + (size_t)t0 * C + r] = tm ? 0.f : sm_mel[i];
    }
}
