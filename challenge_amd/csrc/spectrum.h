// spectrum.h -- frame -> spectrum pieces shared by the fused and the STFT kernels, and the
// per-lane constant block.  Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// frame -> spectrum pieces shared by the fused and the STFT kernels
// ---------------------------------------------------------------------------
template <int LOG2N>
__device__ __forceinline__ void load_frame(cf (&x)[FftCfg<LOG2N>::P], const float* clip, int len, int start,
                                           int lane) {
    constexpr int N = 1 << LOG2N, P = FftCfg<LOG2N>::P;
    const bool interior = (start >= 0) && (start + N <= len) &&
                          ((reinterpret_cast<uintptr_t>(clip + start) & 7) == 0);
    if (interior) {  // wave-uniform
        const cf* p = reinterpret_cast<const cf*>(clip + start);
#pragma unroll
        for (int q = 0; q < P; ++q) x[q] = p[lane + kWave * q];
    } else {
        // rare (clip edges, odd alignment): per-lane reflected indices as 32-bit byte offsets
        // from the uniform clip base, so no 64-bit address lives in VGPRs
        const char* base = reinterpret_cast<const char*>(clip);
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int n = start + 2 * (lane + kWave * q);
            const unsigned o0 = (unsigned)reflect_idx(n, len) * 4u, o1 = (unsigned)reflect_idx(n + 1, len) * 4u;
            x[q] = mk(*reinterpret_cast<const float*>(base + o0), *reinterpret_cast<const float*>(base + o1));
        }
    }
}

// x[q] = Z[lane + 64 q] -> Xlo[q] = X[k], Xhi[q] = X[NC - k], k = lane + 64 q, q < P/2.
// HALF = false leaves out the factor 0.5 (outputs are 2 X).  Uses the wave's LDS
// buffer; ends with the buffer free for reuse.
template <int LOG2N, bool HI, bool HALF, int S>
__device__ __forceinline__ void untangle_multi(const cf (&x)[S][FftCfg<LOG2N>::P], const cf* post, cf* const (&lds)[S],
                                               int lane, cf (&xlo)[S][FftCfg<LOG2N>::P / 2],
                                               cf (&xhi)[S][FftCfg<LOG2N>::P / 2]) {
    constexpr int P = FftCfg<LOG2N>::P;
    // partners of k = lane + 64 q (q < P/2) are NC - k = (64 - lane) + 64 (P - 1 - q), i.e.
    // rows P/2 .. P-1 (lane 0 reads row P - q, lane 0): only the upper half is ever fetched
#pragma unroll
    for (int s = 0; s < S; ++s) {
        cf* wp = lds[s] + lds_pad<untangle_pm(LOG2N)>(lane);
#pragma unroll
        for (int q = P / 2; q < P; ++q) wp[lds_pad<untangle_pm(LOG2N)>(kWave * q)] = x[s][q];
    }
    wave_sync_lds();
#pragma unroll
    for (int s = 0; s < S; ++s) {
        // lane 0, q 0 pairs with itself (slot NC is addressable but unused)
        const cf* rp = lds[s] + lds_pad<untangle_pm(LOG2N)>(kWave - lane);
#pragma unroll
        for (int q = 0; q < P / 2; ++q) {
            const cf zk = x[s][q];
            cf zp = rp[lds_pad<untangle_pm(LOG2N)>(kWave * (P - 1 - q))];
            if (q == 0 && lane == 0) zp = zk;
            const cf zc = mk(zp.x, -zp.y);  // conj(Z[NC-k])
            cf e = zk + zc;                 // 2 E
            const cf d = zk - zc;           // 2 i O
            cf dh = d;                      // 2 i O  (O = -i dh / 2)
            if constexpr (HALF) {
                e *= 0.5f;
                dh *= 0.5f;
            }
            const cf wo = cmul_mi_tw(dh, post[q]);
            xlo[s][q] = e + wo;
            if constexpr (HI) {
                const cf t = e - wo;
                xhi[s][q] = mk(t.x, -t.y);
            }
        }
    }
    wave_sync_lds();
}

template <int LOG2N, bool HI, bool HALF>
__device__ __forceinline__ void untangle(const cf (&x)[FftCfg<LOG2N>::P], const cf* post, cf* lds, int lane,
                                         cf (&xlo)[FftCfg<LOG2N>::P / 2], cf (&xhi)[FftCfg<LOG2N>::P / 2]) {
    constexpr int P = FftCfg<LOG2N>::P;
    cf* const one[1] = {lds};
    untangle_multi<LOG2N, HI, HALF, 1>(reinterpret_cast<const cf(&)[1][P]>(x), post, one, lane,
                                       reinterpret_cast<cf(&)[1][P / 2]>(xlo), reinterpret_cast<cf(&)[1][P / 2]>(xhi));
}

__device__ __forceinline__ float cabs_rn(cf v) { return __builtin_amdgcn_sqrtf(fmaf(v.x, v.x, v.y * v.y)); }

// Untangle fused with the magnitude: x[q] = Z[lane + 64 q] -> mag[k] = 2 |X[k]| for k <= NC/2
// (HI: for every k <= NC), written to the wave's magnitude buffer (which aliases the low part of
// its exchange buffer: the partner rows P/2.. that are ever READ start at point NC/2 + 1, i.e. above byte 4 * (NC + 1),
// so magnitudes can land while partner reads are still queued - a wave's DS ops run in order).  The exchange is not
// padded: a 32-lane group reads points 64 m - lane, 32 consecutive points downwards = 32 distinct 8-byte slots.
// No complex outputs are kept: each bin's registers die as soon as its magnitude is stored.
// timing experiment only (results are wrong): partners taken from the lane's own registers, no LDS exchange
#ifndef IRIS_NO_UNTANGLE_X
#define IRIS_NO_UNTANGLE_X 0
#endif
template <int LOG2N, bool HI, int S>
__device__ __forceinline__ void untangle_mag(const cf (&x)[S][FftCfg<LOG2N>::P], const cf* post, cf* const (&lds)[S],
                                             float* const (&mag)[S], int lane) {
    constexpr int P = FftCfg<LOG2N>::P, NC = (1 << LOG2N) / 2;
    static_assert(8 * lds_pad<untangle_pm(LOG2N)>(NC / 2 + 1) >= 4 * (NC + 1), "magnitudes would overwrite partner rows");
#if !IRIS_NO_UNTANGLE_X
#pragma unroll
    for (int s = 0; s < S; ++s) {
        cf* wp = lds[s] + lds_pad<untangle_pm(LOG2N)>(lane);
#pragma unroll
        for (int q = P / 2; q < P; ++q) wp[lds_pad<untangle_pm(LOG2N)>(kWave * q)] = x[s][q];
    }
    wave_sync_lds();
#endif
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const cf* rp = lds[s] + lds_pad<untangle_pm(LOG2N)>(kWave - lane);
        cf zp[P / 2];
#if IRIS_NO_UNTANGLE_X
        (void)rp;
#pragma unroll
        for (int q = 0; q < P / 2; ++q) zp[q] = x[s][P - 1 - q];
#else
        if constexpr (LOG2N == IRIS_SINGLE_READS_LOG2N) {
            const volatile lds_cf* vp = (const volatile lds_cf*)rp;
#pragma unroll
            for (int q = 0; q < P / 2; ++q) zp[q] = vp[lds_pad<untangle_pm(LOG2N)>(kWave * (P - 1 - q))];
        } else {
#pragma unroll
            for (int q = 0; q < P / 2; ++q) zp[q] = rp[lds_pad<untangle_pm(LOG2N)>(kWave * (P - 1 - q))];
        }
#endif
        if (lane == 0) zp[0] = x[s][0];  // k = 0 pairs with itself
        if constexpr (HI) {
            if (lane == 0) mag[s][NC / 2] = 2.0f * cabs_rn(x[s][P / 2]);  // X[NC/2] = conj(Z[NC/2])
        }
        // Two bins at a time, then their stores: independent chains interleave (a packed op
        // that consumes the previous packed result costs a wait state on this chip) without
        // keeping the whole spectrum live.
        constexpr int PAIR = P / 2 >= 2 ? 2 : 1;  // n_fft 256 has one bin pair per lane
#pragma unroll
        for (int q0 = 0; q0 < P / 2; q0 += PAIR) {
            cf lo[PAIR], hi[PAIR];
#pragma unroll
            for (int j = 0; j < PAIR; ++j) {
                const int q = q0 + j;
                const cf zk = x[s][q];
                // zc = conj(Z[NC-k]); e = zk + zc, d = zk - zc as one packed FMA each (the sign pattern
                // rides on a (+-1, -+1) constant, exact)
                const cf e = __builtin_elementwise_fma(zp[q], mk(1.0f, -1.0f), zk);  // 2 E
                const cf d = __builtin_elementwise_fma(zp[q], mk(-1.0f, 1.0f), zk);  // 2 i O
                const cf wo = cmul_mi_tw(d, post[q]);
                lo[j] = e + wo;
                if constexpr (HI) hi[j] = e - wo;
            }
#pragma unroll
            for (int j = 0; j < PAIR; ++j) {
                const int q = q0 + j;
                mag[s][lane + kWave * q] = cabs_rn(lo[j]);
                if constexpr (HI) mag[s][NC - lane - kWave * q] = cabs_rn(hi[j]);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// per-lane constant block: every lane's twiddles / untangle twiddles / window /
// register mel weights, packed so that a wave fetches it with NV4 coalesced
// 16-byte loads issued back to back (one wait), layout [NV4][64 lanes][4 floats]
// ---------------------------------------------------------------------------
constexpr int kMelRegs = 20;  // register mel window: 5 x 16-byte LDS reads

template <int LOG2N>
struct ConstLayout {
    static constexpr int NTW = FftCfg<LOG2N>::NTW, P = FftCfg<LOG2N>::P;
    static constexpr int OFF_TW = 0, OFF_POST = 2 * NTW, OFF_WIN = OFF_POST + P, OFF_WREG = OFF_WIN + 2 * P,
                         OFF_LO = OFF_WREG + kMelRegs, NF = OFF_LO + 1, NV4 = (NF + 3) / 4;
};

template <int LOG2N>
__device__ __forceinline__ void load_consts(const float* consts, int lane, cf (&tw)[FftCfg<LOG2N>::NTW],
                                            cf (&post)[FftCfg<LOG2N>::P / 2], cf (&win)[FftCfg<LOG2N>::P],
                                            float (&wreg)[kMelRegs], int& lo0) {
    using CL = ConstLayout<LOG2N>;
    float cv[CL::NV4 * 4];
    const float4* src = reinterpret_cast<const float4*>(consts);
#pragma unroll
    for (int v = 0; v < CL::NV4; ++v) {
        const float4 t = src[v * kWave + lane];
        cv[4 * v + 0] = t.x;
        cv[4 * v + 1] = t.y;
        cv[4 * v + 2] = t.z;
        cv[4 * v + 3] = t.w;
    }
#pragma unroll
    for (int i = 0; i < CL::NTW; ++i) tw[i] = mk(cv[CL::OFF_TW + 2 * i], cv[CL::OFF_TW + 2 * i + 1]);
#pragma unroll
    for (int i = 0; i < CL::P / 2; ++i) post[i] = mk(cv[CL::OFF_POST + 2 * i], cv[CL::OFF_POST + 2 * i + 1]);
#pragma unroll
    for (int i = 0; i < CL::P; ++i) win[i] = mk(cv[CL::OFF_WIN + 2 * i], cv[CL::OFF_WIN + 2 * i + 1]);
#pragma unroll
    for (int i = 0; i < kMelRegs; ++i) wreg[i] = cv[CL::OFF_WREG + i];
    lo0 = __float_as_int(cv[CL::OFF_LO]);
}

// the register mel weights of this lane, straight from the global constant block
template <int LOG2N>
__device__ __forceinline__ void reload_wreg(const float* consts, int lane, float (&wreg)[kMelRegs]) {
    using CL = ConstLayout<LOG2N>;
#pragma unroll
    for (int i = 0; i < kMelRegs; ++i) {
        const int fi = CL::OFF_WREG + i;
        wreg[i] = consts[((fi / 4) * kWave + lane) * 4 + (fi % 4)];
    }
}
