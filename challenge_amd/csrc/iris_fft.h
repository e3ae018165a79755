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
//
// A complex number is a 2-wide float vector (one 64-bit register pair), so that a
// complex add/sub is ONE packed-f32 instruction (v_pk_add_f32) and a complex
// multiply is two (v_pk_mul_f32 + v_pk_fma_f32 with op_sel / neg modifiers).
#pragma once
#include <hip/hip_runtime.h>

// experiment switches (A/B builds only; the defaults are the product)
#ifndef IRIS_OLD_CMUL
#define IRIS_OLD_CMUL 0
#endif
// one ds_read_b64 per point (256 B/clk) instead of the merged ds_read2_b64 (128 B/clk) in the exchanges of
// this n_fft (log2; 0 = none).  Measured per shape (scripts/gpu_shapes.py): n_fft 1024 -2.5 % (c2 17.9 -> 17.3 us),
// every other size 1-5 % slower (twice the DS instructions to issue, fewer waves to hide them at 2048)
#ifndef IRIS_SINGLE_READS_LOG2N
#define IRIS_SINGLE_READS_LOG2N 10
#endif

namespace iris {

constexpr int kWave = 64;

typedef float cf __attribute__((ext_vector_type(2)));  // (re, im)
typedef __attribute__((address_space(3))) cf lds_cf;     // the same in LDS, for accesses that must name the address space

__device__ __forceinline__ cf mk(float re, float im) {
    cf r = {re, im};
    return r;
}
// a + (-i) b  and  a - (-i) b      ((-i) b = (b.im, -b.re)).  Written as one packed FMA
// with a (+-1, -+1) constant: the swap of b folds into op_sel, the sign sits in the
// constant, the product by +-1 is exact -- so each is ONE instruction.
__device__ __forceinline__ cf add_mi(cf a, cf b) { return __builtin_elementwise_fma(b.yx, mk(1.0f, -1.0f), a); }
__device__ __forceinline__ cf sub_mi(cf a, cf b) { return __builtin_elementwise_fma(b.yx, mk(-1.0f, 1.0f), a); }
// a * w = a.re * w + a.im * (-w.im, w.re)
__device__ __forceinline__ cf cmul(cf a, cf w) {
    const cf t = a.yy * mk(-w.y, w.x);
    return __builtin_elementwise_fma(a.xx, w, t);
}

// a * w for a per-lane twiddle w that lives in registers across the frame loop.  Written with
// the VOP3P operand modifiers by hand: in C++ the swapped / negated copy (-w.im, w.re) is loop
// invariant, so the compiler hoists it into a second register pair per twiddle (36 VGPRs at
// n_fft 1024 - the difference between 3 and 4 waves per SIMD); op_sel / neg_lo read it out of w.
//   t = (a.im * -w.im, a.im * w.re);  r = (a.re * w.re + t.re, a.re * w.im + t.im)
__device__ __forceinline__ cf cmul_tw(cf a, cf w) {
#if IRIS_OLD_CMUL
    return cmul(a, w);
#endif
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(a), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(r) : "v"(a), "v"(w), "v"(t));
    return r;
}
// (-i d) * w  with  -i d = (d.im, -d.re): the rotation folds into the modifiers as well.
//   t = (d.re * w.im, -d.re * w.re);  r = (d.im * w.re + t.re, d.im * w.im + t.im)
__device__ __forceinline__ cf cmul_mi_tw(cf d, cf w) {
#if IRIS_OLD_CMUL
    return cmul(mk(d.y, -d.x), w);
#endif
    cf t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,0] neg_hi:[0,1]" : "=v"(t) : "v"(d), "v"(w));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(d), "v"(w), "v"(t));
    return r;
}

// cos/sin(2 pi k / 16), k = 0..7
__device__ constexpr float kC16[8] = {1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                                      0.0f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
__device__ constexpr float kS16[8] = {0.0f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                                      1.0f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};

// In-place forward DFT of R points, natural-order output.
template <int R>
__device__ __forceinline__ void dft(cf (&v)[R]) {
    if constexpr (R == 2) {
        const cf a = v[0], b = v[1];
        v[0] = a + b;
        v[1] = a - b;
    } else if constexpr (R == 4) {
        const cf t0 = v[0] + v[2], t1 = v[0] - v[2];
        const cf t2 = v[1] + v[3], t3 = v[1] - v[3];
        v[0] = t0 + t2;
        v[1] = add_mi(t1, t3);
        v[2] = t0 - t2;
        v[3] = sub_mi(t1, t3);
    } else if constexpr (R == 8) {
        // decimation in frequency: 24 complex add/sub (one packed instruction each) + two
        // rotations by (1 -+ i)/sqrt(2) (one packed add + one packed multiply each)
        constexpr float r = 0.70710678118654752f;
        const cf a0 = v[0] + v[4], a1 = v[1] + v[5], a2 = v[2] + v[6], a3 = v[3] + v[7];
        const cf d0 = v[0] - v[4], d1 = v[1] - v[5], d2 = v[2] - v[6], d3 = v[3] - v[7];
        {
            const cf s0 = a0 + a2, s1 = a1 + a3, t0 = a0 - a2, t1 = a1 - a3;
            v[0] = s0 + s1;
            v[4] = s0 - s1;
            v[2] = add_mi(t0, t1);
            v[6] = sub_mi(t0, t1);
        }
        {
            const cf e1 = add_mi(d1, d1) * r;   // W8^1 d1 = r (d.re + d.im, d.im - d.re)
            const cf e3 = sub_mi(d3, d3) * -r;  // W8^3 d3 = -r (d.re - d.im, d.re + d.im)
            const cf s0 = add_mi(d0, d2), t0 = sub_mi(d0, d2);
            const cf s1 = e1 + e3, t1 = e1 - e3;
            v[1] = s0 + s1;
            v[5] = s0 - s1;
            v[3] = add_mi(t0, t1);
            v[7] = sub_mi(t0, t1);
        }
    } else {
        cf e[R / 2], o[R / 2];
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
            const cf w = mk(kC16[k * step], -kS16[k * step]);  // W_R^k = cos - i sin
            const cf t = (k == 0) ? o[k] : cmul(o[k], w);
            v[k] = e[k] + t;
            v[k + R / 2] = e[k] - t;
        }
    }
}

// LDS padding.  pad(i) = i + PM * (i >> SH) moves whole blocks of 2^SH points, per exchange:
//   * ds_write_b64 is serviced 16 lanes at a time over 16 slots of 8 bytes (bank = (a / 4) mod 32) and occupies its
//     issue path for ~6 cycles whatever the LDS array does, so the strided writes of a stage want their 16 lanes on
//     16 distinct slots (complex index mod 16);
//   * a single ds_read_b64 is serviced 32 lanes at a time over 32 slots (bank = (a / 4) mod 64): the unit-stride reads
//     of the next stage are conflict-free only if the 32 consecutive points of a lane group stay consecutive, i.e. no
//     pad inside an aligned block of 32 (merged ds_read2_b64 reads are serviced like the writes, 16 at a time).
//   NS == 1          PM 1, SH 4: the writes have stride R, a pad after every 16 points is what spreads them; a 32-lane
//                    read group then straddles one pad (one 2-way conflict per group - no padding satisfies both sides;
//                    brute-force search over PM <= 16, SH 3..6: scripts/microbench/lds_pad_search.py)
//   NS >= 16         PM 1, SH 4 (16 consecutive lanes write 16 consecutive points: any block pad works)
//   R == 8, NS == 8  PM 4, SH 5: lanes b and b + 8 write 64 points apart = 2 blocks of 32 = 8 slots apart (mod 16), and
//                    nothing is padded inside a block of 32: writes AND 32-lane reads conflict-free (PM 2, SH 4 - round
//                    1 to 3 - left every read group with a conflict: 32 instead of 16 LDS cycles per exchange)
//   otherwise        PM = 16 / R (R >= 4) or NS (R == 2), SH 4
// Every address a stage touches is pad(base) + compile-time offset (no carry into the block index), i.e. one base VGPR
// per access pattern + immediate offsets.
struct PadCfg {
    int pm, sh;
};
template <int PM, int SH = 4>
__device__ __forceinline__ constexpr int lds_pad(int i) { return i + PM * (i >> SH); }
constexpr PadCfg stage_pad(int NS, int R) {
    // (NS >= 16 could go unpadded, but only n_fft 1024 reads with single ds_read_b64 - every other size uses the merged reads,
    // which are serviced 16 lanes at a time and have no conflict to lose - and at n_fft 2048 the unpadded form cost the
    // 168-register variant 20 bytes of scratch: those exchanges keep PM 1)
    return (NS == 1 || NS >= 16) ? PadCfg{1, 4} : ((R == 8 && NS == 8) ? PadCfg{4, 5} : PadCfg{R == 2 ? NS : 16 / R, 4});
}
// the untangle exchange (unit-stride writes, reversed unit-stride reads): unpadded where the reads are single ds_read_b64
// (32 lanes read 32 consecutive points downwards = 32 distinct slots), PM 1 elsewhere (as rounds 1 to 3)
constexpr int untangle_pm(int log2n) { return log2n == IRIS_SINGLE_READS_LOG2N ? 0 : 1; }
constexpr int lds_padded(int n, int pm_max) { return n + pm_max * (n >> 4) + 1; }  // +1: slot NC is addressable

// Orders this wave's LDS accesses for the compiler; a wave's DS instructions execute
// in issue order, so no hardware wait is needed between a write and a dependent read.
__device__ __forceinline__ void wave_sync_lds() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// One Stockham stage, split so that several independent frames ("streams") of one wave can be
// interleaved: stage_fwd = twiddles + butterflies (+ the strided LDS writes unless LAST),
// stage_load = the unit-stride reads that bring the data back to x[q] = data[lane + 64 q].
// NS = product of the radices of the earlier stages.  tw: (P/R)*(R-1) per-lane twiddles
// exp(-2 pi i ((lane + 64u) mod NS) t / (NS R)), index u*(R-1) + t-1.
template <int P, int R, int NS, bool LAST>
__device__ __forceinline__ void stage_fwd(cf (&x)[P], const cf* tw, cf* lds, int lane) {
    constexpr int U = P / R, PM = stage_pad(NS, R).pm, SH = stage_pad(NS, R).sh;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        cf v[R];
#pragma unroll
        for (int t = 0; t < R; ++t) v[t] = x[u + t * U];
        if constexpr (NS > 1) {
#pragma unroll
            for (int t = 1; t < R; ++t) v[t] = cmul_tw(v[t], tw[u * (R - 1) + t - 1]);
        }
        dft<R>(v);
        if constexpr (LAST) {
#pragma unroll
            for (int t = 0; t < R; ++t) x[u + t * U] = v[t];
        } else {
            const int b = lane + kWave * u;
            cf* wp = lds + lds_pad<PM, SH>((b / NS) * (NS * R) + (b % NS));
#pragma unroll
            for (int t = 0; t < R; ++t) wp[lds_pad<PM, SH>(t * NS)] = v[t];
        }
    }
}

template <int P, int R, int NS, bool SINGLE = false>
__device__ __forceinline__ void stage_load(cf (&x)[P], const cf* lds, int lane) {
    constexpr int PM = stage_pad(NS, R).pm, SH = stage_pad(NS, R).sh;
    const cf* rp = lds + lds_pad<PM, SH>(lane);
    if constexpr (SINGLE) {
        // volatile: keeps one ds_read_b64 per point (2 LDS cycles per 512 B) instead of the merged
        // ds_read2_b64 (8 cycles per 1 KiB)
        const volatile lds_cf* vp = (const volatile lds_cf*)rp;
#pragma unroll
        for (int q = 0; q < P; ++q) x[q] = vp[lds_pad<PM, SH>(kWave * q)];
    } else {
#pragma unroll
        for (int q = 0; q < P; ++q) x[q] = rp[lds_pad<PM, SH>(kWave * q)];
    }
}

// The same stage for S streams: all butterflies and writes, one ordering point, all reads.
// timing experiment only (results are wrong): every stage keeps its outputs in registers, no LDS exchange
#ifndef IRIS_NO_EXCHANGE
#define IRIS_NO_EXCHANGE 0
#endif
template <int S, int P, int R, int NS, bool LAST_, bool SINGLE = false>
__device__ __forceinline__ void stage_multi(cf (&x)[S][P], const cf* tw, cf* const (&lds)[S], int lane) {
    constexpr bool LAST = LAST_ || (IRIS_NO_EXCHANGE != 0);
#pragma unroll
    for (int s = 0; s < S; ++s) stage_fwd<P, R, NS, LAST>(x[s], tw, lds[s], lane);
    if constexpr (!LAST) {
        wave_sync_lds();
#pragma unroll
        for (int s = 0; s < S; ++s) stage_load<P, R, NS, SINGLE>(x[s], lds[s], lane);
        wave_sync_lds();
    }
}

// Per-size configuration: points per lane, table sizes, PMMAX = bound of the padded extent over all exchanges in units of
// NC / 16 points (n_fft 1024: max(NC / 16 [PM 1, SH 4], 4 NC / 32 [PM 4, SH 5]) = 2 NC / 16).
template <int LOG2N>
struct FftCfg;

template <>
struct FftCfg<11> {  // n_fft 2048: NC 1024 = 16 * 16 * 4
    static constexpr int P = 16, NSTAGE = 3, NTW = 15 + 4 * 3, PMMAX = 1;
};
template <>
struct FftCfg<10> {  // n_fft 1024: NC 512 = 8 * 8 * 8
    static constexpr int P = 8, NSTAGE = 3, NTW = 7 + 7, PMMAX = 2;
};
template <>
struct FftCfg<9> {  // n_fft 512: NC 256 = 4 * 4 * 4 * 4
    static constexpr int P = 4, NSTAGE = 4, NTW = 3 * 3, PMMAX = 4;
};
template <>
struct FftCfg<8> {  // n_fft 256: NC 128 = 2^7
    static constexpr int P = 2, NSTAGE = 7, NTW = 6, PMMAX = 8;
};

// Complex FFT of S frames held by one wave (S = 1: one frame; S = 2: two frames in flight, so
// that the LDS round trip of one hides behind the butterflies of the other).
// SINGLE: one ds_read_b64 per point in the exchanges (see IRIS_SINGLE_READS_LOG2N; the fused kernel asks for it)
template <int LOG2N, int S, bool SINGLE = false>
__device__ __forceinline__ void fft_frames(cf (&x)[S][FftCfg<LOG2N>::P], const cf* tw, cf* const (&lds)[S],
                                           int lane) {
    if constexpr (LOG2N == 11) {
        stage_multi<S, 16, 16, 1, false, SINGLE>(x, nullptr, lds, lane);
        stage_multi<S, 16, 16, 16, false, SINGLE>(x, tw, lds, lane);
        stage_multi<S, 16, 4, 256, true, SINGLE>(x, tw + 15, lds, lane);
    } else if constexpr (LOG2N == 10) {
        stage_multi<S, 8, 8, 1, false, SINGLE>(x, nullptr, lds, lane);
        stage_multi<S, 8, 8, 8, false, SINGLE>(x, tw, lds, lane);
        stage_multi<S, 8, 8, 64, true, SINGLE>(x, tw + 7, lds, lane);
    } else if constexpr (LOG2N == 9) {
        stage_multi<S, 4, 4, 1, false, SINGLE>(x, nullptr, lds, lane);
        stage_multi<S, 4, 4, 4, false, SINGLE>(x, tw, lds, lane);
        stage_multi<S, 4, 4, 16, false, SINGLE>(x, tw + 3, lds, lane);
        stage_multi<S, 4, 4, 64, true, SINGLE>(x, tw + 6, lds, lane);
    } else {
        stage_multi<S, 2, 2, 1, false, SINGLE>(x, nullptr, lds, lane);
        stage_multi<S, 2, 2, 2, false, SINGLE>(x, tw + 0, lds, lane);
        stage_multi<S, 2, 2, 4, false, SINGLE>(x, tw + 1, lds, lane);
        stage_multi<S, 2, 2, 8, false, SINGLE>(x, tw + 2, lds, lane);
        stage_multi<S, 2, 2, 16, false, SINGLE>(x, tw + 3, lds, lane);
        stage_multi<S, 2, 2, 32, false, SINGLE>(x, tw + 4, lds, lane);
        stage_multi<S, 2, 2, 64, true, SINGLE>(x, tw + 5, lds, lane);
    }
}

template <int LOG2N>
__device__ __forceinline__ void fft_frame(cf (&x)[FftCfg<LOG2N>::P], const cf* tw, cf* lds, int lane) {
    cf* const one[1] = {lds};
    fft_frames<LOG2N, 1>(reinterpret_cast<cf(&)[1][FftCfg<LOG2N>::P]>(x), tw, one, lane);
}

}  // namespace iris
