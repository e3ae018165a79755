// iris_frontend.hip -- HIP kernels + C ABI of the MI355X audio feature frontend.
// Written for gfx950 (CDNA4) only: 64-lane wavefronts, 160 KiB LDS per CU,
// 8 XCDs with private L2s.  See include/iris_frontend.h for the contract and
// DESIGN.md for the data layout and the roofline of each kernel.
#include "../../include/iris_frontend.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <vector>

#include "iris_fft.h"

using namespace iris;

// Diagnostic build (make diag): IRIS_ABLATE=<bits> skips phases of the fused kernel and
// records per-workgroup clock stamps.  In the product build every check folds away.
#ifndef IRIS_DIAG
#define IRIS_DIAG 0
#endif
#define ABL(bit) (IRIS_DIAG && (a.ablate & (bit)))
// 1: frames go global -> registers (prefetched during the mel phase); 0: through LDS-DMA landing buffers
#ifndef IRIS_DIRECT_LOAD
#define IRIS_DIRECT_LOAD 1
#endif
// diagnostic buffer: [4] header, [3 * 4096] per-workgroup stamps, [4096 * 16 * 16] per-wave phase cycles
static constexpr int kDbgPhase0 = 4 + 3 * 4096, kDbgWords = kDbgPhase0 + 4096 * 16 * 16;
#if IRIS_DIAG
#define PH_BEGIN() do { if (ABL(4096)) ph_t = __builtin_amdgcn_s_memtime(); } while (0)
#define PH_MARK(i) do { if (ABL(4096)) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); ph[i] += n_ - ph_t; ph_t = n_; } } while (0)
#else
#define PH_BEGIN() do {} while (0)
#define PH_MARK(i) do {} while (0)
#endif

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
    bool mel_only;  // n_fft == 0: only iris_magmel is available
    float sample_rate, lower_hz, upper_hz;
    // host copies
    std::vector<float> mel;  // [F][M]
    int max_band_len, k_need;
    // device tables
    float* d_consts;  // per-lane constant block [NV4][64][4]
    int* d_band_lo;   // [M] first non-zero bin of each band (magmel)
    int* d_band_len;  // [M]
    float* d_mel;     // [F][M] dense (magmel)
    int* d_bin_band;  // [F] magmel streaming kernel: first band fed by each bin (-1 none)
    float* d_bin_w;   // [F][2] its two weights
    int tri_ok, tri_f_lo, tri_f_hi;  // filterbank is triangular-sparse (<= 2 adjacent bands per bin)
    int* d_fband_lo;  // [M] fused kernel: first bin read, clamped so lo + rows <= limit
    float* d_wband;   // [rows][M] fused kernel: 0.5 * W[lo + i][m]
    int rows, need_hi, mel_mode;
    float* d_ws;  // workspace
    unsigned long long* d_dbg;  // diagnostic stamps
    int streams;                // IRIS_STREAMS: frames in flight per wave (1 or 2)
    size_t ws_floats;
    int num_cu;
    int chunk_target;  // 0 = auto; frames per chunk of the fused kernel (IRIS_CHUNK_FRAMES)
    // timing
    int timing;        // 0 off, n: every n-th launch carries an event pair
    long launch_no;    // launches since timing was enabled
    std::vector<hipEvent_t> ev;  // pairs
    int ev_used;
};

constexpr int kChunk = 4096;        // elements per partial-reduction block
constexpr int kMaxTimedLaunches = 4096;

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
// Hides a wave-uniform pointer from loop-invariant code motion: addresses derived from it are
// computed where they are used instead of being hoisted (and spilled) across the frame loop.
template <typename T>
__device__ __forceinline__ T* opaque(T* p) {
    asm volatile("" : "+s"(p));
    return p;
}

__device__ __forceinline__ int reflect_idx(int i, int len) {
    i = i < 0 ? -i : i;
    return i >= len ? 2 * (len - 1) - i : i;
}

// Wave-wide reductions on the DPP network (no LDS traffic): four row_shr steps leave each
// 16-lane row's result in its last lane, row_bcast:15 / row_bcast:31 carry it across rows, lane 63
// ends up with the whole wave's value, which is returned to every lane.  Lanes/rows a step does not
// reach keep their own value (the `old` operand), which is harmless for min, max and - with a
// zero `old` - for sums.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_take(float old, float v) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROWMASK, 0xf, false));
}
__device__ __forceinline__ float lane63(float v) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_min(float v) {
    v = fminf(v, dpp_take<0x111, 0xf>(v, v));  // row_shr:1
    v = fminf(v, dpp_take<0x112, 0xf>(v, v));  // row_shr:2
    v = fminf(v, dpp_take<0x114, 0xf>(v, v));  // row_shr:4
    v = fminf(v, dpp_take<0x118, 0xf>(v, v));  // row_shr:8
    v = fminf(v, dpp_take<0x142, 0xa>(v, v));  // row_bcast:15 into rows 1 and 3
    v = fminf(v, dpp_take<0x143, 0xc>(v, v));  // row_bcast:31 into rows 2 and 3
    return lane63(v);
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_take<0x111, 0xf>(v, v));
    v = fmaxf(v, dpp_take<0x112, 0xf>(v, v));
    v = fmaxf(v, dpp_take<0x114, 0xf>(v, v));
    v = fmaxf(v, dpp_take<0x118, 0xf>(v, v));
    v = fmaxf(v, dpp_take<0x142, 0xa>(v, v));
    v = fmaxf(v, dpp_take<0x143, 0xc>(v, v));
    return lane63(v);
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_take<0x111, 0xf>(0.f, v);
    v += dpp_take<0x112, 0xf>(0.f, v);
    v += dpp_take<0x114, 0xf>(0.f, v);
    v += dpp_take<0x118, 0xf>(0.f, v);
    v += dpp_take<0x142, 0xa>(0.f, v);
    v += dpp_take<0x143, 0xc>(0.f, v);
    return lane63(v);
}

// Block-wide min/max (up to 16 waves); result valid in every thread.  red: 32 floats.
__device__ __forceinline__ void block_minmax(float& mn, float& mx, float* red /*[32]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    mn = wave_min(mn);
    mx = wave_max(mx);
    __syncthreads();
    if (lane == 0) {
        red[w] = mn;
        red[16 + w] = mx;
    }
    __syncthreads();
    mn = red[0];
    mx = red[16];
    for (int i = 1; i < nw; ++i) {
        mn = fminf(mn, red[i]);
        mx = fmaxf(mx, red[16 + i]);
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
        cf* wp = lds[s] + lds_pad<1>(lane);
#pragma unroll
        for (int q = P / 2; q < P; ++q) wp[lds_pad<1>(kWave * q)] = x[s][q];
    }
    wave_sync_lds();
#pragma unroll
    for (int s = 0; s < S; ++s) {
        // lane 0, q 0 pairs with itself (slot NC is addressable but unused)
        const cf* rp = lds[s] + lds_pad<1>(kWave - lane);
#pragma unroll
        for (int q = 0; q < P / 2; ++q) {
            const cf zk = x[s][q];
            cf zp = rp[lds_pad<1>(kWave * (P - 1 - q))];
            if (q == 0 && lane == 0) zp = zk;
            const cf zc = mk(zp.x, -zp.y);  // conj(Z[NC-k])
            cf e = zk + zc;                 // 2 E
            const cf d = zk - zc;           // 2 i O
            cf o = mk(d.y, -d.x);           // 2 O
            if constexpr (HALF) {
                e *= 0.5f;
                o *= 0.5f;
            }
            const cf wo = cmul(o, post[q]);
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
// its exchange buffer: the partner rows P/2.. live above byte 8 * lds_pad(NC/2) > 4 * (NC + 1),
// so magnitudes can land while partner reads are still queued - a wave's DS ops run in order).
// No complex outputs are kept: each bin's registers die as soon as its magnitude is stored.
template <int LOG2N, bool HI, int S>
__device__ __forceinline__ void untangle_mag(const cf (&x)[S][FftCfg<LOG2N>::P], const cf* post, cf* const (&lds)[S],
                                             float* const (&mag)[S], int lane) {
    constexpr int P = FftCfg<LOG2N>::P, NC = (1 << LOG2N) / 2;
    static_assert(8 * lds_pad<1>(NC / 2) >= 4 * (NC + 1), "magnitudes would overwrite partner rows");
#pragma unroll
    for (int s = 0; s < S; ++s) {
        cf* wp = lds[s] + lds_pad<1>(lane);
#pragma unroll
        for (int q = P / 2; q < P; ++q) wp[lds_pad<1>(kWave * q)] = x[s][q];
    }
    wave_sync_lds();
#pragma unroll
    for (int s = 0; s < S; ++s) {
        const cf* rp = lds[s] + lds_pad<1>(kWave - lane);
        cf zp[P / 2];
#pragma unroll
        for (int q = 0; q < P / 2; ++q) zp[q] = rp[lds_pad<1>(kWave * (P - 1 - q))];
        if (lane == 0) zp[0] = x[s][0];  // k = 0 pairs with itself
        if constexpr (HI) {
            if (lane == 0) mag[s][NC / 2] = 2.0f * cabs_rn(x[s][P / 2]);  // X[NC/2] = conj(Z[NC/2])
        }
        // Two bins at a time, then their stores: independent chains interleave (a packed op
        // that consumes the previous packed result costs a wait state on this chip) without
        // keeping the whole spectrum live.
#pragma unroll
        for (int q0 = 0; q0 < P / 2; q0 += 2) {
            cf lo[2], hi[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int q = q0 + j;
                const cf zk = x[s][q];
                const cf zc = mk(zp[q].x, -zp[q].y);  // conj(Z[NC-k])
                const cf e = zk + zc;                 // 2 E
                const cf d = zk - zc;                 // 2 i O
                const cf wo = cmul(mk(d.y, -d.x), post[q]);
                lo[j] = e + wo;
                if constexpr (HI) hi[j] = e - wo;
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
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

// ---------------------------------------------------------------------------
// K1: fused wav -> mel magnitudes (+ per-wave min/max partials)
//   work unit = chunk: consecutive frames of one clip, all C channels
//   grid      = min(#chunks, #CUs) workgroups of 12 waves (n_fft 2048: 8) looping over chunks
//   per wave  = one frame at a time, claimed from the chunk's LDS queue:
//                 LDS-DMA (global_load_lds) of the NEXT frame into the wave's landing
//                 buffer -- no VGPRs, reflect padding resolved in the DMA's per-lane
//                 source address -- while the current frame is windowed, transformed
//                 (registers + private padded LDS exchanges), untangled, |X| written
//                 to LDS and reduced over the banded mel weights; lane m stores band m of
//                 the frame straight to out[b, m, t, c] (the L2 merges the 4-byte stores)
//   LDS       = landing buffers [waves][N floats] | exchange buffers [waves] | frame queue |
//               mel table (mode 1); after the prologue the waves share nothing but the queue
//   MELMODE 0 = band weights in registers (M <= 64, band length <= 16): each lane reads
//               a 16-byte-aligned window of 20 magnitudes with 5 ds_read_b128
//           1 = band table staged in LDS, 2 = band table read from global (L1/L2)
//   HI        = some band needs bins above n_fft/4 (both halves of the untangle)
//   BANDS     = SpecAugment / filter bands present
// ---------------------------------------------------------------------------
// waves per workgroup: one workgroup per CU holding every wave of the CU, so that all waves are
// of one age class for the issue arbiter (which favours older waves) and share one frame queue
// workgroups per CU (= waves per SIMD): 3 -> <= 168 VGPRs; n_fft 2048 keeps 16 points per
// lane and needs the 256-VGPR budget of 2
constexpr int fused_occ(int log2n) { return 1; }
// waves per workgroup: 3 per SIMD with one frame per wave (168 VGPRs); 2 per SIMD when a wave keeps
// two frames in flight or at n_fft 2048 (256 VGPRs)
constexpr int fused_waves(int log2n, int streams = 1) {
    return (log2n >= 11 || streams > 1) ? 8 : (log2n <= 9 ? 16 : 12);
}

struct FusedArgs {
    const float* wav;    // [B, C, L]
    float* out;          // [B, M, T, C]
    float* partial;      // [B, chunks_per_clip * waves, 2] (min, max) per wave of each chunk
    const float* sumsq;  // nullable [B, n_sq] partial sums of squares (normalize)
    int n_sq;
    const float* consts;  // per-lane constant block (ConstLayout)
    const int* band_lo;   // [M] first bin read by band m (clamped so lo + rows <= limit)
    const float* wband;   // [rows][M], 0.5 * W[lo + i][m]
    int rows;
    const int* t_bands;  // nullable [B, n_tb, 2]
    int n_tb;
    const int* f_bands;  // nullable [B, n_fb, 2]
    int n_fb;
    int B, C, L, T, hop, M;
    int chunk_frames, chunks_per_clip, n_chunks;
    int chunk_base, chunk_rem;  // T = chunks_per_clip * chunk_base + chunk_rem; the first chunk_rem chunks take one more
    int ablate;  // diagnostic only (IRIS_ABLATE): skip phases, results are wrong when non-zero
    unsigned long long* dbg;  // diagnostic only: [4] shader-clock / 100 MHz stamps of workgroup 0
};

// LDS-DMA of one frame.  Inline asm on purpose: hipcc drains an LDS-DMA it knows about
// (s_waitcnt vmcnt(0)) before the next DS access that might alias it, which would
// serialise the prefetch with the FFT.  Hidden from the compiler the DMA stays in flight
// across the whole frame computation; the kernel waits for it by hand right before it
// reads the frame buffer.  M0 = wave-uniform LDS byte address (saved / restored inside
// the statement); the instruction offset applies to the global and the LDS address alike.
//   dma_frame_x4: frame interior and 16-byte aligned -> N/256 pieces of 16 B per lane,
//                 source = SGPR base + lane*16 + imm
//   dma_frame_x1: any frame -> N/64 pieces of 4 B per lane with per-lane source
//                 addresses (reflect padding costs nothing extra)
template <int LOG2N>
__device__ __forceinline__ void dma_frame_x4(const float* src /*uniform*/, unsigned fbuf_lds, unsigned lane16) {
    static_assert(LOG2N >= 8 && LOG2N <= 11, "");
    unsigned keep;
    if constexpr (LOG2N == 8)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(src), "s"(fbuf_lds) : "memory");
    else if constexpr (LOG2N == 9)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(src), "s"(fbuf_lds) : "memory");
    else if constexpr (LOG2N == 10)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(src), "s"(fbuf_lds) : "memory");
    else {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(src), "s"(fbuf_lds) : "memory");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:2048\n\t"
                     "global_load_lds_dwordx4 %1, %2 offset:3072\n\t"
                     "s_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(lane16), "s"(src + 1024), "s"(fbuf_lds + 4096) : "memory");
    }
}

__device__ __forceinline__ void glds4(const float* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int LOG2N>
__device__ __forceinline__ void dma_frame_x1(const float* clip, int len, int start, unsigned fbuf_lds, int lane) {
    constexpr int N = 1 << LOG2N;
#pragma clang loop unroll(disable)
    for (int i = 0; i < N / 64; ++i)
        glds4(clip + reflect_idx(start + 64 * i + lane, len), __builtin_amdgcn_readfirstlane(fbuf_lds + 256 * i));
}

template <int LOG2N>
__device__ __forceinline__ void dma_frame(const float* clip, int len, int start, unsigned fbuf_lds, int lane) {
    constexpr int N = 1 << LOG2N;
    // clip/start are wave-uniform by construction; make that provable for the "s" operands
    const uint64_t u = reinterpret_cast<uint64_t>(clip + start);
    const uint32_t ulo = __builtin_amdgcn_readfirstlane((uint32_t)u);
    const uint32_t uhi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
    const float* src = reinterpret_cast<const float*>(((uint64_t)uhi << 32) | ulo);
    start = __builtin_amdgcn_readfirstlane(start);
    if ((start >= 0) && (start + N <= len) && ((ulo & 15u) == 0))
        dma_frame_x4<LOG2N>(src, fbuf_lds, (unsigned)lane * 16u);
    else
        dma_frame_x1<LOG2N>(clip, len, start, fbuf_lds, lane);
}

template <int LOG2N, int MELMODE, bool HI, bool BANDS, int S>
__global__ __launch_bounds__(64 * fused_waves(LOG2N, S), fused_waves(LOG2N, S) / 4) void k_wav_to_mel(const FusedArgs a) {
    constexpr int kFusedWaves = fused_waves(LOG2N, S);
    constexpr int N = 1 << LOG2N, NC = N / 2, P = FftCfg<LOG2N>::P, NTW = FftCfg<LOG2N>::NTW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the wave index is uniform: keep it (and everything derived from it) in SGPRs
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // LDS: [waves][S] landing buffers (LDS-DMA targets, N floats) | [waves][S] exchange buffers
    // (also |X|) | frame queue | MELMODE 1 tables.  Nothing is shared between waves but the queue.
    constexpr int kXBufBytes = (lds_padded(NC, FftCfg<LOG2N>::PMMAX) * 8 + 15) & ~15;
    constexpr int kLandBytes = kFusedWaves * S * N * 4;
    const float* fbuf[S];
    unsigned fbuf_lds[S];
    cf* lds[S];
    float* magbuf[S];
#pragma unroll
    for (int st = 0; st < S; ++st) {
        char* land = smem + (wv * S + st) * (N * 4);
        char* xb = smem + kLandBytes + (wv * S + st) * kXBufBytes;
        fbuf[st] = reinterpret_cast<const float*>(land);
        fbuf_lds[st] = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)land);
        lds[st] = reinterpret_cast<cf*>(xb);
        magbuf[st] = reinterpret_cast<float*>(xb);
    }
    char* xbuf0 = smem + kLandBytes;
    constexpr int kXAllBytes = kFusedWaves * S * kXBufBytes, kStageBytes = ConstLayout<LOG2N>::NV4 * kWave * 16;
    int* next_frame = reinterpret_cast<int*>(xbuf0 + (kXAllBytes > kStageBytes ? kXAllBytes : kStageBytes));  // [4]
    float* wtab = reinterpret_cast<float*>(next_frame + 4);  // MELMODE 1: [rows][M] then int lo[M]
    int* lotab = reinterpret_cast<int*>(wtab + a.rows * a.M);
    // BANDS: bit tl of this bitmap = frame t0 + tl of the current chunk lies in a time band
    unsigned* tbits = reinterpret_cast<unsigned*>(MELMODE == 1 ? reinterpret_cast<float*>(lotab + a.M) : wtab);
    // MELMODE 1: the chunk's band table, with the clip's frequency bands folded in (all threads)
    auto build_wtab = [&](const int* fbc) {
        for (int i = threadIdx.x; i < a.M; i += blockDim.x) lotab[i] = a.band_lo[i];
        for (int i = threadIdx.x; i < a.rows * a.M; i += blockDim.x) {
            float w = a.wband[i];
            if (BANDS && fbc) {
                const int r = i / a.M, m = i - r * a.M;
                if (in_bands(fbc, a.n_fb, a.band_lo[m] + r)) w = 0.f;
            }
            wtab[i] = w;
        }
    };
    auto build_tbits = [&](const int* tb, int t0, int nt) {  // all threads; publish with a barrier
        for (int base = 0; base < nt; base += blockDim.x) {
            const int i = base + threadIdx.x;
            const unsigned long long m = __ballot(i < nt && in_bands(tb, a.n_tb, t0 + i));
            if (lane == 0) {
                tbits[(base >> 5) + 2 * wv] = (unsigned)m;
                tbits[(base >> 5) + 2 * wv + 1] = (unsigned)(m >> 32);
            }
        }
    };

    unsigned long long real_entry = 0;
    if ABL(512) real_entry = __builtin_amdgcn_s_memrealtime();
    unsigned long long ph[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = 0;  // diag: cycles per phase
    (void)ph;
    (void)ph_t;
    unsigned long long stamp0 = 0, real0 = 0;

    cf tw[NTW], post[P / 2], win[P];  // per-lane constants, resident for the whole kernel
    float wreg[kMelRegs];
    int lo0 = 0;

    const int g0 = xcd_remap(blockIdx.x, gridDim.x);
    // chunk -> clip b, first frame t0, frame count nt (balanced split: sizes differ by at most one)
    auto chunk_clip = [&](int chunk) { return chunk / a.chunks_per_clip; };
    auto chunk_t0 = [&](int chunk, int b) {
        const int ci = chunk - b * a.chunks_per_clip;
        return ci * a.chunk_base + min(ci, a.chunk_rem);
    };
    auto chunk_nt = [&](int chunk, int b) {
        return a.chunk_base + ((chunk - b * a.chunks_per_clip) < a.chunk_rem ? 1 : 0);
    };
    constexpr bool DIRECT = IRIS_DIRECT_LOAD && LOG2N <= 10;  // n_fft 2048 (16 points per lane) has no registers to spare
    cf x[S][P];
    // Fetch of wave-frames ff[] (f = tl * C + c) of a chunk: straight into the x registers
    // (IRIS_DIRECT_LOAD), or by LDS-DMA into this wave's landing buffers
    auto issue_dma = [&](const int (&ff)[S], int b, int t0, int nwf) {
        const float* clip0 = a.wav + (size_t)b * a.C * a.L;
#pragma unroll
        for (int st = 0; st < S; ++st)
            if (ff[st] < nwf && !ABL(8)) {
                const int tl = (a.C == 1) ? ff[st] : ff[st] / a.C, c = ff[st] - tl * a.C;
                if constexpr (DIRECT)
                    load_frame<LOG2N>(x[st], clip0 + (size_t)c * a.L, a.L, (t0 + tl) * a.hop - N / 2, lane);
                else
                    dma_frame<LOG2N>(clip0 + (size_t)c * a.L, a.L, (t0 + tl) * a.hop - N / 2, fbuf_lds[st], lane);
            }
    };
    int f[S], fn[S];  // frames in registers / frames in flight to the landing buffers
    if (g0 < a.n_chunks) {  // first frames of the first chunk: in flight while the constants are fetched
        const int b = chunk_clip(g0);
#pragma unroll
        for (int st = 0; st < S; ++st) f[st] = wv * S + st;
        issue_dma(f, b, chunk_t0(g0, b), chunk_nt(g0, b) * a.C);
    }
    {
        // The constant block is the same for every wave: fetch it from global once per
        // workgroup, through the exchange buffers (idle until the first FFT).
        float4* stage = reinterpret_cast<float4*>(xbuf0);
        const float4* g = reinterpret_cast<const float4*>(a.consts);
        for (int i = threadIdx.x; i < ConstLayout<LOG2N>::NV4 * kWave; i += blockDim.x) stage[i] = g[i];
        if (threadIdx.x == 0) *next_frame = 2 * kFusedWaves * S;
        if constexpr (BANDS) {
            if (a.t_bands && g0 < a.n_chunks) {
                const int b = chunk_clip(g0);
                build_tbits(a.t_bands + (size_t)b * a.n_tb * 2, chunk_t0(g0, b), chunk_nt(g0, b));
            }
        }
        __syncthreads();
        load_consts<LOG2N>(reinterpret_cast<const float*>(stage), lane, tw, post, win, wreg, lo0);
        if constexpr (MELMODE == 1) {
            const int* fbc = nullptr;
            if constexpr (BANDS) {
                if (a.f_bands && g0 < a.n_chunks) fbc = a.f_bands + (size_t)chunk_clip(g0) * a.n_fb * 2;
            }
            build_wtab(fbc);
        }
        __syncthreads();
        if ABL(512) {
            stamp0 = __builtin_amdgcn_s_memtime();
            real0 = __builtin_amdgcn_s_memrealtime();
        }
    }
    for (int chunk = g0; chunk < a.n_chunks; chunk += gridDim.x) {
        PH_BEGIN();
        const int b = chunk_clip(chunk);
        const int t0 = chunk_t0(chunk, b), nt = chunk_nt(chunk, b);
        const int* tb = nullptr;
        const int* fb = nullptr;
        if constexpr (BANDS) {
            tb = a.t_bands ? a.t_bands + (size_t)b * a.n_tb * 2 : nullptr;
            fb = a.f_bands ? a.f_bands + (size_t)b * a.n_fb * 2 : nullptr;
        }
        const int nwf = nt * a.C;  // wave-frames in this chunk: f = tl * C + c

        // Each wave keeps S frames in flight ("streams").  Frames are claimed S at a time from
        // an LDS counter (waves that run ahead take more: the issue arbiter favours older
        // waves, a static split leaves the younger ones a tail).  All cursor state is
        // wave-uniform (SGPRs).  The loop is software-pipelined: while frame i is in its mel
        // phase (its samples are no longer needed in registers) the wave already reads frame
        // i+1 from its landing buffer and claims frame i+2, whose DMA is issued once those reads
        // have returned - neither the LDS round trip of the frame read nor the queue atomic
        // sits on the critical path.
#pragma unroll
        for (int st = 0; st < S; ++st) {
            f[st] = wv * S + st;
            fn[st] = (kFusedWaves + wv) * S + st;  // second round is static too: the queue starts at 2 * waves * S
        }
        if (chunk != g0) issue_dma(f, b, t0, nwf);
        bool mbit[S];             // the frames in f[] lie in a time band (wave-uniform)
#pragma unroll
        for (int st = 0; st < S; ++st) mbit[st] = false;
        if constexpr (BANDS) {
            // Frequency bands zero |X| over bin ranges, i.e. they remove those bins from every mel
            // band: fold them into this chunk's band weights once (register weights here, the
            // LDS table where the chunk starts) instead of touching the magnitudes of every frame.
            if constexpr (MELMODE == 0) {
                if (fb) {
                    if (chunk != g0) reload_wreg<LOG2N>(opaque(a.consts), lane, wreg);  // pristine weights (not hoisted)
                    for (int i = 0; i < a.n_fb; ++i) {  // band bounds are wave-uniform (scalar loads)
                        const int off = fb[2 * i] - lo0, end = off + fb[2 * i + 1];
#pragma unroll
                        for (int r = 0; r < kMelRegs; ++r)
                            if (r >= off && r < end) wreg[r] = 0.f;
                    }
                }
            }
            if (tb) {
#pragma unroll
                for (int st = 0; st < S; ++st) {
                    const int tl = min((a.C == 1) ? f[st] : f[st] / a.C, nt - 1);
                    mbit[st] = (__builtin_amdgcn_readfirstlane(tbits[tl >> 5]) >> (tl & 31)) & 1u;
                }
            }
        }

        float scale = 1.0f;  // normalize: |X| is linear in the waveform, so 1 / (10 rms) scales the mel
        if (a.sumsq != nullptr) {
            float sq = 0.f;
            const float* ssq = opaque(a.sumsq) + (size_t)b * a.n_sq;
            int l0 = lane;
            asm volatile("" : "+v"(l0));  // keep the (rarely used) per-lane address out of the loop's registers
            for (int i = l0; i < a.n_sq; i += kWave) sq += ssq[i];
            sq = wave_sum(sq);
            scale = 1.0f / (sqrtf(sq / ((float)a.C * (float)a.L)) * 10.0f);
        }

        // Output: lane m owns mel band m (+64, ...); a frame's M values go straight to
        // out[b, m, t, c] - 4-byte stores one row pitch apart, merged into full lines by the L2
        // (the whole output is a few MB).  No LDS tile, no workgroup barrier, no write-out phase:
        // after the prologue the waves only share the frame queue.
        // address = (uniform) out + ((b M T + t0) C + f) * 4  +  (per lane) m * T * C * 4
        const unsigned rowpitch_b = (unsigned)a.T * (unsigned)a.C * 4u;
        float* const chunk_out = a.out + ((size_t)b * a.M * a.T + t0) * a.C;
        auto store_band = [&](int fidx, unsigned off, float v) {
            if (!ABL(16))
                asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(v), "s"(chunk_out + fidx) : "memory");
        };
        float mn = INFINITY, mx = -INFINITY;

        auto read_frames = [&]() {  // landing buffers -> registers (asynchronous: lgkmcnt)
#pragma unroll
            for (int st = 0; st < S; ++st) {
                const cf* fb2 = reinterpret_cast<const cf*>(fbuf[st]) + lane;
#pragma unroll
                for (int q = 0; q < P; ++q) x[st][q] = fb2[kWave * q];
            }
        };
        if constexpr (!DIRECT) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            read_frames();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            issue_dma(fn, b, t0, nwf);
        }
        PH_MARK(8);
        while (f[0] < nwf) {
            PH_BEGIN();
            int fcur[S];
            bool live[S];  // stream holds a real frame (otherwise its results are dropped)
#pragma unroll
            for (int st = 0; st < S; ++st) {
                fcur[st] = f[st];
                live[st] = f[st] < nwf;
            }
            const bool more = fn[0] < nwf;  // wave-uniform
            int claimed = 0;
            unsigned mword[S];  // bitmap words of the next frames (LDS reads in flight with the rest)
            // Prefetch into the (by then dead) x registers - straight from global, or from the
            // landing buffers (their DMA was issued a whole FFT ago) -, claim the frames after
            // these and fetch the time-band flags of the next ones.
            auto prefetch = [&]() {
                if (more) {
                    if constexpr (DIRECT) {
                        issue_dma(fn, b, t0, nwf);
                    } else {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        read_frames();
                    }
                    if (lane == 0) claimed = atomicAdd(next_frame, S);
                }
#pragma unroll
                for (int st = 0; st < S; ++st) mword[st] = 0;
                if constexpr (BANDS) {
                    if (tb && more) {
#pragma unroll
                        for (int st = 0; st < S; ++st)
                            mword[st] = tbits[min((a.C == 1) ? fn[st] : fn[st] / a.C, nt - 1) >> 5];
                    }
                }
            };
            // The frame reads, the claim and the flags have returned: rotate the frame cursors.
            auto advance = [&]() {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int st = 0; st < S; ++st) {
                    if constexpr (BANDS) {
                        const int tl = min((a.C == 1) ? fn[st] : fn[st] / a.C, nt - 1);
                        mbit[st] = (__builtin_amdgcn_readfirstlane(mword[st]) >> (tl & 31)) & 1u;
                    }
                    f[st] = fn[st];
                }
                if (more) {
                    claimed = __builtin_amdgcn_readfirstlane(claimed);
#pragma unroll
                    for (int st = 0; st < S; ++st) fn[st] = claimed + st;
                    if constexpr (!DIRECT) issue_dma(fn, b, t0, nwf);
                }
            };
            bool masked[S];
#pragma unroll
            for (int st = 0; st < S; ++st) masked[st] = false;
            if constexpr (BANDS) {
                bool all_masked = true;
#pragma unroll
                for (int st = 0; st < S; ++st) {
                    masked[st] = live[st] && mbit[st];
                    all_masked = all_masked && (masked[st] || !live[st]);
                }
                if (all_masked) {  // wave-uniform: nothing to transform, the frames are all-zero columns
                    prefetch();
#pragma unroll
                    for (int st = 0; st < S; ++st)
                        if (live[st])
                            for (int m = lane; m < a.M; m += kWave) store_band(fcur[st], (unsigned)m * rowpitch_b, 0.f);
                    mn = fminf(mn, 0.f);
                    mx = fmaxf(mx, 0.f);
                    advance();
                    continue;
                }
            }
            if constexpr (DIRECT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the frames (and older stores)
            PH_MARK(0);
#pragma unroll
            for (int st = 0; st < S; ++st)
#pragma unroll
                for (int q = 0; q < P; ++q) x[st][q] *= win[q];
            if (!ABL(1)) fft_frames<LOG2N, S>(x, tw, lds, lane);
            PH_MARK(3);
            // |X| scaled by 2 (the 0.5 of the untangle lives in the band weights)
            untangle_mag<LOG2N, HI, S>(x, post, lds, magbuf, lane);
            wave_sync_lds();
            PH_MARK(4);
            prefetch();
            if constexpr (BANDS && MELMODE == 2) {  // the global table is shared: zero the magnitudes instead
                if (fb) {
#pragma unroll
                    for (int st = 0; st < S; ++st)
                        for (int i = 0; i < a.n_fb; ++i) {
                            const int off = fb[2 * i], end = min(off + fb[2 * i + 1], NC + 1);
                            for (int k = off + lane; k < end; k += kWave) magbuf[st][k] = 0.f;
                        }
                    wave_sync_lds();
                }
            }
#pragma unroll
            for (int st = 0; st < S; ++st) {
                const float keep = masked[st] ? 0.f : scale;
                if constexpr (MELMODE == 0) {
                    const float4* mag4 = reinterpret_cast<const float4*>(magbuf[st] + lo0);  // lo0 % 4 == 0
                    // packed FMAs on two independent accumulators (a dependent packed op costs
                    // a wait state)
                    cf acc2 = mk(0.f, 0.f), acc3 = mk(0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < kMelRegs / 4; ++i) {
                        const float4 m4 = mag4[i];
                        acc2 = __builtin_elementwise_fma(mk(wreg[4 * i + 0], wreg[4 * i + 1]), mk(m4.x, m4.y), acc2);
                        acc3 = __builtin_elementwise_fma(mk(wreg[4 * i + 2], wreg[4 * i + 3]), mk(m4.z, m4.w), acc3);
                    }
                    acc2 += acc3;
                    const float acc = acc2.x + acc2.y;
                    if (live[st] && lane < a.M) {
                        const float v = acc * keep;
                        store_band(fcur[st], __umul24((unsigned)lane, rowpitch_b), v);  // host checks rowpitch < 2^24
                        mn = fminf(mn, v);
                        mx = fmaxf(mx, v);
                    }
                } else {
                    if (live[st]) {
                        for (int m = lane; m < a.M; m += kWave) {
                            float acc = 0.f;
                            if constexpr (MELMODE == 1) {
                                const int lo = lotab[m];
                                for (int i = 0; i < a.rows; ++i) acc = fmaf(wtab[i * a.M + m], magbuf[st][lo + i], acc);
                            } else {
                                const int lo = a.band_lo[m];
                                for (int i = 0; i < a.rows; ++i)
                                    acc = fmaf(a.wband[i * a.M + m], magbuf[st][lo + i], acc);
                            }
                            const float v = acc * keep;
                            store_band(fcur[st], (unsigned)m * rowpitch_b, v);
                            mn = fminf(mn, v);
                            mx = fmaxf(mx, v);
                        }
                    }
                }
            }
            wave_sync_lds();
            PH_MARK(5);
            advance();
            PH_MARK(2);
            if ABL(4096) ph[7] += 1;
        }
        PH_BEGIN();
        // every wave leaves its own (min, max) partial for k_minmax_log_apply
        mn = wave_min(mn);
        mx = wave_max(mx);
        if (lane == 0) {
            a.partial[((size_t)chunk * kFusedWaves + wv) * 2 + 0] = mn;
            a.partial[((size_t)chunk * kFusedWaves + wv) * 2 + 1] = mx;
        }
        if (chunk + (int)gridDim.x < a.n_chunks) {  // another chunk follows: restart the queue
            __syncthreads();
            if (threadIdx.x == 0) *next_frame = 2 * kFusedWaves * S;
            if constexpr (BANDS) {
                const int nc = chunk + (int)gridDim.x, nb = chunk_clip(nc);
                if (a.t_bands) build_tbits(a.t_bands + (size_t)nb * a.n_tb * 2, chunk_t0(nc, nb), chunk_nt(nc, nb));
                if constexpr (MELMODE == 1) {
                    if (a.f_bands) build_wtab(a.f_bands + (size_t)nb * a.n_fb * 2);
                }
            }
            __syncthreads();
        }
        PH_MARK(10);
    }
    if (ABL(4096) && lane == 0 && a.dbg && blockIdx.x < 4096) {
        PH_MARK(11);  // since the last mark: loop exit to kernel end
        for (int i = 0; i < 16; ++i) a.dbg[kDbgPhase0 + ((size_t)blockIdx.x * 16 + wv) * 16 + i] = ph[i];
    }
    if (ABL(512) && threadIdx.x == 0 && a.dbg) {
        if (blockIdx.x == 0) {
            a.dbg[0] = __builtin_amdgcn_s_memtime() - stamp0;
            a.dbg[1] = __builtin_amdgcn_s_memrealtime() - real0;
        }
        a.dbg[4 + 3 * blockIdx.x + 0] = real_entry;
        a.dbg[4 + 3 * blockIdx.x + 1] = real0;
        a.dbg[4 + 3 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    }
}

// ---------------------------------------------------------------------------
// K2: STFT only, reference layout [B, F, T, 2C]
// ---------------------------------------------------------------------------
struct StftArgs {
    const float* wav;
    float* spec;
    const float* consts;
    int B, C, L, T, hop, tile_frames, tiles_per_clip;
};

template <int LOG2N>
__global__ __launch_bounds__(256) void k_stft(const StftArgs a) {
    constexpr int N = 1 << LOG2N, NC = N / 2, P = FftCfg<LOG2N>::P, NTW = FftCfg<LOG2N>::NTW;
    constexpr int F = NC + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wg / a.tiles_per_clip, tile = wg - b * a.tiles_per_clip;
    const int t0 = tile * a.tile_frames;
    const int nt = min(a.tile_frames, a.T - t0);
    const int C2 = 2 * a.C;
    const int row = a.tile_frames * C2 + 1;  // odd stride: conflict-free column writes

    constexpr int kWaveBufBytes = (lds_padded(NC, FftCfg<LOG2N>::PMMAX) * 8 + 15) & ~15;
    cf* lds = reinterpret_cast<cf*>(smem + wv * kWaveBufBytes);
    float* tile_out = reinterpret_cast<float*>(smem + 4 * kWaveBufBytes);  // [F][row]

    cf tw[NTW], post[P / 2], win[P];
    float wreg_unused[kMelRegs];
    int lo_unused;
    load_consts<LOG2N>(a.consts, lane, tw, post, win, wreg_unused, lo_unused);

    const int nwf = nt * a.C;
    for (int f = wv; f < nwf; f += 4) {
        const int tl = f / a.C, c = f - tl * a.C;
        const float* clip = a.wav + ((size_t)b * a.C + c) * a.L;
        cf x[P];
        load_frame<LOG2N>(x, clip, a.L, (t0 + tl) * a.hop - N / 2, lane);
#pragma unroll
        for (int q = 0; q < P; ++q) x[q] *= win[q];
        fft_frame<LOG2N>(x, tw, lds, lane);
        cf xlo[P / 2], xhi[P / 2];
        untangle<LOG2N, true, true>(x, post, lds, lane, xlo, xhi);
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

    float mn = 0.f, den = 1.f;
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
        den = fmaxf(hi - lo, eps_div);
    }
    if (vec) {
        if (iv < end) {
            float* e = reinterpret_cast<float*>(&v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float y = e[j];
                if (do_minmax) y = (y - mn) / den;
                if (do_log) y = logf(y + eps_log);
                e[j] = y;
            }
            *reinterpret_cast<float4*>(p + iv) = v;
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
static int const_nv4(int log2n) {
    switch (log2n) {
        case 11: return ConstLayout<11>::NV4;
        case 10: return ConstLayout<10>::NV4;
        case 9: return ConstLayout<9>::NV4;
        default: return ConstLayout<8>::NV4;
    }
}
static size_t wave_buf_bytes(int log2n) {
    const int NC = (1 << log2n) / 2;
    switch (log2n) {
        case 11: return (size_t)lds_padded(NC, FftCfg<11>::PMMAX) * 8;
        case 10: return (size_t)lds_padded(NC, FftCfg<10>::PMMAX) * 8;
        case 9: return (size_t)lds_padded(NC, FftCfg<9>::PMMAX) * 8;
        default: return (size_t)lds_padded(NC, FftCfg<8>::PMMAX) * 8;
    }
}

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

static int plan_streams(const iris_plan* p) {
    return (p->streams == 2 && (p->log2n == 9 || p->log2n == 10)) ? 2 : 1;
}

// LDS of the fused kernel: landing + exchange buffers of every wave (the constant block is
// staged through the exchange area once), the frame queue, the MELMODE 1 tables
static size_t fused_lds_bytes(const iris_plan* p, int streams, int chunk_frames = 0) {
    const size_t xbuf = (wave_buf_bytes(p->log2n) + 15) & ~(size_t)15;
    const size_t waves = (size_t)fused_waves(p->log2n, streams) * streams;
    size_t bytes = waves * (size_t)p->n_fft * 4 + std::max(waves * xbuf, (size_t)const_nv4(p->log2n) * 64 * 16) + 16;
    if (p->mel_mode == 1) bytes += ((size_t)p->rows * p->n_mel + p->n_mel) * 4;
    // time-band bitmap of a chunk, written 2 words per wave per pass over the chunk's frames
    const size_t pass = (size_t)fused_waves(p->log2n, streams) * 64;
    bytes += (((size_t)chunk_frames + pass - 1) / pass * pass / 32 + 2) * 4;
    return bytes;
}

typedef void (*fused_kernel_t)(const FusedArgs);

template <int LOG2N, int MELMODE, int S>
static fused_kernel_t fused_kernel_hb(bool hi, bool bands) {
    if (hi)
        return bands ? k_wav_to_mel<LOG2N, MELMODE, true, true, S> : k_wav_to_mel<LOG2N, MELMODE, true, false, S>;
    return bands ? k_wav_to_mel<LOG2N, MELMODE, false, true, S> : k_wav_to_mel<LOG2N, MELMODE, false, false, S>;
}
template <int LOG2N, int S>
static fused_kernel_t fused_kernel_mm(int mel_mode, bool hi, bool bands) {
    if constexpr (LOG2N <= 10) {
        if (mel_mode == 0)  // register weights exist only for the half-spectrum variant up to n_fft 1024
            return bands ? k_wav_to_mel<LOG2N, 0, false, true, S> : k_wav_to_mel<LOG2N, 0, false, false, S>;
    }
    if (mel_mode == 1) return fused_kernel_hb<LOG2N, 1, S>(hi, bands);
    return fused_kernel_hb<LOG2N, 2, S>(hi, bands);
}
// two frame streams per wave exist for n_fft 512 / 1024
template <int LOG2N>
static fused_kernel_t fused_kernel_m(int mel_mode, bool hi, bool bands, int streams) {
    if constexpr (LOG2N == 9 || LOG2N == 10) {
        if (streams == 2) return fused_kernel_mm<LOG2N, 2>(mel_mode, hi, bands);
    }
    return fused_kernel_mm<LOG2N, 1>(mel_mode, hi, bands);
}
static fused_kernel_t fused_kernel(int log2n, int mel_mode, bool hi, bool bands, int streams) {
    switch (log2n) {
        case 11: return fused_kernel_m<11>(mel_mode, hi, bands, streams);
        case 10: return fused_kernel_m<10>(mel_mode, hi, bands, streams);
        case 9: return fused_kernel_m<9>(mel_mode, hi, bands, streams);
        default: return fused_kernel_m<8>(mel_mode, hi, bands, streams);
    }
}
static const void* stft_kernel(int log2n) {
    switch (log2n) {
        case 11: return (const void*)k_stft<11>;
        case 10: return (const void*)k_stft<10>;
        case 9: return (const void*)k_stft<9>;
        default: return (const void*)k_stft<8>;
    }
}

// Dynamic LDS above the 64 KiB default must be opted into once per kernel.
static hipError_t allow_big_lds(const iris_plan* p) {
    constexpr int kMaxLds = 160 * 1024;
    hipError_t e;
    for (int v = 0; v < 4; ++v) {
        const int streams = (v & 2) ? 2 : 1;
        if (streams == 2 && p->log2n != 9 && p->log2n != 10) continue;
        e = hipFuncSetAttribute((const void*)fused_kernel(p->log2n, p->mel_mode, p->need_hi != 0, (v & 1) != 0, streams),
                                hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
        if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute(stft_kernel(p->log2n), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
}

extern "C" int iris_abi_version(void) { return IRIS_ABI_VERSION; }
extern "C" const char* iris_last_error(void) { return g_err; }

extern "C" int iris_plan_create(iris_plan** out, int device, int n_fft, int hop, int n_mel, int n_bins,
                                float sample_rate, float lower_hz, float upper_hz, int channels, int max_batch,
                                int max_len, const float* mel_host) {
    if (!out) return fail(IRIS_E_INVALID, "iris_plan_create: out is NULL");
    *out = nullptr;
    // n_fft == 0: mel-only plan (iris_magmel on any n_bins >= 2; no FFT entry points)
    const bool mel_only = (n_fft == 0);
    int log2n = mel_only ? 8 : ilog2_exact(n_fft);
    if (!mel_only && (log2n < 8 || log2n > 11))
        return fail(IRIS_E_UNSUPPORTED, "n_fft=%d: must be a power of two in [256, 2048] (or 0 for a mel-only plan)",
                    n_fft);
    if (n_mel <= 0) return fail(IRIS_E_INVALID, "n_mel=%d must be positive", n_mel);
    if (channels <= 0 || max_batch <= 0) return fail(IRIS_E_INVALID, "channels and max_batch must be positive");
    if (mel_only) {
        if (n_bins < 2) return fail(IRIS_E_INVALID, "n_bins=%d must be >= 2", n_bins);
        hop = 1;
        max_len = std::max(max_len, 1);
    } else {
        if (hop <= 0) return fail(IRIS_E_INVALID, "hop=%d must be positive", hop);
        if (n_bins != n_fft / 2 + 1)
            return fail(IRIS_E_INVALID, "n_bins=%d must equal n_fft/2+1=%d", n_bins, n_fft / 2 + 1);
        if (max_len <= n_fft / 2)
            return fail(IRIS_E_INVALID, "max_len=%d must exceed n_fft/2 (reflect padding)", max_len);
    }

    iris_plan* p = new (std::nothrow) iris_plan();
    if (!p) return fail(IRIS_E_NOMEM, "out of host memory");
    p->device = device;
    p->n_fft = n_fft;
    p->mel_only = mel_only;
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
    p->d_consts = nullptr;
    p->d_band_lo = p->d_band_len = p->d_fband_lo = nullptr;
    p->d_bin_band = nullptr;
    p->d_bin_w = nullptr;
    p->d_wband = p->d_mel = p->d_ws = nullptr;
    p->d_dbg = nullptr;
    p->streams = 1;
    if (const char* e = getenv("IRIS_STREAMS")) p->streams = atoi(e) == 2 ? 2 : 1;
    p->timing = 0;
    p->launch_no = 0;
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
    // streaming magmel tables: valid when every bin's non-zeros sit in <= 2 adjacent bands and
    // the first band index never decreases with the bin (true for triangular filterbanks)
    std::vector<int> bin_band(n_bins, -1);
    std::vector<float> bin_w((size_t)n_bins * 2, 0.f);
    p->tri_ok = 1;
    p->tri_f_lo = n_bins;
    p->tri_f_hi = 0;
    {
        int prev = -1;
        for (int f = 0; f < n_bins && p->tri_ok; ++f) {
            int first = -1, last = -1;
            for (int m = 0; m < n_mel; ++m)
                if (p->mel[(size_t)f * n_mel + m] != 0.f) {
                    if (first < 0) first = m;
                    last = m;
                }
            if (first < 0) continue;
            if (last - first > 1 || first < prev) {
                p->tri_ok = 0;
                break;
            }
            prev = first;
            bin_band[f] = first;
            bin_w[2 * (size_t)f] = p->mel[(size_t)f * n_mel + first];
            bin_w[2 * (size_t)f + 1] = last > first ? p->mel[(size_t)f * n_mel + last] : 0.f;
            p->tri_f_lo = std::min(p->tri_f_lo, f);
            p->tri_f_hi = std::max(p->tri_f_hi, f + 1);
        }
    }
    // fused kernel tables: which half of the spectrum it must produce, and per band a
    // window of `rows` bins [flo, flo + rows) inside the bins the kernel writes
    const int NC = mel_only ? 2 * (n_bins - 1) / 2 : n_fft / 2;
    p->need_hi = p->k_need > NC / 2 ? 1 : 0;
    // bins the kernel writes to its magnitude buffer: [0, limit)
    const int limit = p->need_hi ? ((n_bins + 3) & ~3) : NC / 2;
    // (n_fft 2048 keeps 16 points per lane: no registers to spare for the weights -> table modes)
    // (the full-spectrum untangle needs the registers too: need_hi -> table modes)
    if (log2n <= 10 && !p->need_hi && n_mel <= 64 && p->max_band_len + 3 <= kMelRegs && limit >= kMelRegs) {
        p->mel_mode = 0;  // 16-byte aligned register window of kMelRegs bins per band
        p->rows = kMelRegs;
    } else {
        p->rows = std::min(std::max(p->max_band_len, 1), limit);
        p->mel_mode = ((size_t)p->rows * n_mel + n_mel) * 4 <= 32 * 1024 ? 1 : 2;
    }
    std::vector<int> flo(n_mel, 0);
    std::vector<float> wband((size_t)p->rows * n_mel, 0.f);
    for (int m = 0; m < n_mel; ++m) {
        int first = p->mel_mode == 0 ? (lo[m] & ~3) : lo[m];
        flo[m] = std::max(0, std::min(first, limit - p->rows));
        for (int i = 0; i < p->rows; ++i) {
            const int f = flo[m] + i;
            wband[(size_t)i * n_mel + m] = f < n_bins ? 0.5f * p->mel[(size_t)f * n_mel + m] : 0.f;
        }
    }

    DeviceGuard guard(device);
    if (!guard.ok) {
        delete p;
        return fail(IRIS_E_INVALID, "cannot select HIP device %d", device);
    }
    std::vector<float2> tw, post, win;
    build_tables(log2n, tw, post, win);
    // pack the per-lane constant block (ConstLayout)
    const int ntw = fft_ntw(log2n), P = fft_p(log2n);
    const int off_post = 2 * ntw, off_win = off_post + P, off_wreg = off_win + 2 * P, off_lo = off_wreg + kMelRegs;
    const int nv4 = (off_lo + 1 + 3) / 4;
    std::vector<float> consts((size_t)nv4 * 64 * 4, 0.f);
    auto put = [&](int lane, int idx, float v) { consts[((size_t)(idx / 4) * 64 + lane) * 4 + (idx % 4)] = v; };
    for (int lane = 0; lane < 64; ++lane) {
        for (int i = 0; i < ntw; ++i) {
            put(lane, 2 * i, tw[(size_t)i * 64 + lane].x);
            put(lane, 2 * i + 1, tw[(size_t)i * 64 + lane].y);
        }
        for (int i = 0; i < P / 2; ++i) {
            put(lane, off_post + 2 * i, post[(size_t)i * 64 + lane].x);
            put(lane, off_post + 2 * i + 1, post[(size_t)i * 64 + lane].y);
        }
        for (int i = 0; i < P; ++i) {
            put(lane, off_win + 2 * i, win[(size_t)i * 64 + lane].x);
            put(lane, off_win + 2 * i + 1, win[(size_t)i * 64 + lane].y);
        }
        if (p->mel_mode == 0 && lane < n_mel) {
            for (int i = 0; i < p->rows; ++i) put(lane, off_wreg + i, wband[(size_t)i * n_mel + lane]);
            float bits;
            memcpy(&bits, &flo[lane], sizeof(float));
            put(lane, off_lo, bits);
        }
    }
    int rc;
    if ((rc = upload(&p->d_consts, consts)) ||
        (rc = upload(&p->d_band_lo, lo)) || (rc = upload(&p->d_band_len, len)) ||
        (rc = upload(&p->d_fband_lo, flo)) || (rc = upload(&p->d_wband, wband)) ||
        (rc = upload(&p->d_bin_band, bin_band)) || (rc = upload(&p->d_bin_w, bin_w)) ||
        (rc = upload(&p->d_mel, p->mel))) {
        iris_plan_destroy(p);
        return rc;
    }

    if (!mel_only) {
        hipError_t e = allow_big_lds(p);
        if (e != hipSuccess) {
            iris_plan_destroy(p);
            return fail((int)e, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
        }
    }
    {
        int cu = 0;
        hipError_t e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device);
        if (e != hipSuccess || cu <= 0) cu = 256;
        p->num_cu = cu;
    }
    p->chunk_target = 0;
    if (const char* e = getenv("IRIS_CHUNK_FRAMES")) p->chunk_target = std::max(0, atoi(e));
    if (!mel_only && fused_lds_bytes(p, 1) > 160 * 1024) {
        iris_plan_destroy(p);
        return fail(IRIS_E_UNSUPPORTED, "n_mel=%d: the band table does not fit the LDS", n_mel);
    }

    // workspace of the fused path: [B, tiles, 2] min/max partials (worst case one
    // frame per tile) + [B, chunks] sums of squares for IRIS_F_NORMALIZE
    const int t_max = 1 + max_len / hop;
    const size_t wav_row = (size_t)channels * max_len;
    p->ws_floats = 2 * 16 * (size_t)max_batch * t_max + (size_t)max_batch * ((wav_row + kChunk - 1) / kChunk) + 64;
    (void)hipMalloc((void**)&p->d_dbg, kDbgWords * sizeof(unsigned long long));
    if (p->d_dbg) (void)hipMemset(p->d_dbg, 0, kDbgWords * sizeof(unsigned long long));
    hipError_t e;
    e = hipMalloc((void**)&p->d_ws, p->ws_floats * sizeof(float));
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
    if (p->d_dbg && getenv("IRIS_ABLATE") && (atoi(getenv("IRIS_ABLATE")) & 4096)) {
        std::vector<unsigned long long> h(kDbgWords, 0);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), p->d_dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int waves = 0;
        for (int w = 0; w < 4096 * 16; ++w) {
            const unsigned long long* r = &h[kDbgPhase0 + (size_t)w * 16];
            if (!r[7]) continue;
            ++waves;
            for (int i = 0; i < 16; ++i) sum[i] += (double)r[i];
        }
        if (waves && sum[7] > 0)
            fprintf(stderr, "[iris dbg] %d waves, %.2f frames each; cycles per frame: dma-wait %.0f, frame-read %.0f, "
                    "claim+dma-issue %.0f, window+fft %.0f, untangle+mag %.0f, mel %.0f; per wave: chunk setup %.0f, chunk barrier %.0f, tile write-out %.0f, "
                    "block min/max %.0f, exit %.0f; write-out parts: setup %.0f, lds issue %.0f, lds wait %.0f\n",
                    waves, sum[7] / waves, sum[0] / sum[7], sum[1] / sum[7], sum[2] / sum[7], sum[3] / sum[7],
                    sum[4] / sum[7], sum[5] / sum[7], sum[8] / waves, sum[6] / waves, sum[9] / waves,
                    sum[10] / waves, sum[11] / waves, sum[12] / waves, sum[13] / waves, sum[14] / waves);
        if (sum[15] > 0) fprintf(stderr, "[iris dbg] dummy LDS read after the barrier: %.0f cycles\n", sum[15] / waves);
        if (sum[1] > 0 && (atoi(getenv("IRIS_ABLATE")) & 16384)) fprintf(stderr, "[iris dbg] whole chunk, cold pass %.0f cycles, warm pass %.0f cycles\n", sum[0] / waves, sum[1] / waves);

    }
    if (p->d_dbg && getenv("IRIS_ABLATE") && (atoi(getenv("IRIS_ABLATE")) & 512)) {
        std::vector<unsigned long long> h(4 + 3 * 4096, 0);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), p->d_dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        fprintf(stderr, "[iris dbg] workgroup 0: %llu shader cycles, %llu x 10 ns -> %.3f GHz\n", h[0], h[1],
                h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0);
        unsigned long long e0 = ~0ull, e1 = 0, l0 = ~0ull, l1 = 0, x0 = ~0ull, x1 = 0;
        double pro = 0, loop = 0;
        int n = 0;
        for (int i = 0; i < 4096; ++i) {
            const unsigned long long* r = &h[4 + 3 * i];
            if (!r[0]) continue;
            ++n;
            e0 = std::min(e0, r[0]); e1 = std::max(e1, r[0]);
            l0 = std::min(l0, r[1]); l1 = std::max(l1, r[1]);
            x0 = std::min(x0, r[2]); x1 = std::max(x1, r[2]);
            pro += (double)(r[1] - r[0]); loop += (double)(r[2] - r[1]);
        }
        if (const char* path = getenv("IRIS_DBG_DUMP")) {
            if (FILE* fp = fopen(path, "w")) {
                for (int i = 0; i < 4096; ++i) {
                    const unsigned long long* r = &h[4 + 3 * i];
                    if (r[0]) fprintf(fp, "%d %llu %llu %llu\n", i, r[0] - e0, r[1] - e0, r[2] - e0);
                }
                fclose(fp);
            }
        }
        if (n)
            fprintf(stderr, "[iris dbg] %d workgroups (last launch): entry spread %.2f us, loop-start spread %.2f us, "
                    "exit spread %.2f us, first entry -> last exit %.2f us, mean prologue %.2f us, mean loop %.2f us\n",
                    n, (e1 - e0) * 0.01, (l1 - l0) * 0.01, (x1 - x0) * 0.01, (x1 - e0) * 0.01, pro / n * 0.01,
                    loop / n * 0.01);
    }
    (void)hipFree(p->d_dbg);
    for (hipEvent_t ev : p->ev) (void)hipEventDestroy(ev);
    (void)hipFree(p->d_consts);
    (void)hipFree(p->d_band_lo);
    (void)hipFree(p->d_band_len);
    (void)hipFree(p->d_fband_lo);
    (void)hipFree(p->d_bin_band);
    (void)hipFree(p->d_bin_w);
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
    if (p->mel_only) return fail(IRIS_E_UNSUPPORTED, "%s: plan was created mel-only (n_fft = 0)", who);
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
    a.consts = p->d_consts;
    a.B = batch;
    a.C = p->channels;
    a.L = len;
    a.T = 1 + len / p->hop;
    a.hop = p->hop;
    const int NC = p->n_fft / 2, F = NC + 1;
    int tf = 16;
    auto lds_of = [&](int t) { return 4 * ((wave_buf_bytes(p->log2n) + 15) & ~(size_t)15) + (size_t)F * (t * 2 * p->channels + 1) * 4; };
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
    const bool aligned = (reinterpret_cast<uintptr_t>(spec) & (8 * p->channels - 1)) == 0;
    if (p->tri_ok && (p->channels == 1 || p->channels == 2) && aligned &&
        (size_t)p->n_mel * 64 * p->channels * sizeof(float) <= 64 * 1024 && getenv("IRIS_MAGMEL_GENERIC") == nullptr) {
        MagmelTriArgs t;
        t.spec = spec;
        t.mel = mel;
        t.bin_band = p->d_bin_band;
        t.bin_w = p->d_bin_w;
        t.t_bands = a.t_bands;
        t.n_tb = n_tb;
        t.f_bands = a.f_bands;
        t.n_fb = n_fb;
        t.B = batch;
        t.F = p->n_bins;
        t.T = n_frames;
        t.M = p->n_mel;
        t.is_magphase = is_magphase;
        t.f_lo = p->tri_f_lo;
        t.f_hi = p->tri_f_hi;
        const dim3 grid((n_frames + 63) / 64, batch);
        // split the bins over 8 waves when the grid alone cannot fill the chip
        const int threads = (size_t)grid.x * grid.y * 4 < (size_t)p->num_cu * 8 ? 512 : 256;
        const size_t lds = (size_t)p->n_mel * 64 * p->channels * sizeof(float);
        if (p->channels == 1) k_magmel_tri<1><<<grid, threads, lds, (hipStream_t)stream>>>(t);
        else k_magmel_tri<2><<<grid, threads, lds, (hipStream_t)stream>>>(t);
    } else {
        const int tc = n_frames * p->channels;
        k_magmel<<<dim3((tc + 63) / 64, batch), 256, 0, (hipStream_t)stream>>>(a);
    }
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
        if (reinterpret_cast<uintptr_t>(workspace) & 7)
            return fail(IRIS_E_INVALID, "iris_minmax_log: workspace must be 8-byte aligned");
        if (!workspace || workspace_floats < iris_minmax_log_workspace(n_rows, row_len))
            return fail(IRIS_E_CAPACITY, "iris_minmax_log: workspace %zu floats < %zu", workspace_floats,
                        iris_minmax_log_workspace(n_rows, row_len));
        k_minmax_partial<<<dim3((unsigned)n_part, n_rows), 256, 0, s>>>(x, workspace, row_len, (int)n_part);
    }
    k_minmax_log_apply<<<dim3((unsigned)((row_len + kApply - 1) / kApply), n_rows), 256, 0, s>>>(x, workspace, (int)n_part, row_len, do_minmax,
                                                                     do_log, eps_div, eps_log);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

constexpr int kMaxChunkFrames = 16384;
// Chunk geometry of the fused kernel for `per_cu` workgroups per CU: every workgroup one
// chunk when the problem is large enough, chunks never span clips.
static void fused_geometry(const iris_plan* p, int batch, int T, int per_cu, int* chunk_frames, int* chunks_per_clip) {
    const int slots = p->num_cu * per_cu;
    const long total = (long)batch * T;
    int target = p->chunk_target > 0 ? p->chunk_target : (int)((total + slots - 1) / slots);
    target = std::max(target, std::min(8, T));
    int cpc = (T + target - 1) / target;
    // rounding up per clip can overshoot the slots by a few chunks, which would cost a whole
    // second round: prefer slightly larger chunks that fit one round
    if (p->chunk_target == 0 && (long)batch * cpc > slots && batch <= slots) cpc = std::max(1, slots / batch);
    cpc = std::max(cpc, (T + kMaxChunkFrames - 1) / kMaxChunkFrames);  // bounds the time-band bitmap in LDS
    *chunks_per_clip = cpc;
    *chunk_frames = (T + cpc - 1) / cpc;
}

// Geometry + grid for the residency the hardware really grants (registers and LDS): start from
// the register-limited occupancy and go down until the occupancy query agrees.
static int fused_config(const iris_plan* p, fused_kernel_t kernel, int batch, int T, int streams, int* chunk_frames,
                        int* chunks_per_clip, int* grid, size_t* lds) {
    for (int per_cu = fused_occ(p->log2n); per_cu >= 1; --per_cu) {
        fused_geometry(p, batch, T, per_cu, chunk_frames, chunks_per_clip);
        *lds = fused_lds_bytes(p, streams, *chunk_frames);
        if (*lds > 160 * 1024) return fail(IRIS_E_UNSUPPORTED, "fused kernel needs %zu B of LDS", *lds);
        int resident = 0;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void*)kernel,
                                                                    64 * fused_waves(p->log2n, streams), *lds);
        if (e != hipSuccess) return fail((int)e, "hipOccupancyMaxActiveBlocksPerMultiprocessor: %s",
                                         hipGetErrorString(e));
        if (resident >= per_cu) {
            *grid = std::min(batch * *chunks_per_clip, p->num_cu * per_cu);
            return IRIS_OK;
        }
    }
    return fail(IRIS_E_UNSUPPORTED, "fused kernel does not fit one workgroup per CU (LDS %zu B)", *lds);
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
    a.consts = p->d_consts;
    a.band_lo = p->d_fband_lo;
    a.wband = p->d_wband;
    a.rows = p->rows;
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
    const int do_minmax = (flags & IRIS_F_MINMAX) ? 1 : 0, do_log = (flags & IRIS_F_LOG) ? 1 : 0;
    a.ablate = 0;
    if (const char* e = getenv("IRIS_ABLATE")) a.ablate = atoi(e);
    a.dbg = p->d_dbg;
    const bool bands = (n_tb > 0) || (n_fb > 0);
    const int streams = plan_streams(p);
    const fused_kernel_t kernel = fused_kernel(p->log2n, p->mel_mode, p->need_hi != 0, bands, streams);
    int grid = 0;
    size_t lds = 0;
    if ((rc = fused_config(p, kernel, batch, a.T, streams, &a.chunk_frames, &a.chunks_per_clip, &grid, &lds)))
        return rc;
    if ((size_t)p->n_mel * a.T * p->channels * 4 > 0xffffffffull || (size_t)a.T * p->channels * 4 >= (1u << 24))
        return fail(IRIS_E_UNSUPPORTED, "iris_wav_to_logmel: clip too long (%d frames x %d channels)", a.T, p->channels);
    a.n_chunks = batch * a.chunks_per_clip;
    a.chunk_base = a.T / a.chunks_per_clip;
    a.chunk_rem = a.T % a.chunks_per_clip;
    const int waves = fused_waves(p->log2n, streams);
    const int parts_per_chunk = waves;
    const size_t n_partial = 2 * (size_t)a.n_chunks * parts_per_chunk;
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

    // bench hook: the kernel's own start/stop timestamps are attached to an event pair by the
    // AMD launch extension (no extra packets on the stream, unlike hipEventRecord brackets)
    const bool timed = p->timing > 0 && (p->launch_no++ % p->timing) == 0 && p->ev_used < kMaxTimedLaunches;
    hipError_t e;
    if (timed) {
        while ((int)p->ev.size() < 2 * (p->ev_used + 1)) {
            hipEvent_t ev;
            HIP_TRY(hipEventCreate(&ev));
            p->ev.push_back(ev);
        }
        FusedArgs args = a;
        void* kargs[] = {&args};
        e = hipExtLaunchKernel((const void*)kernel, dim3(grid), dim3(64 * waves), kargs, lds, s,
                               p->ev[2 * p->ev_used], p->ev[2 * p->ev_used + 1], 0);
        if (e == hipSuccess) p->ev_used++;
    } else {
        kernel<<<grid, 64 * waves, lds, s>>>(a);
        e = hipGetLastError();
    }
    HIP_TRY(e);
    if (do_minmax || do_log) {
        const size_t row_len = (size_t)p->n_mel * a.T * p->channels;
        const unsigned n_chunks = (unsigned)((row_len + kApply - 1) / kApply);
        k_minmax_log_apply<<<dim3(n_chunks, batch), 256, 0, s>>>(out, p->d_ws, a.chunks_per_clip * parts_per_chunk, row_len, do_minmax,
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

extern "C" int iris_agc_clip(const iris_agc_row* rows_dev, size_t n_rows, float clip_factor, float eps,
                             float clipvalue, void* stream) {
    if (!rows_dev) return fail(IRIS_E_INVALID, "iris_agc_clip: rows is NULL");
    if (n_rows == 0) return IRIS_OK;
    const int grid = (int)std::min<size_t>((n_rows + 3) / 4, 4096);
    k_agc_clip<<<grid, 256, 0, (hipStream_t)stream>>>(rows_dev, n_rows, clip_factor, eps, clipvalue);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// ---------------------------------------------------------------------------
// batched sample synthesis (merge_complex_specs, pipeline.py:6-110)
// ---------------------------------------------------------------------------
// frame t of the output -> frame of the (virtually zero-padded) source, or -1 inside the padding
__device__ __forceinline__ int mix_frame(const iris_mix_src& s, int t) {
    const int fr = s.off + t - s.pad;
    return (fr >= 0 && fr < s.T) ? fr : -1;
}

// active[s][t] = 1 when max over (freq, chan2) of voice frame t is > 0 (pipeline.py:57); one thread
// per output frame walks the bins (loads coalesced along t)
__global__ __launch_bounds__(256) void k_mix_active(const iris_mix_src* srcs, int n_bins, int n_frame, int chan2,
                                                    float* active) {
    const iris_mix_src s = srcs[blockIdx.y];
    if (s.kind != 1) return;  // uniform
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_frame) return;
    const int fr = mix_frame(s, t);
    float mx = -INFINITY;
    if (fr >= 0) {
        const float* p = s.src + (size_t)fr * chan2;
        for (int f = 0; f < n_bins; ++f, p += (size_t)s.T * chan2)
            for (int c = 0; c < chan2; ++c) mx = fmaxf(mx, p[c]);
    }
    active[(size_t)blockIdx.y * n_frame + t] = (fr >= 0 && mx > 0.f) ? 1.f : 0.f;
}

// One block per sample: voices are accepted in slot order unless their labels would overlap the
// labels accepted so far (pipeline.py:72-84); writes the sample's label planes and one flag per voice.
__global__ __launch_bounds__(256) void k_mix_labels(const iris_mix_src* srcs, const int32_t* first,
                                                    const float* label_vecs, const float* active, float* flags,
                                                    float* labels, int n_frame, int max_voices, int n_classes) {
    extern __shared__ float lsum[];  // [n_frame][n_classes] labels accepted so far, summed over voices
    const int b = blockIdx.x, plane = n_frame * n_classes;
    float* lab = labels + (size_t)b * max_voices * plane;
    for (int i = threadIdx.x; i < max_voices * plane; i += blockDim.x) lab[i] = 0.f;
    for (int i = threadIdx.x; i < plane; i += blockDim.x) lsum[i] = 0.f;
    __syncthreads();
    for (int si = first[b]; si < first[b + 1]; ++si) {
        const iris_mix_src s = srcs[si];
        if (s.kind != 1) continue;  // uniform
        const float* lv = label_vecs + (size_t)s.label_row * n_classes;
        const float* act = active + (size_t)si * n_frame;
        int over = 0;
        for (int i = threadIdx.x; i < plane; i += blockDim.x) {
            const int t = i / n_classes, c = i - t * n_classes;
            over |= (lsum[i] + lv[c] * act[t]) >= 2.f;
        }
        over = __syncthreads_or(over);
        if (threadIdx.x == 0) flags[si] = over ? 0.f : 1.f;
        if (!over && s.slot >= 0 && s.slot < max_voices) {
            for (int i = threadIdx.x; i < plane; i += blockDim.x) {
                const int t = i / n_classes, c = i - t * n_classes;
                const float l = lv[c] * act[t];
                lsum[i] += l;
                lab[(size_t)s.slot * plane + i] = l;
            }
        }
        __syncthreads();
    }
}

// spec_out[b, f, t, :] = background + accepted voices + noises, added in table order with
// separately rounded multiply and add (no FMA contraction: equals the op-by-op reference)
template <int C2>
__global__ __launch_bounds__(256) void k_mix_sum(const iris_mix_src* srcs, const int32_t* first, const float* flags,
                                                 float* out, int n_bins, int n_frame) {
#pragma clang fp contract(off)  // gain * x is rounded before it is added, as in the op-by-op reference
    typedef float vecT __attribute__((ext_vector_type(C2)));
    const int b = blockIdx.z, f = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_frame) return;
    vecT acc = vecT(0.f);
    for (int si = first[b]; si < first[b + 1]; ++si) {
        const iris_mix_src s = srcs[si];  // uniform
        if (s.kind == 0) {
            const int fr = (s.off + t) % s.T;
            acc = *reinterpret_cast<const vecT*>(s.src + ((size_t)f * s.T + fr) * C2);
            continue;
        }
        const float keep = s.kind == 1 ? flags[si] : 1.f;
        const int fr = mix_frame(s, t);
        if (fr < 0 || keep == 0.f) continue;  // adds exactly zero
        const vecT v = *reinterpret_cast<const vecT*>(s.src + ((size_t)f * s.T + fr) * C2);
#pragma unroll
        for (int c = 0; c < C2; ++c) {
            const float scaled = s.gain * v[c];
            acc[c] = acc[c] + scaled;
        }
    }
    *reinterpret_cast<vecT*>(out + (((size_t)b * n_bins + f) * n_frame + t) * C2) = acc;
}

extern "C" size_t iris_mix_workspace(int n_srcs, int n_frame) {
    return (size_t)std::max(n_srcs, 0) * ((size_t)std::max(n_frame, 0) + 1);
}

extern "C" int iris_mix_specs(const iris_mix_src* srcs_dev, int n_srcs, const int32_t* first_dev,
                              const float* label_vecs_dev, float* spec_out, float* labels_out, int batch,
                              int n_bins, int n_frame, int chan2, int max_voices, int n_classes, float* workspace,
                              size_t workspace_floats, void* stream) {
    if (!srcs_dev || !first_dev || !label_vecs_dev || !spec_out || !labels_out || !workspace)
        return fail(IRIS_E_INVALID, "iris_mix_specs: NULL argument");
    if (batch <= 0 || n_srcs < batch || n_bins <= 0 || n_frame <= 0 || max_voices <= 0 || n_classes <= 0)
        return fail(IRIS_E_INVALID, "iris_mix_specs: bad sizes (batch %d, %d sources, %d bins, %d frames)", batch,
                    n_srcs, n_bins, n_frame);
    if (chan2 != 1 && chan2 != 2 && chan2 != 4 && chan2 != 8)
        return fail(IRIS_E_UNSUPPORTED, "iris_mix_specs: chan2 %d (1, 2, 4 or 8)", chan2);
    if (batch > 65535 || n_bins > 65535 || n_srcs > 65535)
        return fail(IRIS_E_UNSUPPORTED, "iris_mix_specs: batch / bins / sources above 65535");
    if (workspace_floats < iris_mix_workspace(n_srcs, n_frame))
        return fail(IRIS_E_CAPACITY, "iris_mix_specs: workspace %zu floats < %zu", workspace_floats,
                    iris_mix_workspace(n_srcs, n_frame));
    const size_t lds = (size_t)n_frame * n_classes * sizeof(float);
    if (lds > 64 * 1024) return fail(IRIS_E_UNSUPPORTED, "iris_mix_specs: n_frame x n_classes too large for the LDS");
    hipStream_t s = (hipStream_t)stream;
    float* active = workspace;                           // [n_srcs][n_frame]
    float* flags = workspace + (size_t)n_srcs * n_frame;  // [n_srcs]
    const unsigned tblocks = (unsigned)((n_frame + 255) / 256);
    k_mix_active<<<dim3(tblocks, n_srcs), 256, 0, s>>>(srcs_dev, n_bins, n_frame, chan2, active);
    k_mix_labels<<<batch, 256, lds, s>>>(srcs_dev, first_dev, label_vecs_dev, active, flags, labels_out, n_frame,
                                         max_voices, n_classes);
    const dim3 grid(tblocks, n_bins, batch);
    switch (chan2) {
        case 1: k_mix_sum<1><<<grid, 256, 0, s>>>(srcs_dev, first_dev, flags, spec_out, n_bins, n_frame); break;
        case 2: k_mix_sum<2><<<grid, 256, 0, s>>>(srcs_dev, first_dev, flags, spec_out, n_bins, n_frame); break;
        case 4: k_mix_sum<4><<<grid, 256, 0, s>>>(srcs_dev, first_dev, flags, spec_out, n_bins, n_frame); break;
        default: k_mix_sum<8><<<grid, 256, 0, s>>>(srcs_dev, first_dev, flags, spec_out, n_bins, n_frame); break;
    }
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_timing_enable(iris_plan* p, int enable) {
    if (!p) return fail(IRIS_E_INVALID, "iris_timing_enable: NULL plan");
    p->timing = enable > 0 ? enable : 0;
    p->launch_no = 0;
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
