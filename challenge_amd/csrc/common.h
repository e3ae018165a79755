// common.h -- includes, diagnostics switches, error plumbing, the plan, small device helpers.
// Part of the single translation unit iris_frontend.hip (included first).
#pragma once
#include "../../include/iris_frontend.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <type_traits>
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
// diagnostic buffer: [4] header, [kDbgWg * 4096] per-workgroup stamps (entry, loop start, exit, tile complete, clip
// range known, spare), [4096 * 16 * 16] per-wave phase cycles
static constexpr int kDbgWg = 6, kDbgPhase0 = 4 + kDbgWg * 4096, kDbgWords = kDbgPhase0 + 4096 * 16 * 16;
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
    // fp16-MFMA mel variant: availability, device tables, staged bins, selected precision (0 fp32, 1 fp16 MFMA)
    int mfma_ok, mfma_kb, mel_precision;
    void* d_wfrag;
    int* d_tile_ks;
    float* d_ws;  // workspace
    unsigned long long* d_slots;  // fused epilogue: [n_slots][2] {epoch, value} granules
    unsigned* h_status;           // status word in pinned, coherent HOST memory: the kernel raises it with a system-scope
    unsigned* d_status;           // store (through d_status, its device address), the host reads it without synchronising
    unsigned long long timeout_ticks;  // bound of the epilogue's waits, in s_memrealtime ticks (100 MHz)
    size_t n_slots;
    unsigned epoch;  // launches of the fused-epilogue kernel so far (granule tag; never 0)
    int epilogue;    // IRIS_EPILOGUE_*
    int last_form;   // form the last iris_wav_to_logmel call of this plan took: IRIS_EPILOGUE_* (-1: none yet / nothing to apply)
    unsigned long long* d_dbg;  // diagnostic stamps (IRIS_DIAG builds only; nullptr otherwise)
    int ablate;                 // IRIS_DIAG builds: IRIS_ABLATE bits, read once at plan creation
    bool magmel_generic;        // IRIS_MAGMEL_GENERIC set at plan creation: iris_magmel takes the generic kernel
    int streams;                // IRIS_STREAMS: frames in flight per wave (1 or 2)
    size_t ws_floats;
    int num_cu;
    int chunk_target;  // 0 = auto; frames per chunk of the fused kernel (IRIS_CHUNK_FRAMES)
    // timing
    int timing;        // 0 off, n: every n-th launch carries an event pair
    long launch_no;    // launches since timing was enabled
    std::vector<hipEvent_t> ev;  // pairs around the fused kernel
    int ev_used;
    std::vector<hipEvent_t> ev2;  // pairs around the min-max / log kernel of the same calls
    int ev2_used;
    // launch geometry of the fused kernel per (kernel, batch, frames, chunk bitmap): the occupancy query runs once per
    // shape, never on the hot launch path
    struct FusedGeom {
        const void* kernel;
        int batch, T;
        int chunk_frames, chunks_per_clip, grid;
        size_t lds;
    };
    std::vector<FusedGeom> geom_cache;
};

constexpr int kChunk = 4096;        // elements per partial-reduction block
constexpr int kMaxTimedLaunches = 4096;
constexpr int kTimingSkip = 4;  // launches after iris_timing_enable that are never sampled (idle-GPU dispatch, clock ramp)

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

// One value through per-sample min-max and log (data_utils.py:37-55): (v - mn) * inv with inv = 1 / max(mx - mn, 1e-8)
// formed ONCE per sample by an IEEE division, and ln as the hardware log2 (v_log_f32) times ln 2.  Against the fp64
// oracle this is as accurate as IEEE division per element + libm logf (max abs error 2.4e-6 vs 3.0e-6 on ln, 1.1e-7 on
// the [0, 1] value either way: the fp32 rounding of the inputs dominates; scripts/microbench/log_accuracy.hip,
// profiles/r3/log_accuracy.log) at a fifth of the instructions - the min-max / log epilogue is issue-bound.  Shared by
// the fused kernel's epilogue and k_minmax_log_apply, so both forms of the step give identical bits.
__device__ __forceinline__ float minmax_log_value(float v, float mn, float inv, int do_minmax, int do_log, float eps_log) {
    if (do_minmax) v = (v - mn) * inv;
    if (do_log) v = __builtin_amdgcn_logf(v + eps_log) * 0.69314718055994530942f;
    return v;
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
