// k_fused.h -- K1, the fused hot path: waveform -> (log-)mel in one kernel.
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// K1: fused wav -> mel magnitudes -> (FUSE) per-sample min-max and log, or (!FUSE) per-wave min/max partials for K2
//   work unit = chunk: consecutive frames of one clip, all C channels
//   grid      = min(#chunks, #CUs) workgroups looping over chunks; one workgroup per CU with every
//               wave the registers allow (16 at n_fft <= 1024, 12 with SpecAugment bands at 1024 and at 2048, 8
//               for the other 2048 variants: fused_waves)
//   per wave  = one frame at a time, claimed from the chunk's LDS queue, software-pipelined:
//                 frame i is windowed, transformed (registers + private padded LDS exchanges),
//                 untangled with the magnitude fused in, |X| goes to LDS; then - its sample
//                 registers now dead - frame i+1 is loaded into them straight from global memory
//                 and frame i+2 claimed, both in flight behind the banded mel reduction of frame i;
//                 lane m puts band m of the frame into the chunk's LDS tile (FUSE) or stores it straight to
//                 out[b, m, t, c] (the L2 merges the 4-byte stores).
//   LDS       = staging area of the constant block (re-read at the top of every chunk by the FUSE variants) |
//               exchange buffers [waves] (also |X|) | frame queue | mel table (mode 1) | time-band bitmap (BANDS) |
//               mel tile [M][pitch] + reduction scratch (FUSE); between prologue and epilogue the waves share
//               nothing but the queue
//   MELMODE 0 = band weights in registers (M <= 64, aligned band span <= 20, half spectrum): each lane
//               reads a 16-byte-aligned window of 20 magnitudes with 5 ds_read_b128
//           3 = band weights in registers, two bands per lane (64 < M <= 128, aligned band span <= 8
//               bins - e.g. the reference's 80 mel over 257 bins): 2 x 2 ds_read_b128 per frame
//           1 = band table staged in LDS as float4 rows, 2 = band table read from global (L1/L2)
//   HI        = some band needs bins above n_fft/4 (both halves of the untangle)
//   BANDS     = SpecAugment / filter bands present (time bands: per-chunk bitmap, masked frames skip
//               the transform; frequency bands: folded into the chunk's band weights)
//   S         = frames in flight per wave (1; 2 exists in diagnostic builds for n_fft 512 / 1024 and measured slower)
//   FUSE      = min-max / log inside the kernel (clip-level granule exchange, see the epilogue) instead of per-wave
//               partials for a second kernel: 1 = the chunk's mel values wait in an LDS tile and reach HBM once, finished;
//               2 = no tile (chunks of any size; round-5 experiment, selectable, never the default): the raw mel goes
//               to `out` as in the unfused form and the SAME workgroup finishes its chunk's rows in place once the clip's
//               range is known - its own stores, a barrier apart, read back through L2 / Infinity Cache.  Same bits; 3-5 %
//               slower than the two-kernel form at the batch sizes whose tile does not fit (EXPERIMENTS.md)
// ---------------------------------------------------------------------------
// One workgroup per CU holding every wave of the CU: all waves are of one age class for the issue
// arbiter (which favours older waves) and share one frame queue.
// waves per workgroup: 4 per SIMD at n_fft <= 1024 (<= 128 VGPRs: a twiddle takes one register pair, see
// cmul_tw) - except the n_fft 1024 variants with bands, which need 134 and stay at 3 per SIMD rather than
// spill (a kernel with scratch pays ~5 us more per dispatch); 2 when a wave keeps two frames in flight or at
// n_fft 2048.  (A/B at c2, both HBM-rotating and cache-resident: 16 waves = 12 waves within 0.3 %.)
#ifndef IRIS_W1024
#define IRIS_W1024 16
#endif
#ifndef IRIS_W2048
#define IRIS_W2048 0  // 0 = automatic: 12 where the registers allow it (see fused_waves), else 8
#endif
#ifndef IRIS_S2_WAVES
#define IRIS_S2_WAVES 8
#endif
// Round-5 occupancy experiment (A/B builds only, the defaults are the product; EXPERIMENTS.md): IRIS_EXP_WIN_LDS reads the
// window from the LDS staging area every frame instead of keeping it in 16 registers, IRIS_EXP_MELMODE1 (host_plan.h) takes
// the LDS band table instead of 20 register weights - together the n_fft 1024 kernel without epilogue fits 96 VGPRs = five
// waves per SIMD -, IRIS_WGS_PER_CU launches that many workgroups per CU (two of 10 waves: 1,024 threads cap one workgroup
// at 16 waves) for the two-kernel form
#ifndef IRIS_EXP_WIN_LDS
#define IRIS_EXP_WIN_LDS 0
#endif
#ifndef IRIS_WGS_PER_CU
#define IRIS_WGS_PER_CU 1
#endif
// frames go global -> registers up to this n_fft (log2); above it through LDS-DMA landing buffers.  Round 2: n_fft
// 2048 too - the prefetch targets the sample registers themselves (dead during the mel phase), so it costs no
// registers, frees 8 KB of landing buffer per wave, and the variant without bands and without the upper spectrum half
// then fits 168 VGPRs = 12 waves per CU (c5: 34.3 us LDS-DMA / 8 waves -> 32.5 direct / 8 -> 30.7 direct / 12).
#ifndef IRIS_DIRECT_MAX
#define IRIS_DIRECT_MAX 11
#endif
constexpr bool fused_direct(int log2n) { return IRIS_DIRECT_LOAD && log2n <= IRIS_DIRECT_MAX; }
// hi: the variant computes both halves of the untangle (n_fft 2048: 12 waves would spill, so 8)
#ifndef IRIS_FUSE2048_12
#define IRIS_FUSE2048_12 1
#endif
#define IRIS_FUSE12(fuse, mel_mode) (!(fuse) || (IRIS_FUSE2048_12 && (mel_mode) != 2 && (fuse) != 2))
// fuse: the variant applies min-max / log itself; its epilogue-only kernel arguments are loaded late (late_arg*), which
// keeps twelve waves at n_fft 2048 free of scratch - except with the global band table (mel_mode 2) and in the in-place
// form (fuse 2: the frame loop also keeps the output row addressing live), which stay at 8
constexpr int fused_waves(int log2n, int streams = 1, bool bands = false, bool hi = false, int fuse = 0, int mel_mode = 1) {
    return streams > 1 ? IRIS_S2_WAVES
                       : (log2n >= 11 ? ((IRIS_W2048 == 0 && fused_direct(11) && !bands && !hi && IRIS_FUSE12(fuse, mel_mode)) ? 12 : (IRIS_W2048 ? IRIS_W2048 : 8))
                                      : (log2n == 10 ? (bands ? 12 : IRIS_W1024) : 16));
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
    // fp16-MFMA mel variant (k_fused_mfma.h): A fragments [tiles][8 k-steps][64 lanes][8 halfs], per tile
    // (first k-step, k-step count), bins staged per frame (multiple of 32)
    const void* wfrag;
    const int* tile_ks;
    int kb;
    // fused epilogue (FUSE): min-max / log applied inside the kernel.  The chunk's mel values are collected in an LDS
    // tile [M][pitch] at byte offset tile_off; the workgroups of a clip exchange their (min, max) through 8-byte
    // {epoch, value} granules slots[chunk][2] (agent-scope sc1 stores / loads; epoch = the plan's launch counter,
    // never 0); a wait that does not complete within ~2 s sets *status and fills the chunk with NaN
    unsigned long long* slots;
    unsigned* status;  // device address of the plan's host-resident status word
    unsigned long long timeout_ticks;
    unsigned epoch;
    int tile_off, pitch, do_minmax, do_log;
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

typedef __attribute__((address_space(1))) unsigned long long gu64;

// Kernel arguments that only the epilogue reads, fetched from the kernarg segment WHERE THEY ARE USED: the compiler
// loads every field of the by-value argument struct at kernel entry and keeps it in scalar registers across the frame
// loop, which has none to spare (spilled SGPRs take a VGPR, and at n_fft 2048 that VGPR pushed mel weights to scratch).
// `asm volatile` keeps the load out of reach of loop-invariant code motion.
__device__ __forceinline__ unsigned late_arg32(unsigned offset) {
    unsigned v;
    asm volatile("s_load_dword %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "s"(offset) : "memory");
    return v;
}
__device__ __forceinline__ unsigned long long late_arg64(unsigned offset) {
    unsigned long long v;
    asm volatile("s_load_dwordx2 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)"
                 : "=s"(v) : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "s"(offset) : "memory");
    return v;
}
// The epilogue's arguments in ONE round trip to the scalar cache (five separate late loads, each with its own wait, stood
// on the critical path between a workgroup's last frame and its publication / write-out: ~0.1 us each)
struct LateEpilogueArgs {
    unsigned long long out, slots;
    unsigned epoch, do_minmax, do_log;
};
template <unsigned OFF_OUT, unsigned OFF_SLOTS, unsigned OFF_EPOCH, unsigned OFF_MM, unsigned OFF_LG>
__device__ __forceinline__ LateEpilogueArgs late_epilogue_args() {
    LateEpilogueArgs r;
    asm volatile("s_load_dwordx2 %0, %5, %6\n\t"
                 "s_load_dwordx2 %1, %5, %7\n\t"
                 "s_load_dword %2, %5, %8\n\t"
                 "s_load_dword %3, %5, %9\n\t"
                 "s_load_dword %4, %5, %10\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&s"(r.out), "=&s"(r.slots), "=&s"(r.epoch), "=&s"(r.do_minmax), "=&s"(r.do_log)
                 : "s"(__builtin_amdgcn_kernarg_segment_ptr()), "n"(OFF_OUT), "n"(OFF_SLOTS), "n"(OFF_EPOCH), "n"(OFF_MM), "n"(OFF_LG)
                 : "memory");
    return r;
}
#define LATE32(field) late_arg32((unsigned)offsetof(FusedArgs, field))
#define LATE64(field) late_arg64((unsigned)offsetof(FusedArgs, field))
constexpr unsigned long long kEpilogueTimeoutTicks = 200000000ull;  // s_memrealtime runs at 100 MHz: 2 s

template <int LOG2N, int MELMODE, bool HI, bool BANDS, int S, int FUSE>
__global__ __launch_bounds__(64 * fused_waves(LOG2N, S, BANDS, HI, FUSE, MELMODE), fused_waves(LOG2N, S, BANDS, HI, FUSE, MELMODE) * (FUSE == 0 ? IRIS_WGS_PER_CU : 1) / 4) void k_wav_to_mel(const FusedArgs a) {
    constexpr int kFusedWaves = fused_waves(LOG2N, S, BANDS, HI, FUSE, MELMODE);
    constexpr int N = 1 << LOG2N, NC = N / 2, P = FftCfg<LOG2N>::P, NTW = FftCfg<LOG2N>::NTW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // the wave index is uniform: keep it (and everything derived from it) in SGPRs
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // LDS: [waves][S] landing buffers (LDS-DMA targets, N floats) | [waves][S] exchange buffers
    // (also |X|) | frame queue | MELMODE 1 tables.  Nothing is shared between waves but the queue.
    constexpr int kXBufBytes = (lds_padded(NC, FftCfg<LOG2N>::PMMAX) * 8 + 15) & ~15;
    // landing area: LDS-DMA targets, or - with direct loads - only the staging area of the constant block
    constexpr int kLandBytes = fused_direct(LOG2N) ? ((ConstLayout<LOG2N>::NV4 * kWave * 16 + 15) & ~15) : kFusedWaves * S * N * 4;
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
    // MELMODE 1: the chunk's band table as float4 [rows / 4][M] (weights of 4 consecutive bins of a
    // band's 16-byte aligned window), with the clip's frequency bands folded in (all threads)
    auto build_wtab = [&](const int* fbc) {
        for (int i = threadIdx.x; i < a.M; i += blockDim.x) lotab[i] = a.band_lo[i];
        for (int i = threadIdx.x; i < a.rows * a.M; i += blockDim.x) {
            const int r = i / a.M, m = i - r * a.M;
            float w = a.wband[i];
            if (BANDS && fbc) {
                if (in_bands(fbc, a.n_fb, a.band_lo[m] + r)) w = 0.f;
            }
            wtab[((r >> 2) * a.M + m) * 4 + (r & 3)] = w;
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
    constexpr bool DIRECT = fused_direct(LOG2N);
    cf x[S][P];
    // Fetch of wave-frames ff[] (f = tl * C + c) of a chunk: straight into the x registers
    // (IRIS_DIRECT_LOAD), or by LDS-DMA into this wave's landing buffers
    auto issue_dma = [&](const int (&ff)[S], int b, int t0, int nwf) {
        const float* clip0 = a.wav + (size_t)b * a.C * a.L;
#pragma unroll
        for (int st = 0; st < S; ++st) {
            if (ff[st] < nwf && ABL(8) && ABL(64)) {  // diag: the frame comes from LDS instead (cost of an LDS-resident sample ring)
#pragma unroll
                for (int q = 0; q < P; ++q) x[st][q] = const_cast<const volatile cf*>(lds[st])[lane + kWave * q];
            }
            if (ff[st] < nwf && !ABL(8)) {
                const int tl = (a.C == 1) ? ff[st] : ff[st] / a.C, c = ff[st] - tl * a.C;
                if constexpr (DIRECT)
                    load_frame<LOG2N>(x[st], clip0 + (size_t)c * a.L, a.L, (t0 + tl) * a.hop - N / 2, lane);
                else
                    dma_frame<LOG2N>(clip0 + (size_t)c * a.L, a.L, (t0 + tl) * a.hop - N / 2, fbuf_lds[st], lane);
            }
        }
    };
    int f[S], fn[S];  // frames in registers / frames in flight to the landing buffers
    // The constant block is REQUESTED before the first frames (round 4): a wave's loads return in order, so with the frames
    // requested first the L2-resident constants could only be staged once the frames' HBM round trip had completed (the
    // staging loop's wait read vmcnt(0)), and the 0.6 us of LDS reads behind the barrier started from there.  Requested
    // first they are staged and read while the frames are still on their way (the wait in front of the LDS stores leaves the
    // frame loads outstanding).  (Round 2's IRIS_CONSTS_FIRST experiment reordered the source lines only and measured
    // "equal": the compiler had kept vmcnt(0).)
#ifndef IRIS_CONSTS_FIRST
#define IRIS_CONSTS_FIRST 1
#endif
    // The staging itself is LDS-DMA (global_load_lds_dwordx4 from inline asm, one 1-KiB row of the block per instruction,
    // rows w, w + W, ... by wave w): no registers, nothing for the compiler's wait-count pass to be conservative about -
    // for loads it knows it merges the interior / edge paths of the frame loads and falls back to vmcnt(0), which drains
    // the frames too.  The wait is set by hand below: vmcnt(P) once a wave has frame loads behind its rows (at most P
    // of them may stay outstanding: the rows are older and land first), vmcnt(0) for a wave without a frame.
    constexpr bool kConstsFirst = IRIS_CONSTS_FIRST && DIRECT;
    if constexpr (kConstsFirst) {
        const unsigned stage_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem);
        for (int v = wv; v < ConstLayout<LOG2N>::NV4; v += kFusedWaves)  // wave-uniform
            dma_frame_x4<8>(a.consts + v * (kWave * 4), stage_lds + (unsigned)v * (kWave * 16u), (unsigned)lane * 16u);
    }
    if (g0 < a.n_chunks) {  // first frames of the first chunk: in flight while the constants are staged and read
        const int b = chunk_clip(g0);
#pragma unroll
        for (int st = 0; st < S; ++st) f[st] = wv * S + st;
        issue_dma(f, b, chunk_t0(g0, b), chunk_nt(g0, b) * a.C);
    }
    {
        // The constant block is the same for every wave: fetch it from global once per workgroup.
        // With direct frame loads the landing area is free for it, so a wave whose constants have
        // arrived starts transforming while the others still read theirs (the 12 x 19 KB go through
        // one LDS pipe); with LDS-DMA frames it is staged through the exchange buffers, which need a
        // second barrier before the first FFT may overwrite them.
        static_assert(!DIRECT || kLandBytes >= kStageBytes, "constant block does not fit the landing area");
        float4* stage = reinterpret_cast<float4*>(DIRECT ? smem : xbuf0);
        const float4* g = reinterpret_cast<const float4*>(a.consts);
        if constexpr (kConstsFirst) {
            const bool has_frame = g0 < a.n_chunks && wv * S < chunk_nt(g0, chunk_clip(g0)) * a.C && !ABL(8);  // wave-uniform
            if (has_frame) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            for (int i = threadIdx.x; i < ConstLayout<LOG2N>::NV4 * kWave; i += blockDim.x) stage[i] = g[i];
        }
        if (threadIdx.x == 0) *next_frame = 2 * kFusedWaves * S;
        if constexpr (BANDS) {
            if (a.t_bands && g0 < a.n_chunks) {
                const int b = chunk_clip(g0);
                build_tbits(a.t_bands + (size_t)b * a.n_tb * 2, chunk_t0(g0, b), chunk_nt(g0, b));
            }
        }
        __syncthreads();
        // FUSE: the constants are (re)read from the staging area at the top of every chunk instead, so that they are
        // dead - and their 72 registers free - during the epilogue
        if constexpr (FUSE == 0) load_consts<LOG2N>(reinterpret_cast<const float*>(stage), lane, tw, post, win, wreg, lo0);
        if constexpr (MELMODE == 1) {
            const int* fbc = nullptr;
            if constexpr (BANDS) {
                if (a.f_bands && g0 < a.n_chunks) fbc = a.f_bands + (size_t)chunk_clip(g0) * a.n_fb * 2;
            }
            build_wtab(fbc);
        }
        if constexpr (!DIRECT || MELMODE == 1) __syncthreads();
        if ABL(512) {
            stamp0 = __builtin_amdgcn_s_memtime();
            real0 = __builtin_amdgcn_s_memrealtime();
        }
    }
    for (int chunk = g0; chunk < a.n_chunks; chunk += gridDim.x) {
        PH_BEGIN();
        if constexpr (FUSE != 0) {
            static_assert(FUSE == 0 || DIRECT, "the staging area must survive the frame loop");
            load_consts<LOG2N>(reinterpret_cast<const float*>(smem), lane, tw, post, win, wreg, lo0);
        }
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
                    if (FUSE == 0 && chunk != g0) reload_wreg<LOG2N>(opaque(a.consts), lane, wreg);  // pristine weights (not hoisted)
                    for (int i = 0; i < a.n_fb; ++i) {  // band bounds are wave-uniform (scalar loads)
                        const int off = fb[2 * i] - lo0, end = off + fb[2 * i + 1];
#pragma unroll
                        for (int r = 0; r < kMelRegs; ++r)
                            if (r >= off && r < end) wreg[r] = 0.f;
                    }
                }
            }
            if constexpr (MELMODE == 3) {  // two windows of 8 bins: wreg[0..7] at lo0 & 0xffff, wreg[8..15] at lo0 >> 16
                if (fb) {
                    if (FUSE == 0 && chunk != g0) reload_wreg<LOG2N>(opaque(a.consts), lane, wreg);
                    for (int i = 0; i < a.n_fb; ++i) {
                        const int offa = fb[2 * i] - (lo0 & 0xffff), enda = offa + fb[2 * i + 1];
                        const int offb = fb[2 * i] - (lo0 >> 16), endb = offb + fb[2 * i + 1];
#pragma unroll
                        for (int r = 0; r < 8; ++r) {
                            if (r >= offa && r < enda) wreg[r] = 0.f;
                            if (r >= offb && r < endb) wreg[8 + r] = 0.f;
                        }
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
        const unsigned rowpitch_b = FUSE == 1 ? 0u : (unsigned)a.T * (unsigned)a.C * 4u;
        float* const chunk_out = FUSE == 1 ? nullptr : a.out + ((size_t)b * a.M * a.T + t0) * a.C;
        // FUSE: the value goes to the chunk's LDS tile [M][pitch] instead (pitch is odd: the 64 lanes of a frame hit
        // 64 different banks); min-max / log and the coalesced write-out follow once the clip's range is known
        float* const tile = reinterpret_cast<float*>(smem + a.tile_off);
        auto store_band = [&](int fidx, int m, float v) {
            if constexpr (FUSE == 1) {
                tile[__umul24((unsigned)m, (unsigned)a.pitch) + fidx] = v;
            } else {
                const unsigned off = __umul24((unsigned)m, rowpitch_b);  // host checks rowpitch < 2^24
                if (!ABL(16))
                    asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"(v), "s"(chunk_out + fidx) : "memory");
            }
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
                            for (int m = lane; m < a.M; m += kWave) store_band(fcur[st], m, 0.f);
                    mn = fminf(mn, 0.f);
                    mx = fmaxf(mx, 0.f);
                    advance();
                    continue;
                }
            }
            // (direct loads: the compiler waits for each sample register where it is first used; stores issued
            // from inline asm only make those counted waits more conservative, never less)
            PH_MARK(0);
            if constexpr (IRIS_EXP_WIN_LDS && DIRECT && (P % 2 == 0)) {
                // experiment: window pairs (win[2k], win[2k + 1]) = one float4 of the staged constant block per lane
                const float4* w4 = reinterpret_cast<const float4*>(smem) + (ConstLayout<LOG2N>::OFF_WIN / 4) * kWave + lane;
#pragma unroll
                for (int st = 0; st < S; ++st)
#pragma unroll
                    for (int k = 0; k < P / 2; ++k) {
                        const float4 w = w4[k * kWave];
                        x[st][2 * k] *= mk(w.x, w.y);
                        x[st][2 * k + 1] *= mk(w.z, w.w);
                    }
            } else {
#pragma unroll
                for (int st = 0; st < S; ++st)
#pragma unroll
                    for (int q = 0; q < P; ++q) x[st][q] *= win[q];
            }
            if (!ABL(1)) fft_frames<LOG2N, S, LOG2N == IRIS_SINGLE_READS_LOG2N>(x, tw, lds, lane);
            PH_MARK(3);
            // |X| scaled by 2 (the 0.5 of the untangle lives in the band weights)
            if (!ABL(2)) untangle_mag<LOG2N, HI, S>(x, post, lds, magbuf, lane);
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
                if constexpr (MELMODE == 3) {
                    // two bands per lane (m = lane and lane + 64), 8-bin windows: four reads, then the FMAs
                    const float4* ga = reinterpret_cast<const float4*>(magbuf[st] + (lo0 & 0xffff));
                    const float4* gb = reinterpret_cast<const float4*>(magbuf[st] + (lo0 >> 16));
                    const float4 a0 = ga[0], a1 = ga[1], b0 = gb[0], b1 = gb[1];
                    cf sa = mk(wreg[0], wreg[1]) * mk(a0.x, a0.y), sb = mk(wreg[8], wreg[9]) * mk(b0.x, b0.y);
                    cf ta = mk(wreg[2], wreg[3]) * mk(a0.z, a0.w), tb2 = mk(wreg[10], wreg[11]) * mk(b0.z, b0.w);
                    sa = __builtin_elementwise_fma(mk(wreg[4], wreg[5]), mk(a1.x, a1.y), sa);
                    sb = __builtin_elementwise_fma(mk(wreg[12], wreg[13]), mk(b1.x, b1.y), sb);
                    ta = __builtin_elementwise_fma(mk(wreg[6], wreg[7]), mk(a1.z, a1.w), ta);
                    tb2 = __builtin_elementwise_fma(mk(wreg[14], wreg[15]), mk(b1.z, b1.w), tb2);
                    sa += ta;
                    sb += tb2;
                    if (live[st]) {
                        const float va = (sa.x + sa.y) * keep, vb = (sb.x + sb.y) * keep;
                        store_band(fcur[st], lane, va);
                        mn = fminf(mn, va);
                        mx = fmaxf(mx, va);
                        if (lane + kWave < a.M) {
                            store_band(fcur[st], lane + kWave, vb);
                            mn = fminf(mn, vb);
                            mx = fmaxf(mx, vb);
                        }
                    }
                } else if constexpr (MELMODE == 0) {
                    const float4* mag4 = reinterpret_cast<const float4*>(magbuf[st] + lo0);  // lo0 % 4 == 0
                    // packed FMAs on two independent accumulators (a dependent packed op costs
                    // a wait state)
                    cf acc2 = mk(0.f, 0.f), acc3 = mk(0.f, 0.f);
#pragma unroll
                    for (int i = 0; i < kMelRegs / 4; ++i) {
                        const float4 m4 = ABL(4) ? make_float4(1.f, 1.f, 1.f, 1.f) : mag4[i];
                        acc2 = __builtin_elementwise_fma(mk(wreg[4 * i + 0], wreg[4 * i + 1]), mk(m4.x, m4.y), acc2);
                        acc3 = __builtin_elementwise_fma(mk(wreg[4 * i + 2], wreg[4 * i + 3]), mk(m4.z, m4.w), acc3);
                    }
                    acc2 += acc3;
                    const float acc = acc2.x + acc2.y;
                    if (live[st] && lane < a.M) {
                        const float v = acc * keep;
                        store_band(fcur[st], lane, v);
                        mn = fminf(mn, v);
                        mx = fmaxf(mx, v);
                    }
                } else {
                    if (live[st]) {
                        for (int m = lane; m < a.M; m += kWave) {
                            float acc = 0.f;
                            if (ABL(32)) {
                                acc = 1.f;
                            } else if constexpr (MELMODE == 1) {
                                const float4* g4 = reinterpret_cast<const float4*>(magbuf[st] + lotab[m]);  // lo % 4 == 0
                                const float4* w4 = reinterpret_cast<const float4*>(wtab) + m;
                                cf acc2 = mk(0.f, 0.f), acc3 = mk(0.f, 0.f);
                                for (int i4 = 0; i4 < a.rows / 4; ++i4) {
                                    const float4 w = w4[i4 * a.M], g = g4[i4];
                                    acc2 = __builtin_elementwise_fma(mk(w.x, w.y), mk(g.x, g.y), acc2);
                                    acc3 = __builtin_elementwise_fma(mk(w.z, w.w), mk(g.z, g.w), acc3);
                                }
                                acc2 += acc3;
                                acc = acc2.x + acc2.y;
                            } else {
                                const int lo = a.band_lo[m];
                                for (int i = 0; i < a.rows; ++i)
                                    acc = fmaf(a.wband[i * a.M + m], magbuf[st][lo + i], acc);
                            }
                            const float v = acc * keep;
                            store_band(fcur[st], m, v);
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
        mn = wave_min(mn);
        mx = wave_max(mx);
        if constexpr (FUSE == 0) {
            // every wave leaves its own (min, max) partial for k_minmax_log_apply
            if (lane == 0) {
                a.partial[((size_t)chunk * kFusedWaves + wv) * 2 + 0] = mn;
                a.partial[((size_t)chunk * kFusedWaves + wv) * 2 + 1] = mx;
            }
        } else {
            // Fused epilogue: the chunk's mel values sit in the LDS tile.  Workgroup range -> (clips split over several
            // workgroups) one 8-byte {epoch, value} granule each for min and max, published with agent-scope stores and
            // swept by wave 0 until every chunk of the clip carries this launch's epoch -> minmax_log_value: (x - min) *
            // (1 / max(max - min, 1e-8)), ln(x + 1e-8) -> coalesced rows of out[b, m, t0 .. t0 + nt, :].  Every workgroup publishes before
            // it waits and all workgroups of the grid are resident (grid <= CUs), so the waits always complete; the
            // sweep is bounded all the same (status word + NaN output instead of a hang).
            float* red = tile + (FUSE == 1 ? (size_t)a.M * a.pitch : 0);  // [2 * waves + 4]
            int le = lane;  // epilogue lane index, hidden from loop-invariant code motion: per-lane addresses of the
            asm volatile("" : "+v"(le));  // epilogue must not be hoisted across the frame loop (they would spill there)
            const LateEpilogueArgs late = late_epilogue_args<(unsigned)offsetof(FusedArgs, out), (unsigned)offsetof(FusedArgs, slots),
                                                             (unsigned)offsetof(FusedArgs, epoch), (unsigned)offsetof(FusedArgs, do_minmax),
                                                             (unsigned)offsetof(FusedArgs, do_log)>();
            if (lane == 0) {
                red[wv] = mn;
                red[kFusedWaves + wv] = mx;
            }
            // FUSE 2: the frame loop's stores were issued from inline asm, invisible to the compiler's wait-count pass - this
            // wave's raw mel must have reached the L2 before any wave of the workgroup reads it back (a workgroup's waves
            // share one vector L1, write-through: workgroup-scope visibility needs the wait and the barrier, no cache action)
            if constexpr (FUSE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();  // tile complete, wave ranges visible
            if (ABL(512) && threadIdx.x == 0 && a.dbg) a.dbg[4 + kDbgWg * blockIdx.x + 3] = __builtin_amdgcn_s_memrealtime();
            if (wv == 0) {
                float lo = le < kFusedWaves ? red[le] : INFINITY, hi = le < kFusedWaves ? red[kFusedWaves + le] : -INFINITY;
                lo = wave_min(lo);
                hi = wave_max(hi);
                unsigned failed = 0;
                if (late.do_minmax && a.chunks_per_clip > 1) {
                    gu64* slots = (gu64*)late.slots + 2 * (size_t)b * a.chunks_per_clip;
                    const unsigned epoch = late.epoch;
                    const unsigned long long tag = (unsigned long long)epoch << 32;
                    if (lane == 0) {
                        const int ci = chunk - b * a.chunks_per_clip;
                        __hip_atomic_store(slots + 2 * ci, tag | __float_as_uint(lo), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(slots + 2 * ci + 1, tag | __float_as_uint(hi), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    const unsigned long long t_begin = __builtin_amdgcn_s_memrealtime();
                    for (;;) {
                        bool ok = true;
                        float l2 = INFINITY, h2 = -INFINITY;
                        for (int i = le; i < a.chunks_per_clip; i += kWave) {
                            const unsigned long long g0v = __hip_atomic_load(slots + 2 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            const unsigned long long g1v = __hip_atomic_load(slots + 2 * i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = ok && ((g0v >> 32) == epoch) && ((g1v >> 32) == epoch);
                            l2 = fminf(l2, __uint_as_float((unsigned)g0v));
                            h2 = fmaxf(h2, __uint_as_float((unsigned)g1v));
                        }
                        if (__all(ok)) {
                            lo = wave_min(l2);
                            hi = wave_max(h2);
                            break;
                        }
                        if (__builtin_amdgcn_s_memrealtime() - t_begin >= LATE64(timeout_ticks)) {
                            failed = 1;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if (lane == 0) {
                    red[2 * kFusedWaves + 0] = lo;
                    red[2 * kFusedWaves + 1] = hi;
                    red[2 * kFusedWaves + 2] = __uint_as_float(failed);
                    if (failed) __hip_atomic_store((unsigned*)LATE64(status), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
            __syncthreads();
            if (ABL(512) && threadIdx.x == 0 && a.dbg) a.dbg[4 + kDbgWg * blockIdx.x + 4] = __builtin_amdgcn_s_memrealtime();
            const float cmn = red[2 * kFusedWaves], cmx = red[2 * kFusedWaves + 1];
            const bool failed = __float_as_uint(red[2 * kFusedWaves + 2]) != 0;
            const float inv = 1.0f / fmaxf(cmx - cmn, 1e-8f);
            const int mm = (int)late.do_minmax, lg = (int)late.do_log;
            const unsigned rowpitch_e = (unsigned)a.T * (unsigned)a.C * 4u;
            float* const out_e = (float*)late.out + ((size_t)b * a.M * a.T + t0) * a.C;
            // Write-out: wave w owns rows w, w + W, ... of the tile and walks them row by row, 64 columns at a time, with every
            // address formed on the SCALAR unit (row base in an SGPR pair, lane offset = 4 lane): 8 vector instructions per trip.
            // All waves of the CU run this phase at the same time, so it is bound by vector ISSUE: the round-3 form (one flat
            // run over the wave's elements: 5 trips instead of 8 for 79 columns, but ~35 vector instructions per trip for the
            // (row, column) walk, a 32-bit multiply and the selects, plus an integer division up front) cost as many issue
            // slots as a whole frame's transform.
            int we = wv;  // the wave index, hidden from loop-invariant code motion like `le` (row offsets would otherwise be
            asm volatile("" : "+s"(we));  // precomputed into scalar registers that the frame loop has none to spare for)
            const unsigned l4 = (unsigned)le * 4u;
            const float* trow = tile + (size_t)we * a.pitch;   // uniform (LDS)
            const char* grow = reinterpret_cast<const char*>(out_e) + (size_t)we * rowpitch_e;  // uniform
            int wstep = kFusedWaves;  // laundered like `we`: the two row strides below are chunk-invariant and would otherwise be
            asm volatile("" : "+s"(wstep));  // hoisted into scalar registers that live across the frame loop
            const size_t tstep = (size_t)wstep * a.pitch, gstep = (size_t)wstep * rowpitch_e;
            if constexpr (FUSE == 2) {
                // In place: wave w finishes rows w, w + W, ... of ITS chunk's columns out[b, m, t0 .. t0 + nt, :] - four 256-byte
                // pieces of a row in flight per trip, read back from where this workgroup's own frame loop put them
                for (int m2 = we; m2 < a.M; m2 += kFusedWaves, grow += gstep) {  // uniform
                    float* const row = reinterpret_cast<float*>(const_cast<char*>(grow));
                    for (int c1 = 0; c1 < nwf; c1 += 4 * kWave) {  // uniform
                        float v[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int idx = c1 + j * kWave + le;
                            v[j] = idx < nwf ? row[idx] : 0.f;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int idx = c1 + j * kWave + le;
                            float y = v[j];
                            if (mm) y = (y - cmn) * inv;
                            if (lg) y = __builtin_amdgcn_logf(y + 1e-8f) * 0.69314718055994530942f;  // = minmax_log_value, same bits
                            if (failed) y = NAN;
                            if (idx < nwf) row[idx] = y;
                        }
                    }
                }
            }
            int m = FUSE == 2 ? a.M : we, c0 = 0;
            // software-pipelined by one trip: the next LDS read is in flight behind this trip's math and store
            float cur = (m < a.M && le < nwf) ? trow[le] : 0.f;
            while (m < a.M) {  // uniform
                const bool act = c0 + le < nwf;
                const char* gdst = grow + (size_t)c0 * 4;
                c0 += kWave;
                if (c0 >= nwf) {
                    c0 = 0;
                    m += kFusedWaves;
                    trow += tstep;
                    grow += gstep;
                }
                const float nxt = (m < a.M && c0 + le < nwf) ? trow[c0 + le] : 0.f;
                float y = cur;
                if (mm) y = (y - cmn) * inv;                                   // uniform branches: the flags are scalars
                if (lg) y = __builtin_amdgcn_logf(y + 1e-8f) * 0.69314718055994530942f;   // = minmax_log_value, same bits
                if (failed) y = NAN;
                if (act) asm volatile("global_store_dword %0, %1, %2" ::"v"(l4), "v"(y), "s"(gdst) : "memory");
                cur = nxt;
            }
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
        a.dbg[4 + kDbgWg * blockIdx.x + 0] = real_entry;
        a.dbg[4 + kDbgWg * blockIdx.x + 1] = real0;
        a.dbg[4 + kDbgWg * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
    }
}
