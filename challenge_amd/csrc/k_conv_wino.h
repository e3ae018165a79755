// k_conv_wino.h -- Conv2D(3x3 'same', Cin -> Cout) + bias + ReLU (+ MaxPool 2x2 'same') as a Winograd F(2x2, 3x3) transform on
// the fp32 matrix cores: blocks 2-5 of the CRNN (sj_train.py:191-201, 222-242: 64 ... 512 channels), for inference.
// Part of the single translation unit iris_frontend.hip (and of scripts/microbench/wino_conv.hip, which builds it alone).
#pragma once
// ---------------------------------------------------------------------------
// MIOpen runs these layers as implicit GEMMs at the fp32 MFMA rate (232-249 us for 38.7 GFLOP = the ceiling of
// v_mfma_f32_32x32x2_f32 at the clock the chip sustains), so only fewer multiplies can be faster: F(2x2, 3x3) needs 16 instead
// of 36 per 2 x 2 output tile and channel pair (2.25x).   Y = A^T [ (G g G^T) (.) (B^T d B) ] A   per tile, summed over the
// input channels BEFORE the output transform:
//   U[p][cin][cout] = G g G^T                     once per layer, on the host side of the ABI (iris_wino_pack_weights)
//   V[p][tile][cin] = B^T d B                     4 x 4 input patch d of a tile -> 16 positions p, computed on the fly
//   M[p][tile][cout] = sum_cin V U                16 GEMMs - the matrix cores
//   Y[tile][2 x 2][cout] = A^T M A, + bias, ReLU (a 2 x 2 output tile IS a pooling window: MaxPool = an in-lane max)
// Decomposition: a wave owns 32 tiles x 32 output channels x all 16 positions = 16 accumulators of v_mfma_f32_32x32x2_f32
// (256 AGPRs: one wave per SIMD, 512-register budget); a workgroup of 4 waves = 64 tiles (TR tile rows x TC tile columns)
// x 64 output channels.  K runs over the input channels in chunks of 8, everything double-buffered in LDS.
// Activations are CHANNEL-CHUNKED: x [B][Cin / 8][H][W][8] - a chunk of 8 channels of a row of pixels is contiguous, so
// the input of a chunk arrives by LDS-DMA (global_load_lds_dwordx4: no registers, no wait counters shared with the
// compiler's own loads) in whole cache lines; out-of-image pixels are LDS slots zeroed once per block and masked out of the DMA.  Each wave stages,
// privately, exactly the pixels ITS transform items read (3 patch rows x one 16-byte channel group), so the hand-over raw ->
// patch registers -> next DMA needs no workgroup barrier.  U comes by LDS-DMA too (the host packs it in LDS order).
// One wave per SIMD means nothing else hides latency: the K loop is written as 64 SLOTS per chunk - one MFMA each, fenced
// by scheduling barriers - and every other instruction of the chunk (the LDS reads of the next operand group, the transform
// of the NEXT chunk and its LDS writes, the DMA requests of the chunk after that) sits in the shadow of one of those MFMAs.
// The transformed input (4x the activation) and M (4x the output) never exist in memory.
// y: [B][Cout / 8][Ho][Wo][8] (the next layer's input) or channels-last [B][Ho][Wo][Cout] (`out_nhwc`); Cin % 8 == 0,
// Cout % 64 == 0.
// ---------------------------------------------------------------------------
#ifdef IRIS_WINO_STANDALONE
typedef float f32x16 __attribute__((ext_vector_type(16)));
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
#include "bn_epilogue.h"

// timing experiments only (results wrong when non-zero): 1 no per-chunk barrier, 2 no LDS-DMA of U, 4 no transform of the next
// chunk (patch reads, row / column stage, V writes), 8 no operand reads after a chunk's first group, 16 no LDS-DMA of the input
#ifndef IRIS_WINO_ABLATE
#define IRIS_WINO_ABLATE 0
#endif
#define WINO_ABL(bit) ((IRIS_WINO_ABLATE & (bit)) != 0)
constexpr int kWinoTM = 64, kWinoTN = 64, kWinoKC = 8;
constexpr int kWinoPackMaxJobs = 48;
constexpr int kWinoVFloats = 16 * 4 * kWinoTM * 2;   // [pos 16][pair 2][hl 2][tile 64][2]
constexpr int kWinoUFloats = 16 * 4 * kWinoTN * 2;   // [pos 16][pair 2][hl 2][cout 64][2]
constexpr int kWinoBuf = kWinoVFloats + kWinoUFloats;
constexpr int kWinoRawPieces = 448;                  // per wave: 7 LDS-DMA rows of 64 pieces of 16 bytes (>= TR 3 (2 TC + 2))
constexpr size_t kWinoLdsBytes = 2 * (size_t)kWinoBuf * sizeof(float) + 4 * (size_t)kWinoRawPieces * 16;  // 128 KiB + 28 KiB

// floats of the packed weights: [cout block][chunk][pos][pair][hl][cout 64][2]
static size_t wino_packed_floats(int cin, int cout) { return (size_t)16 * cin * cout; }

typedef __attribute__((address_space(3))) float wino_lds_float;
__device__ __forceinline__ unsigned wino_lds_addr(const float* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(wino_lds_float*)p);
}
// 1 KiB global -> LDS without registers: lane l moves 16 bytes to LDS byte address lds + 16 l (M0 = lds) - from src + 16 l
// (uniform source: a contiguous row) or from its own address (gathered pieces).
// Inline asm: the compiler would drain an LDS-DMA it knows about before the next LDS access.
__device__ __forceinline__ void wino_dma16(const float* src /*uniform*/, unsigned lds /*uniform*/, unsigned lane16) {
    const uint64_t a = reinterpret_cast<uint64_t>(src);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const float* s = reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(lane16), "s"(s), "s"(lds) : "memory");
}
// the same for gathered pieces: lane l moves 16 bytes from base + off[l] (base uniform, off a 32-bit byte offset) - the lanes of
// `mask` only: the others touch neither memory nor their LDS slot (which the caller has zeroed: out-of-image pixels)
__device__ __forceinline__ void wino_dma16_gather(const float* base /*uniform*/, unsigned off, unsigned long long mask /*uniform*/,
                                                  unsigned lds /*uniform*/) {
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const float* s = reinterpret_cast<const float*>(((uint64_t)hi << 32) | lo);
    unsigned keep;
    unsigned long long keep_exec;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %5\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, %4\n\t"
                 "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "=&s"(keep_exec) : "v"(off), "s"(s), "s"(mask), "s"(lds) : "memory");
}
typedef struct { f32x2 lo, hi; } wino_v4;  // four channels as two register pairs: sums and differences are v_pk_add_f32
// packed adds / subtractions written out (the compiler scalarises both the vector fsub and the fma-by-(-1) form of it here):
// ONE v_pk_add_f32 per register pair, the sign of b on the VOP3P neg modifiers
__device__ __forceinline__ f32x2 wino_add2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ f32x2 wino_sub2(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ wino_v4 operator+(wino_v4 a, wino_v4 b) { return {wino_add2(a.lo, b.lo), wino_add2(a.hi, b.hi)}; }
__device__ __forceinline__ wino_v4 operator-(wino_v4 a, wino_v4 b) { return {wino_sub2(a.lo, b.lo), wino_sub2(a.hi, b.hi)}; }

struct WinoWave {
    f32x16 acc[16];
    f32x2 ops[2][8];      // operand pairs of a group of 4 positions: [0..3] = V (A operand), [4..7] = U (B operand); double-buffered
    wino_v4 d[3][4];      // this thread's patch rows (three of the four) x columns, one channel group of 4
    wino_v4 w[2][4];      // B^T d, two rows xi of this thread's half
};

// TC: tile columns of a workgroup's block (64 / TC tile rows): min(ceil(W / 2), 64) rounded up to a power of two
// in_nhwc: x is channels-last [B][H][W][Cin] (a chunk of a pixel = 32 bytes at a stride of Cin x 4: every gather touches 64
// cache lines instead of 16 - 14 % slower over the CRNN's layers, what a training pass pays for keeping its activations where
// the weight-gradient kernel (k_conv_wino_wrw.h: lane = channel) reads them); relu == 0 and bias == nullptr: the bare convolution (training: BatchNorm follows)
template <bool POOL, int TC, bool IN_NHWC, bool BN = false>
__global__ __launch_bounds__(256, 1) void k_conv3x3_wino(const float* __restrict__ x, const float* __restrict__ u,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
                                                         int Cin, int Cout, int out_nhwc, int relu, double* __restrict__ bn_sums) {
    // BN / bn_sums (training form only: bare convolution, channels-last out, no pooling): the per-channel sum z / sum z^2 of the
    // BatchNorm that follows, accumulated here from the registers the output is stored from (bn_epilogue.h); a template parameter so
    // that the instantiations without it are exactly the code measured before (see k_conv_wino_b3.h)
    constexpr bool in_nhwc = IN_NHWC;
    extern __shared__ __attribute__((aligned(16))) float wino_lds[];
    constexpr int TR = kWinoTM / TC, PW = 2 * TC + 2, kPieces = TR * 3 * PW, kDmaRows = (kPieces + 63) / 64;
    static_assert(kPieces <= kWinoRawPieces, "raw region too small");
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // the wave index is uniform
    const int hl = lane >> 5, li = lane & 31;
    const int wm = wv & 1, wn = wv >> 1;  // the wave's tile half / output-channel half of the workgroup's 64 x 64 block
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    const int n_rows = B * TH;                                    // tile rows of the whole batch
    const int row_blocks = (n_rows + TR - 1) / TR, col_blocks = (TW + TC - 1) / TC;
    const int cout_blocks = Cout / kWinoTN, n_chunks = Cin / kWinoKC;
    const int n_work = row_blocks * col_blocks * cout_blocks;
    const size_t plane = (size_t)H * W * 8;                       // floats of one channel chunk of one image
    // transform role of this wave: channel group (4 of the chunk's 8 channels) and half of the 16 positions; lane = tile
    const int p_cg = wv & 1, p_half = wv >> 1;
    float* const raw = wino_lds + 2 * kWinoBuf + wv * (kWinoRawPieces * 4);   // this wave's private staging area
    const unsigned raw_l = wino_lds_addr(raw);
    WinoWave s;

    for (int work = blockIdx.x; work < n_work; work += gridDim.x) {
        // consecutive work items share a tile block (its input stays in L2) and walk the output-channel blocks
        const int cb = work % cout_blocks, blk = work / cout_blocks;
        const int cbk = blk % col_blocks, rb = blk / col_blocks;
        const int R0 = rb * TR, tc0 = cbk * TC;
        const float* const ubase = u + ((size_t)cb * n_chunks) * (kWinoUFloats);
        // ---- the input pieces this lane requests per chunk: piece i = lane + 64 k of the wave's staging order
        //      [strip (tile row of the block)][patch row r of this half][pixel px of the strip], 16 bytes = channel group p_cg
        // A piece's LDS slot, its offset and whether it lies inside the image are the same for every chunk of the work item:
        // byte offset inside a chunk plane (32 bit) + a lane mask per DMA row; the slots of out-of-image pieces are zeroed HERE,
        // once, and never requested (EXEC-masked DMA), so no zero source and no per-chunk address arithmetic is needed.
        unsigned poff[kDmaRows];
        unsigned long long pmask[kDmaRows];
        __syncthreads();  // the previous work item's last buffers - and every wave's last read of its staging area - are done
#pragma unroll
        for (int k = 0; k < kDmaRows; ++k) {
            const int i = lane + 64 * k;
            const int strip = i / (3 * PW), r = (i / PW) % 3, px = i % PW;
            const int R = R0 + strip, b_ = R / TH, th = R - b_ * TH;
            const int hh = 2 * th - 1 + p_half + r, ww = 2 * tc0 - 1 + px;
            const bool ok = i < kPieces && R < n_rows && hh >= 0 && hh < H && ww >= 0 && ww < W;
            poff[k] = !ok ? 0u : in_nhwc ? ((unsigned)(((size_t)b_ * H + hh) * W + ww) * (unsigned)Cin + 4u * p_cg) * 4u
                                         : ((unsigned)((((size_t)b_ * (Cin / 8)) * H + hh) * W + ww) * 8u + 4u * p_cg) * 4u;
            pmask[k] = __ballot(ok);
            if (!ok && i < kWinoRawPieces) reinterpret_cast<float4*>(raw)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        auto dma_x = [&](int chunk, int k0, int k1) {  // rows k0 .. k1 - 1 of the wave's staging area for `chunk`
            const float* xc = x + (size_t)chunk * (in_nhwc ? (size_t)8 : plane);  // uniform
#pragma unroll
            for (int k = 0; k < kDmaRows; ++k)
                if (k >= k0 && k < k1) wino_dma16_gather(xc, poff[k], pmask[k], raw_l + 1024u * k);
        };
        auto dma_u = [&](int chunk, int buf, int j0, int j1) {  // the chunk of U: one contiguous 32 KiB block in LDS order, 8 rows of
            const float* us_g = ubase + (size_t)chunk * kWinoUFloats;  // 1 KiB per wave (rows j0 .. j1 - 1 of them)
            const unsigned us_l = wino_lds_addr(wino_lds + buf * kWinoBuf + kWinoVFloats);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j >= j0 && j < j1) wino_dma16(us_g + (wv + 4 * j) * 256, us_l + (wv + 4 * j) * 1024u, (unsigned)lane * 16u);
        };
        // this lane's tile inside the block and its patch in the staging area
        const int trow = lane / TC, tcol = lane % TC;
        const wino_v4* const patch = reinterpret_cast<const wino_v4*>(raw) + (trow * 3) * PW + 2 * tcol;
        auto read_patch = [&](int i) { s.d[i >> 2][i & 3] = patch[(i >> 2) * PW + (i & 3)]; };  // i = 4 r + c
        auto row_stage = [&](auto half_tag, int e, int c) {
            // rows of B^T d: half 0 holds patch rows 0, 1, 2 -> xi 0 = d0 - d2, xi 1 = d1 + d2; half 1 holds rows 1, 2, 3 ->
            // xi 2 = d2 - d1, xi 3 = d1 - d3
            constexpr int half = decltype(half_tag)::value;
            const wino_v4 a = s.d[0][c], b = s.d[1][c], dd = s.d[2][c];
            if constexpr (half == 0) s.w[e][c] = e == 0 ? a - dd : b + dd;
            else s.w[e][c] = e == 0 ? b - a : a - dd;
        };
        auto col_stage = [&](int buf, int e, int nu) {
            const wino_v4 (&w)[4] = s.w[e];
            const wino_v4 v = nu == 0 ? w[0] - w[2] : nu == 1 ? w[1] + w[2] : nu == 2 ? w[2] - w[1] : w[1] - w[3];
            const int pos = 4 * (2 * p_half + e) + nu;
            // the chunk's channels 4 p_cg + {0, 1, 2, 3} are operand pair p_cg: lanes hl 0 take (.x, .y) for their two MFMA
            // steps, lanes hl 1 take (.z, .w) - register pairs as they are (iris_wino_pack_weights orders U the same way)
            f32x2* dst = reinterpret_cast<f32x2*>(wino_lds + buf * kWinoBuf) + ((pos * 2 + p_cg) * 2) * kWinoTM + lane;
            dst[0] = v.lo;
            dst[kWinoTM] = v.hi;
        };
        auto read_op = [&](f32x2 (&ops)[8], int buf, int gi, int j) {  // operand j of group gi = (pr, g): positions 4 g + (j & 3)
            const int pr = gi >> 2, p = 4 * (gi & 3) + (j & 3);
            const f32x2* base = reinterpret_cast<const f32x2*>(wino_lds + buf * kWinoBuf);
            ops[j] = j < 4 ? base[((p * 2 + pr) * 2 + hl) * kWinoTM + 32 * wm + li]
                           : base[kWinoVFloats / 2 + ((p * 2 + pr) * 2 + hl) * kWinoTN + 32 * wn + li];
        };

#pragma unroll
        for (int p = 0; p < 16; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) s.acc[p][r] = 0.f;

        // ---- prologue (not overlapped): V(0), U(0) into buffer 0; the patches of chunk 1 in registers; the input of chunk 2 requested
        dma_u(0, 0, 0, 8);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the zeros above are in place before any request can land beside them
        dma_x(0, 0, kDmaRows);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 12; ++i) read_patch(i);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the staging area is free again
        if (n_chunks > 1) dma_x(1, 0, kDmaRows);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (p_half == 0) row_stage(std::integral_constant<int, 0>{}, e, c);
                else row_stage(std::integral_constant<int, 1>{}, e, c);
            }
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) col_stage(0, e, nu);
        }
        if (n_chunks > 1) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 12; ++i) read_patch(i);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }

        // One chunk: 8 groups (4 positions x the two MFMA steps of an operand pair) of 8 slots.  NEXT: another chunk follows
        // (its transform rides in this chunk's slots); the last chunk of a work item runs the bare MFMA sequence.
        // Timeline of the wave's own memory traffic in chunk c (a wave's vector-memory operations return in order):
        //   slots  0 ..  6  request the input of chunk c + 2 into the staging area (its previous content, chunk c + 1, went to
        //                   registers at the end of chunk c - 1)
        //   slots  8 .. 15  request U(c + 1) into buffer buf ^ 1
        //   slots 20 .. 27  row stage, 32 .. 39 column stage + V(c + 1) writes into buffer buf ^ 1
        //   slot  52        wait until at most the 8 U requests are outstanding: the staging area holds chunk c + 2
        //   slots 52 .. 63  patches of chunk c + 2 -> registers (consumed during chunk c + 1)
        //   top of c + 1    wait for everything (U(c + 1) has landed), barrier
        auto chunk_body = [&](auto next_tag, auto half_tag, int chunk) {
            constexpr bool next = decltype(next_tag)::value;
            const int buf = chunk & 1;
            // (the last two chunks of a work item request / read a chunk that exists instead of chunk + 2: valid memory, never used)
            const int chunk2 = min(chunk + 2, n_chunks - 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!WINO_ABL(1)) __syncthreads();  // buffer `buf` is complete; buffer buf ^ 1 is free (its readers passed this barrier)
#pragma unroll
            for (int j = 0; j < 8; ++j) read_op(s.ops[0], buf, 0, j);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gi = 0; gi < 8; ++gi) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = j & 3, p = 4 * (gi & 3) + q, sl = 8 * gi + j;
                    const f32x2 a = s.ops[gi & 1][q], b = s.ops[gi & 1][4 + q];
                    s.acc[p] = (j < 4) ? __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, s.acc[p], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, s.acc[p], 0, 0, 0);
                    if (gi < 7 && !WINO_ABL(8)) read_op(s.ops[(gi + 1) & 1], buf, gi + 1, j);   // next group's operand j
                    if constexpr (next) {
                        if (sl < kDmaRows && !WINO_ABL(16)) dma_x(chunk2, sl, sl + 1);
                        if (sl >= 8 && sl < 16 && !WINO_ABL(2)) dma_u(chunk + 1, buf ^ 1, sl - 8, sl - 7);
                        if (sl >= 20 && sl < 28 && !WINO_ABL(4)) row_stage(half_tag, (sl - 20) >> 2, (sl - 20) & 3);
                        if (sl >= 32 && sl < 40 && !WINO_ABL(4)) col_stage(buf ^ 1, (sl - 32) >> 2, (sl - 32) & 3);
                        if (sl == 52 && !WINO_ABL(4)) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                        if (sl >= 52 && !WINO_ABL(4)) read_patch(sl - 52);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        if (p_half == 0) {  // wave-uniform: two copies of the loop, no per-lane select in the row stage
            for (int chunk = 0; chunk + 1 < n_chunks; ++chunk) chunk_body(std::true_type{}, std::integral_constant<int, 0>{}, chunk);
        } else {
            for (int chunk = 0; chunk + 1 < n_chunks; ++chunk) chunk_body(std::true_type{}, std::integral_constant<int, 1>{}, chunk);
        }
        chunk_body(std::false_type{}, std::integral_constant<int, 0>{}, n_chunks - 1);

        // ---- output transform, bias, ReLU (, MaxPool) and store: lane = output channel, register r = tile
        const int co = cb * kWinoTN + 32 * wn + li;
        const float bj = bias ? bias[co] : 0.f;
        const float floor_ = relu ? 0.f : -INFINITY;  // max(v, floor_): ReLU, or nothing
        const int Ho = POOL ? TH : H, Wo = POOL ? TW : W;
        constexpr int kPix = POOL ? 1 : 4;                       // output pixels per tile
        constexpr int kCcStride = 32 * kPix * 8 + 8;             // floats between channel chunks in the staging area (+ 8: banks)
        // Chunked output: the values go through LDS (the operand buffers are free once every wave has left the K loop), so
        // that a wave stores whole runs - 32 tiles of a tile row are 64 (pooled: 32) consecutive pixels x 8 channels = 2 KiB
        // contiguous per channel chunk and output row - with 16 bytes per lane instead of 4-byte stores 32 bytes at a time.
        float* const stage = wino_lds + wv * (4 * kCcStride);    // [chunk 4][i 2][tile 32][j 2][8] (pooled: [chunk 4][tile 32][8])
        // (measured, round 5: the staged form gains 7-10 us per POOLED layer; for the unpooled ones - four times the values -
        // the extra LDS pass and its registers cost 15 us more than the scattered stores: they keep storing from registers)
        const bool direct = out_nhwc || !POOL;
        if (!direct) __syncthreads();
        // The wave's 32 tiles lie in one tile row of the block (TC = 16: two, registers 0-7 and 8-15): image, tile row, validity
        // and the output row's address are formed once per strip, not once per tile (an integer division and 64-bit address
        // arithmetic per pixel made this loop cost 6 us per work item: 50 us of a 32 x 256 layer)
        constexpr int kStrips = TC >= 32 ? 1 : 2;
        const int ps = out_nhwc ? Cout : 8;   // floats between pixels
        float* e_base[kStrips];               // (b, first output row of the strip, pixel 0, co)
        bool e_ok[kStrips], e_row1[kStrips];  // strip exists; its second output row is inside the image
        int e_th[kStrips];
#pragma unroll
        for (int ss = 0; ss < kStrips; ++ss) {
            const int strip = TC >= 64 ? 0 : (TC == 32 ? wm : 2 * wm + ss);
            const int R = R0 + strip, b_ = R / TH, th_ = R - b_ * TH;
            e_ok[ss] = R < n_rows;
            e_row1[ss] = 2 * th_ + 1 < H;
            e_th[ss] = th_;
            e_base[ss] = y + (size_t)b_ * Ho * Wo * Cout + (out_nhwc ? (size_t)co : ((size_t)(co >> 3) * Ho * Wo) * 8 + (co & 7))
                         + (size_t)(POOL ? th_ : 2 * th_) * Wo * ps;
        }
        [[maybe_unused]] BnEpilogue bn = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hl;
            float sr[2][4];  // A^T M: rows (m0 + m1 + m2), (m1 - m2 - m3) per column nu
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                const float m0 = s.acc[nu][r], m1 = s.acc[4 + nu][r], m2 = s.acc[8 + nu][r], m3 = s.acc[12 + nu][r];
                sr[0][nu] = m0 + m1 + m2;
                sr[1][nu] = m1 - m2 - m3;
            }
            float o[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                o[i][0] = sr[i][0] + sr[i][1] + sr[i][2];
                o[i][1] = sr[i][1] - sr[i][2] - sr[i][3];
            }
            const int ss = kStrips == 2 ? (r >> 3) : 0;   // (registers 8-15 hold MFMA rows 16-31)
            const int tw_ = tc0 + ((32 * wm + row) & (TC - 1));
            const int ow = 2 * tw_;
            const bool col1 = ow + 1 < W;
            float pooled = 0.f;
            if constexpr (POOL) {
                pooled = o[0][0];  // (oh, ow) is inside whenever the tile exists; values outside the image never win
                if (col1) pooled = fmaxf(pooled, o[0][1]);
                if (e_row1[ss]) {
                    pooled = fmaxf(pooled, o[1][0]);
                    if (col1) pooled = fmaxf(pooled, o[1][1]);
                }
                pooled = fmaxf(pooled + bj, floor_);
            }
            if constexpr (BN && !POOL) {
                {   // the statistics of what is stored below: z itself (no bias, no ReLU here)
                    if (r == 0) bn.k = o[0][0];
                    const float in_tile = (e_ok[ss] && tw_ < TW) ? 1.f : 0.f, c1 = col1 ? in_tile : 0.f, r1 = e_row1[ss] ? 1.f : 0.f;
                    bn_epilogue_add(bn, o[0][0], in_tile);
                    bn_epilogue_add(bn, o[0][1], c1);
                    bn_epilogue_add(bn, o[1][0], in_tile * r1);
                    bn_epilogue_add(bn, o[1][1], c1 * r1);
                }
            }
            if (direct) {  // channels-last (the stack's last layer) or unpooled: straight from the registers
                if (e_ok[ss] && tw_ < TW) {
                    float* const yp = e_base[ss] + (unsigned)((POOL ? tw_ : ow) * ps);
                    if constexpr (POOL) {
                        yp[0] = pooled;
                    } else {
                        const unsigned rowp = (unsigned)(Wo * ps);
                        yp[0] = fmaxf(o[0][0] + bj, floor_);
                        if (col1) yp[ps] = fmaxf(o[0][1] + bj, floor_);
                        if (e_row1[ss]) {
                            yp[rowp] = fmaxf(o[1][0] + bj, floor_);
                            if (col1) yp[rowp + ps] = fmaxf(o[1][1] + bj, floor_);
                        }
                    }
                }
            } else {
                float* const sp = stage + (li >> 3) * kCcStride + (li & 7);
                if constexpr (POOL) {
                    sp[row * 8] = pooled;
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) sp[((i * 32 + row) * 2 + j) * 8] = fmaxf(o[i][j] + bj, floor_);
                }
            }
        }
        if constexpr (BN && !POOL) bn_epilogue_flush(bn, bn_sums, Cout, co, (int)(blockIdx.x % (unsigned)bn_slots(Cout)));
        if (!direct) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the wave reads back what its own lanes wrote: LDS operations
            __builtin_amdgcn_wave_barrier();                         // of a wave execute in order, this only pins the compiler
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // float4 q = 64 n + lane of a chunk's [i][tile][j][half] (pooled: [tile][half]) block: lane -> (tile, j, half), n -> i and
            // the upper tile bit.  The lane's two tiles (n even / odd) and where they land in y:
            constexpr int kPerChunk = 32 * kPix * 2;  // float4 per channel chunk
            constexpr int kN = kPerChunk / 64;        // store instructions per channel chunk (unpooled 4, pooled 1)
            const int half8 = lane & 1, jj = POOL ? 0 : (lane >> 1) & 1;
            size_t goff[POOL ? 1 : 2];
            bool gok[POOL ? 1 : 2][2];  // [tile slot][i]
#pragma unroll
            for (int ts = 0; ts < (POOL ? 1 : 2); ++ts) {
                const int tl = POOL ? (lane >> 1) : ((lane >> 2) + 16 * ts);
                const int t = 32 * wm + tl;
                const int R = R0 + t / TC, tw_ = tc0 + t % TC;
                const int b_ = R / TH, th_ = R - b_ * TH;
                const int oh = POOL ? th_ : 2 * th_, ow = (POOL ? tw_ : 2 * tw_) + jj;
                const bool tile_ok = R < n_rows && tw_ < TW && ow < Wo;
                goff[ts] = ((((size_t)b_ * (Cout / 8) + (cb * kWinoTN + 32 * wn) / 8) * Ho + oh) * Wo + ow) * 8 + 4 * half8;
                gok[ts][0] = tile_ok && oh < Ho;
                gok[ts][1] = tile_ok && oh + 1 < Ho;
            }
            const size_t cstride = (size_t)Ho * Wo * 8;  // floats between channel chunks of y
#pragma unroll
            for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                for (int n = 0; n < kN; ++n) {
                    const int i = POOL ? 0 : (n >> 1), ts = POOL ? 0 : (n & 1);
                    const float4 v = *reinterpret_cast<const float4*>(stage + cc * kCcStride + (n * 64 + lane) * 4);
                    if (gok[ts][i]) *reinterpret_cast<float4*>(y + goff[ts] + cc * cstride + (size_t)i * Wo * 8) = v;
                }
        }
    }
}

// Host side of the packing: weight [Cout][Cin][3][3] (OIHW, contiguous) -> U = G g G^T in the kernel's LDS order
// [cout block][chunk of 8 cin][pos 16][pair 2][hl 2][cout 64][2], where (pair, hl, j) <-> channel 4 pair + 2 hl + j of the chunk.
extern "C" size_t iris_wino_packed_len(int cin, int cout) { return (cin > 0 && cout > 0) ? wino_packed_floats(cin, cout) : 0; }

extern "C" int iris_wino_pack_weights(const float* weight_host, int cin, int cout, float* packed_host) {
    if (!weight_host || !packed_host) return fail(IRIS_E_INVALID, "iris_wino_pack_weights: NULL argument");
    if (cin <= 0 || cout <= 0 || (cin % kWinoKC) || (cout % kWinoTN))
        return fail(IRIS_E_UNSUPPORTED, "iris_wino_pack_weights: cin %d must be a multiple of %d, cout %d of %d", cin, kWinoKC, cout, kWinoTN);
    static const double G[4][3] = {{1.0, 0.0, 0.0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0.0, 0.0, 1.0}};
    const int n_chunks = cin / kWinoKC;
    for (int o = 0; o < cout; ++o)
        for (int c = 0; c < cin; ++c) {
            const float* g = weight_host + ((size_t)o * cin + c) * 9;
            double t[4][3];
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
            const int cb = o / kWinoTN, oc = o % kWinoTN, chunk = c / kWinoKC, k = c % kWinoKC;
            const int pair = k >> 2, hl = (k >> 1) & 1, j2 = k & 1;  // channel 4 pair + 2 hl + j2: step j2 of the pair, lanes hl
            for (int xi = 0; xi < 4; ++xi)
                for (int nu = 0; nu < 4; ++nu) {
                    const double v = t[xi][0] * G[nu][0] + t[xi][1] * G[nu][1] + t[xi][2] * G[nu][2];
                    const int pos = 4 * xi + nu;
                    packed_host[((size_t)cb * n_chunks + chunk) * kWinoUFloats + ((((size_t)pos * 2 + pair) * 2 + hl) * kWinoTN + oc) * 2 + j2] = (float)v;
                }
        }
    return IRIS_OK;
}

// The same packing on the DEVICE, from a weight tensor with arbitrary element strides (a channels_last parameter as it is):
// a training step re-packs its weights every step, inside the stream (and inside a captured hipGraph).  transposed != 0 packs
// the weights of the backward-data pass: dx = conv(dz, W') with W'[ci][co][i][j] = W[co][ci][2 - i][2 - j], i.e. `cin` counts
// the ORIGINAL output channels and `cout` the original input channels.
// One workgroup = one (cout block of 64, chunk of 8 cin) = 32 KiB of the packed tensor, assembled in LDS and written out in
// whole lines (a thread per weight writing its 16 values straight to memory touched a different line with every 4-byte store:
// 12 us per layer on average, 35 for 512 x 512 - nine times the traffic's worth).  Threads run along whichever of the two
// channel axes is the denser one in memory.
__device__ __forceinline__ void wino_pack_block(float* __restrict__ u /* LDS, kWinoUFloats */, int block, const float* __restrict__ w, long so,
                                                long si, long sh, long sw, int cin, int cout, int transposed, float* __restrict__ packed) {
    const int n_chunks = cin / kWinoKC;
    const int cb = block / n_chunks, chunk = block - cb * n_chunks;
    const long str_o = transposed ? si : so, str_c = transposed ? so : si;   // element strides along the packed cout / cin
    const int t = threadIdx.x;
    const int oc = str_c <= str_o ? t >> 3 : t & 63, k = str_c <= str_o ? t & 7 : t >> 6;
    const float* const src = w + (long)(cb * kWinoTN + oc) * str_o + (long)(chunk * kWinoKC + k) * str_c;
    float g[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) g[i][j] = transposed ? src[(long)(2 - i) * sh + (long)(2 - j) * sw] : src[(long)i * sh + (long)j * sw];
    // G g: rows (g0), (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, (g2)
    float tr[4][3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        tr[0][j] = g[0][j];
        tr[1][j] = 0.5f * (g[0][j] + g[1][j] + g[2][j]);
        tr[2][j] = 0.5f * (g[0][j] - g[1][j] + g[2][j]);
        tr[3][j] = g[2][j];
    }
    const int pair = k >> 2, hl = (k >> 1) & 1, j2 = k & 1;
    float* const dst = u + ((pair * 2 + hl) * kWinoTN + oc) * 2 + j2;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        const float v[4] = {tr[xi][0], 0.5f * (tr[xi][0] + tr[xi][1] + tr[xi][2]), 0.5f * (tr[xi][0] - tr[xi][1] + tr[xi][2]), tr[xi][2]};
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) dst[(4 * xi + nu) * (4 * kWinoTN * 2)] = v[nu];
    }
    __syncthreads();
    float4* const out = reinterpret_cast<float4*>(packed + ((size_t)cb * n_chunks + chunk) * kWinoUFloats);
#pragma unroll
    for (int q = 0; q < kWinoUFloats / 4 / 512; ++q) out[q * 512 + t] = reinterpret_cast<const float4*>(u)[q * 512 + t];
}

__global__ __launch_bounds__(512) void k_wino_pack(const float* __restrict__ w, long so, long si, long sh, long sw, int cin, int cout,
                                                   int transposed, float* __restrict__ packed) {
    __shared__ __attribute__((aligned(16))) float u[kWinoUFloats];
    wino_pack_block(u, (int)blockIdx.x, w, so, si, sh, sw, cin, cout, transposed, packed);
}

// All the packings of a training step in ONE launch (round 6): 23 launches of ~6.7 us (12 forward + 11 backward-data layers, each a
// few dozen workgroups) were 0.16 ms of a 10 ms step, most of it launch latency and tails.  The jobs travel BY VALUE in the kernel
// arguments (<= 4 KiB: 48 jobs), so the launch needs no device-side table and is captured into a hipGraph like any other.
struct WinoPackJobs {
    int n, pad;
    iris_pack_job j[kWinoPackMaxJobs];
};
__global__ __launch_bounds__(512) void k_wino_pack_multi(const WinoPackJobs jobs) {
    __shared__ __attribute__((aligned(16))) float u[kWinoUFloats];
    int k = 0;
    while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].first_block) ++k;   // uniform
    const iris_pack_job& jb = jobs.j[k];
    wino_pack_block(u, (int)blockIdx.x - jb.first_block, jb.weight, jb.stride_o, jb.stride_i, jb.stride_h, jb.stride_w, jb.cin, jb.cout,
                    jb.transposed, jb.packed);
}

extern "C" int iris_wino_pack_weights_device(const float* weight, long stride_o, long stride_i, long stride_h, long stride_w, int cin,
                                             int cout, int transposed, float* packed, void* stream) {
    if (!weight || !packed) return fail(IRIS_E_INVALID, "iris_wino_pack_weights_device: NULL argument");
    if (cin <= 0 || cout <= 0 || (cin % kWinoKC) || (cout % kWinoTN))
        return fail(IRIS_E_UNSUPPORTED, "iris_wino_pack_weights_device: cin %d must be a multiple of %d, cout %d of %d", cin, kWinoKC, cout, kWinoTN);
    if (reinterpret_cast<uintptr_t>(packed) & 15) return fail(IRIS_E_INVALID, "iris_wino_pack_weights_device: packed must be 16-byte aligned");
    k_wino_pack<<<(unsigned)((cout / kWinoTN) * (cin / kWinoKC)), 512, 0, (hipStream_t)stream>>>(weight, stride_o, stride_i, stride_h, stride_w,
                                                                                                 cin, cout, transposed, packed);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

template <bool POOL, bool IN_NHWC, bool BN = false>
static hipError_t wino_launch(int tc, unsigned grid, hipStream_t s, const float* x, const float* packed, const float* bias,
                              float* y, int batch, int height, int width, int cin, int cout, int out_nhwc, int relu, double* bn_sums) {
    if (tc >= 64) k_conv3x3_wino<POOL, 64, IN_NHWC, BN><<<grid, 256, kWinoLdsBytes, s>>>(x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, bn_sums);
    else if (tc >= 32) k_conv3x3_wino<POOL, 32, IN_NHWC, BN><<<grid, 256, kWinoLdsBytes, s>>>(x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, bn_sums);
    else k_conv3x3_wino<POOL, 16, IN_NHWC, BN><<<grid, 256, kWinoLdsBytes, s>>>(x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, bn_sums);
    return hipGetLastError();
}

template <bool POOL, bool IN_NHWC, bool BN = false>
static hipError_t wino_set_lds_limit() {
    const void* ks[3] = {(const void*)k_conv3x3_wino<POOL, 64, IN_NHWC, BN>, (const void*)k_conv3x3_wino<POOL, 32, IN_NHWC, BN>,
                         (const void*)k_conv3x3_wino<POOL, 16, IN_NHWC, BN>};
    for (const void* k : ks) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kWinoLdsBytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// x: channel-chunked [B][cin / 8][H][W][8] (IRIS_WINO_IN_NHWC: channels-last [B][H][W][cin]); packed: iris_wino_pack_weights[_device];
// bias: nullable; y: chunked [B][cout / 8][Ho][Wo][8] (IRIS_WINO_OUT_NHWC: channels-last [B][Ho][Wo][cout]); flags = IRIS_WINO_*
static int conv3x3_wino_impl(const float* x, const float* packed, const float* bias, float* y, int batch, int height, int width,
                             int cin, int cout, int flags, double* bn_sums, void* stream) {
    if (!x || !packed || !y) return fail(IRIS_E_INVALID, "iris_conv3x3_wino: NULL argument");
    if (bn_sums && (bias || (flags & (IRIS_WINO_POOL | IRIS_WINO_RELU)) || !(flags & IRIS_WINO_OUT_NHWC)))
        return fail(IRIS_E_INVALID, "iris_conv3x3_wino_bn: the statistics are those of the bare convolution, channels-last out (no bias / ReLU / pooling)");
    if (batch <= 0 || height <= 0 || width <= 0) return fail(IRIS_E_INVALID, "iris_conv3x3_wino: empty tensor");
    if (flags & ~(IRIS_WINO_POOL | IRIS_WINO_OUT_NHWC | IRIS_WINO_IN_NHWC | IRIS_WINO_RELU)) return fail(IRIS_E_INVALID, "iris_conv3x3_wino: flags 0x%x", flags);
    if (cin <= 0 || cout <= 0 || (cin % kWinoKC) || (cout % kWinoTN))
        return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_wino: cin %d must be a multiple of %d, cout %d of %d", cin, kWinoKC, cout, kWinoTN);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) & 15)
        return fail(IRIS_E_INVALID, "iris_conv3x3_wino: x and the packed weights must be 16-byte aligned");
    if ((long long)batch * height * width * cin >= 1073741824LL)
        return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_wino: tensor too large for 32-bit byte offsets (>= 2^30 elements)");
    int dev = 0, n_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    static std::atomic<unsigned> attr_set[64];
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        HIP_TRY((wino_set_lds_limit<false, false>()));
        HIP_TRY((wino_set_lds_limit<false, true>()));
        HIP_TRY((wino_set_lds_limit<true, false>()));
        HIP_TRY((wino_set_lds_limit<true, true>()));
        HIP_TRY((wino_set_lds_limit<false, false, true>()));
        HIP_TRY((wino_set_lds_limit<false, true, true>()));
        if (dev >= 0 && dev < 64) attr_set[dev].store(1u, std::memory_order_release);
    }
    const int pool = (flags & IRIS_WINO_POOL) != 0, out_nhwc = (flags & IRIS_WINO_OUT_NHWC) != 0;
    const int in_nhwc = (flags & IRIS_WINO_IN_NHWC) != 0, relu = (flags & IRIS_WINO_RELU) != 0;
    const int th = (height + 1) / 2, tw = (width + 1) / 2;
    const int tc = tw > 32 ? 64 : (tw > 16 ? 32 : 16), tr = kWinoTM / tc;
    const long long n_work = (((long long)batch * th + tr - 1) / tr) * ((tw + tc - 1) / tc) * (cout / kWinoTN);
    if (n_work >= 2147483647LL) return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_wino: too many tiles");
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
    const unsigned grid = (unsigned)std::min<long long>(n_work, n_cu);  // persistent: one workgroup (4 waves, 156 KiB of LDS) per CU
    const hipStream_t st = (hipStream_t)stream;
    hipError_t e;
    if (pool) e = in_nhwc ? wino_launch<true, true>(tc, grid, st, x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, nullptr)
                          : wino_launch<true, false>(tc, grid, st, x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, nullptr);
    else if (bn_sums) e = in_nhwc ? wino_launch<false, true, true>(tc, grid, st, x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, bn_sums)
                                  : wino_launch<false, false, true>(tc, grid, st, x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, bn_sums);
    else e = in_nhwc ? wino_launch<false, true>(tc, grid, st, x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, nullptr)
                     : wino_launch<false, false>(tc, grid, st, x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, nullptr);
    HIP_TRY(e);
    return IRIS_OK;
}

extern "C" int iris_conv3x3_wino(const float* x, const float* packed, const float* bias, float* y, int batch, int height, int width,
                                 int cin, int cout, int flags, void* stream) {
    return conv3x3_wino_impl(x, packed, bias, y, batch, height, width, cin, cout, flags, nullptr, stream);
}
// The training form (bare convolution, flags = IRIS_WINO_OUT_NHWC [| IRIS_WINO_IN_NHWC]) that ALSO accumulates the statistics of
// the BatchNorm behind it: bn_sums_zeroed = DEVICE double [iris_bn_sums_len(cout)], zero on entry, consumed by
// iris_bn_relu_apply_sums0 / iris_bn_relu_pool_apply_sums0 (no iris_bn_stats pass over z)
extern "C" int iris_conv3x3_wino_bn(const float* x, const float* packed, float* y, int batch, int height, int width, int cin, int cout,
                                    int flags, double* bn_sums_zeroed, void* stream) {
    if (!bn_sums_zeroed) return fail(IRIS_E_INVALID, "iris_conv3x3_wino_bn: NULL argument");
    return conv3x3_wino_impl(x, packed, nullptr, y, batch, height, width, cin, cout, flags, bn_sums_zeroed, stream);
}
