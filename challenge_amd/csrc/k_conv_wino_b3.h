// k_conv_wino_b3.h -- the Winograd F(2x2, 3x3) convolution of k_conv_wino.h with its 16 GEMMs on the BF16 matrix cores at fp32
// accuracy: both operands split into THREE bf16 terms, six of the nine partial products accumulated in fp32 (round 6).
// Part of the single translation unit iris_frontend.hip (and of scripts/microbench/wino_conv.hip, which builds it alone).
#pragma once
// ---------------------------------------------------------------------------
// Why.  v_mfma_f32_32x32x2_f32 runs at 64 FLOP / clock / SIMD - 1/16 of v_mfma_f32_32x32x16_bf16 (1,024) - and every convolution
// kernel of this library is bound by that rate (MI355X_MICROARCH.md, Matrix cores).  An fp32 value is the EXACT sum of three
// bf16 values (24 significant bits = 3 x 8: hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid), every subtraction exact),
// so   a b = sum_{i, j} a_i b_j   and the six products with i + j <= 4 (a1 b1, a1 b2, a2 b1, a1 b3, a2 b2, a3 b1) carry everything
// down to 2^-24 |a b|: the three dropped ones are <= 2^-25 |a b| together.  Each bf16 x bf16 product is exact in fp32, the
// accumulation is the matrix core's fp32 one.  Six bf16 MFMAs of K = 16 (32 cycles each) replace eight fp32 MFMAs of K = 2
// (64 cycles each): 192 instead of 512 matrix-pipe cycles per 16 input channels - the BOUND moves by 2.67x.
// What it costs: the split itself (U once per layer at packing time; V on the vector ALU, 5.5 instructions per value), and 1.5x
// the operand bytes.
// Decomposition (differs from k_conv_wino.h, whose waves each own all 16 positions of 32 tiles x 32 channels):
//   workgroup = 4 waves = 64 tiles x 64 output channels (as there), ONE wave per SIMD, persistent;
//   wave xi owns ROW xi of the 4 x 4 position grid - 4 positions x (2 tile blocks x 2 channel blocks of 32 x 32) = 16
//   accumulators = 256 registers - and is its own producer: it reads the two patch rows B^T's row xi combines straight from the
//   staged input (LDS), transforms and splits them in registers IN THE MFMA OPERAND LAYOUT (lane = tile, 8 consecutive input
//   channels per lane) and never writes V anywhere: no V buffer, no barrier between transform and GEMM, LDS is read-only in the
//   K loop (LDS stores are the expensive direction on this chip: ~80 B / clock / CU against 256 for reads).
//   Every A operand meets two B operands and vice versa (2 x 2 register blocking): 1 operand fetch per MFMA instead of 2.
//   U (the split weights, bf16) is read by each wave straight from global memory / L2 into registers in operand order - a
//   position's U is used by exactly one wave, so LDS would add nothing.
//   K runs over the input channels in chunks of 16 (one MFMA K-step); the staged input is triple-buffered, requested two
//   chunks ahead by LDS-DMA, one workgroup barrier per chunk.
//   Output transform Y = A^T M A: a wave holds only row xi of M, so it applies the COLUMN half in registers (2 values per tile
//   and channel from its 4 positions) and the four waves exchange those through LDS (128 KiB per 64 x 64 block, once per work
//   item); wave w then finishes block w (tile block w & 1, channel block w >> 1): row half, bias, ReLU, MaxPool, store.
// Layouts: x channel-chunked [B][Cin / 8][H][W][8] or channels-last (IRIS_WINO_IN_NHWC); y chunked or channels-last as in
// k_conv_wino.h.  Cin % 16 == 0, Cout % 64 == 0.
// ---------------------------------------------------------------------------
typedef __bf16 b3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned b3_u32x4 __attribute__((ext_vector_type(4)));
#ifndef IRIS_WINO_STANDALONE
// (f32x16 comes from common.h in the library build)
#endif

// timing experiments only (results wrong when non-zero): 1 no U loads / waits in the K loop, 2 no preparation (row stage, column
// stage, split), 4 no MFMAs, 8 no barrier / LDS-DMA in the K loop
#ifndef IRIS_B3_ABLATE
#define IRIS_B3_ABLATE 0
#endif
#define B3_ABL(bit) ((IRIS_B3_ABLATE & (bit)) != 0)
#ifdef IRIS_B3_FREE_SCHED   // experiment: no per-slot scheduling barriers (the compiler's own interleaving)
#define B3_SLOT_FENCE() do { } while (0)
#else
#define B3_SLOT_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
constexpr int kB3KC = 16;                       // input channels per chunk = one K-step of v_mfma_f32_32x32x16_bf16
constexpr int kB3Rows = 4 * 4;                  // staged pixel rows per chunk at most: 4 per strip, up to 4 strips (TC = 16)
constexpr int kB3BufSlots = 2304;               // 16-byte pieces per chunk buffer (>= 16 rows x 4 quarters x 2 parities x 17; 36 KiB)
constexpr int kB3Bufs = 3;
constexpr size_t kB3ExchangeBytes = 4 * 2 * 4 * 16 * 64 * sizeof(float);   // [xi][j][block][r][lane]
constexpr size_t kB3LdsBytes = kB3ExchangeBytes > (size_t)kB3Bufs * kB3BufSlots * 16 ? kB3ExchangeBytes : (size_t)kB3Bufs * kB3BufSlots * 16;

// bytes of the packed, split weights: [cout block 64][chunk 16][pos 16][term 3][channel block 2][lane 64] x 16 bytes
static size_t wino_b3_packed_bytes(int cin, int cout) { return (size_t)96 * cin * cout; }

// two fp32 -> packed bf16 pair (round to nearest even): one instruction
__device__ __forceinline__ unsigned b3_cvt_pk(float lo, float hi) {
    unsigned r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}
// (x0, x1) -> three packed bf16 pairs, x_k = t0_k + t1_k + t2_k exactly (11 vector instructions per pair)
__device__ __forceinline__ void b3_split_pair(float x0, float x1, unsigned& t0, unsigned& t1, unsigned& t2) {
    t0 = b3_cvt_pk(x0, x1);
    const float r0 = x0 - __uint_as_float(t0 << 16), r1 = x1 - __uint_as_float(t0 & 0xffff0000u);
    t1 = b3_cvt_pk(r0, r1);
    const float s0 = r0 - __uint_as_float(t1 << 16), s1 = r1 - __uint_as_float(t1 & 0xffff0000u);
    t2 = b3_cvt_pk(s0, s1);
}

// 16 bytes global -> registers, invisible to the compiler's wait-count bookkeeping (it cannot see the LDS-DMA requests issued
// beside these loads, so its own vmcnt would drain a DMA issued a moment ago); B3_WAIT_U pairs with it
// address = uniform base (SGPR pair) + per-lane byte offset (VGPR, 32 bit) + IMM (0 .. 4095): no vector ALU per load
template <int IMM>
__device__ __forceinline__ b3_u32x4 b3_gload16(const uint4* base /*uniform*/, unsigned voff) {
    b3_u32x4 v;
    const uint64_t a = reinterpret_cast<uint64_t>(base);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    const uint4* sb = reinterpret_cast<const uint4*>(((uint64_t)hi << 32) | lo);
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(v) : "v"(voff), "s"(sb), "n"(IMM) : "memory");
    return v;
}
// wait until at most N vector-memory operations of this wave are outstanding; U (the six operand registers of a position) is
// tied to the wait so that no consumer can be scheduled in front of it
#define B3_WAIT_U(N, U)                                                                                                            \
    asm volatile("s_waitcnt vmcnt(%6)" : "+v"(U[0][0]), "+v"(U[0][1]), "+v"(U[0][2]), "+v"(U[1][0]), "+v"(U[1][1]), "+v"(U[1][2]) \
                 : "n"(N) : "memory")

// ---- packing: weight [Cout][Cin][3][3] (any strides) -> U = G g G^T, split, in the kernel's operand order ---------------------
// thread = (channel block nb, lane): output channel 64 cb + 32 nb + (lane & 31), input channels 16 chunk + 8 (lane >> 5) + j
__device__ __forceinline__ void wino_pack_b3_block(int block, const float* __restrict__ w, long so, long si, long sh, long sw, int cin, int cout,
                                                   int transposed, uint4* __restrict__ packed) {
    const int n_chunks = cin / kB3KC;
    const int cb = block / n_chunks, chunk = block - cb * n_chunks;
    const int nb = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long str_o = transposed ? si : so, str_c = transposed ? so : si;
    const int oc = cb * 64 + nb * 32 + (lane & 31);
    float u[16][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float* const src = w + (long)oc * str_o + (long)(chunk * kB3KC + 8 * (lane >> 5) + j) * str_c;
        float g[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) g[a][b] = transposed ? src[(long)(2 - a) * sh + (long)(2 - b) * sw] : src[(long)a * sh + (long)b * sw];
        float tr[4][3];  // G g: rows (g0), (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, (g2) - the arithmetic of k_wino_pack
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            tr[0][b] = g[0][b];
            tr[1][b] = 0.5f * (g[0][b] + g[1][b] + g[2][b]);
            tr[2][b] = 0.5f * (g[0][b] - g[1][b] + g[2][b]);
            tr[3][b] = g[2][b];
        }
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) {
            u[4 * xi + 0][j] = tr[xi][0];
            u[4 * xi + 1][j] = 0.5f * (tr[xi][0] + tr[xi][1] + tr[xi][2]);
            u[4 * xi + 2][j] = 0.5f * (tr[xi][0] - tr[xi][1] + tr[xi][2]);
            u[4 * xi + 3][j] = tr[xi][2];
        }
    }
    uint4* const out = packed + ((size_t)(cb * n_chunks + chunk) * 16 * 3 * 2 + nb) * 64 + lane;
#pragma unroll
    for (int pos = 0; pos < 16; ++pos) {
        unsigned t[3][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) b3_split_pair(u[pos][2 * k], u[pos][2 * k + 1], t[0][k], t[1][k], t[2][k]);
#pragma unroll
        for (int term = 0; term < 3; ++term) out[(size_t)(pos * 3 + term) * 2 * 64] = make_uint4(t[term][0], t[term][1], t[term][2], t[term][3]);
    }
}

__global__ __launch_bounds__(128) void k_wino_pack_b3(const float* __restrict__ w, long so, long si, long sh, long sw, int cin, int cout,
                                                      int transposed, uint4* __restrict__ packed) {
    wino_pack_b3_block((int)blockIdx.x, w, so, si, sh, sw, cin, cout, transposed, packed);
}
// every packing of a training step in one launch (k_conv_wino.h: k_wino_pack_multi)
__global__ __launch_bounds__(128) void k_wino_pack_b3_multi(const WinoPackJobs jobs) {
    int k = 0;
    while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].first_block) ++k;   // uniform
    const iris_pack_job& jb = jobs.j[k];
    wino_pack_b3_block((int)blockIdx.x - jb.first_block, jb.weight, jb.stride_o, jb.stride_i, jb.stride_h, jb.stride_w, jb.cin, jb.cout,
                       jb.transposed, reinterpret_cast<uint4*>(jb.packed));
}

extern "C" int iris_wino_pack_weights_device_multi(iris_pack_job* jobs_host, int n_jobs, int split_bf16, void* stream) {
    if (!jobs_host || n_jobs <= 0) return fail(IRIS_E_INVALID, "iris_wino_pack_weights_device_multi: no jobs");
    const int kc = split_bf16 ? kB3KC : kWinoKC;
    for (int i = 0; i < n_jobs; ++i) {
        const iris_pack_job& jb = jobs_host[i];
        if (!jb.weight || !jb.packed) return fail(IRIS_E_INVALID, "iris_wino_pack_weights_device_multi: job %d: NULL pointer", i);
        if (jb.cin <= 0 || jb.cout <= 0 || (jb.cin % kc) || (jb.cout % 64))
            return fail(IRIS_E_UNSUPPORTED, "iris_wino_pack_weights_device_multi: job %d: cin %d must be a multiple of %d, cout %d of 64", i, jb.cin, kc, jb.cout);
        if (reinterpret_cast<uintptr_t>(jb.packed) & 15) return fail(IRIS_E_INVALID, "iris_wino_pack_weights_device_multi: job %d: packed must be 16-byte aligned", i);
    }
    for (int base = 0; base < n_jobs; base += kWinoPackMaxJobs) {
        WinoPackJobs jobs;
        jobs.n = std::min(kWinoPackMaxJobs, n_jobs - base);
        jobs.pad = 0;
        long long blocks = 0;
        for (int i = 0; i < jobs.n; ++i) {
            iris_pack_job& jb = jobs_host[base + i];
            jb.first_block = (int)blocks;
            jobs.j[i] = jb;
            blocks += (long long)(jb.cout / 64) * (jb.cin / kc);
        }
        if (blocks >= 2147483647LL) return fail(IRIS_E_UNSUPPORTED, "iris_wino_pack_weights_device_multi: too many blocks");
        if (split_bf16) k_wino_pack_b3_multi<<<(unsigned)blocks, 128, 0, (hipStream_t)stream>>>(jobs);
        else k_wino_pack_multi<<<(unsigned)blocks, 512, 0, (hipStream_t)stream>>>(jobs);
        HIP_TRY(hipGetLastError());
    }
    return IRIS_OK;
}

extern "C" size_t iris_wino_b3_packed_len(int cin, int cout) { return (cin > 0 && cout > 0) ? wino_b3_packed_bytes(cin, cout) / 4 : 0; }  // in floats

extern "C" int iris_wino_b3_pack_weights_device(const float* weight, long stride_o, long stride_i, long stride_h, long stride_w, int cin,
                                                int cout, int transposed, float* packed, void* stream) {
    if (!weight || !packed) return fail(IRIS_E_INVALID, "iris_wino_b3_pack_weights_device: NULL argument");
    if (cin <= 0 || cout <= 0 || (cin % kB3KC) || (cout % 64))
        return fail(IRIS_E_UNSUPPORTED, "iris_wino_b3_pack_weights_device: cin %d must be a multiple of %d, cout %d of 64", cin, kB3KC, cout);
    if (reinterpret_cast<uintptr_t>(packed) & 15) return fail(IRIS_E_INVALID, "iris_wino_b3_pack_weights_device: packed must be 16-byte aligned");
    k_wino_pack_b3<<<(unsigned)((cout / 64) * (cin / kB3KC)), 128, 0, (hipStream_t)stream>>>(weight, stride_o, stride_i, stride_h, stride_w, cin,
                                                                                             cout, transposed, reinterpret_cast<uint4*>(packed));
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

#ifdef IRIS_B3_STAMPS
// timing experiments only: wave 0 of every workgroup records (s_memrealtime [100 MHz], s_memtime [shader clock]) at six points of its
// FIRST two work items into this buffer: [workgroup][item 2][point 10][2]
__device__ unsigned long long g_b3_stamps[256 * 2 * 10 * 2];
extern "C" int iris_b3_read_stamps(unsigned long long* host) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_b3_stamps), sizeof(g_b3_stamps));
}
#define B3_STAMP(item, point)                                                                                  \
    do {                                                                                                       \
        if (tid == 0 && (item) < 2 && blockIdx.x < 256) {                                                      \
            g_b3_stamps[((blockIdx.x * 2 + (item)) * 10 + (point)) * 2 + 0] = __builtin_amdgcn_s_memrealtime(); \
            g_b3_stamps[((blockIdx.x * 2 + (item)) * 10 + (point)) * 2 + 1] = __builtin_amdgcn_s_memtime();     \
        }                                                                                                      \
    } while (0)
#else
#define B3_STAMP(item, point) do { } while (0)
#endif

// ---- the convolution --------------------------------------------------------------------------------------------------------
// BN: also accumulate the statistics of the BatchNorm behind the convolution (training form only).  A TEMPLATE parameter, not a test of
// the pointer: with the test in the kernel the build for bn_sums == nullptr faulted on the 32-tile-column geometry (the code with the
// branch compiled out did not: profiles/r6/b3_bn_runtime_branch_fault.log) - the instantiation without BN is the code that was measured
template <bool POOL, int TC, bool IN_NHWC, bool BN = false>
__global__ __launch_bounds__(256, 1) void k_conv3x3_wino_b3(const float* __restrict__ x, const uint4* __restrict__ u3,
                                                            const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
                                                            int Cin, int Cout, int out_nhwc, int relu, double* __restrict__ bn_sums) {
    extern __shared__ __attribute__((aligned(16))) float b3_lds[];
    constexpr int TR = 64 / TC, PW = 2 * TC + 2, PH = TC + 1, ROWS = 4 * TR;
    constexpr int kSlots = ROWS * 4 * 2 * PH;          // 16-byte pieces of a chunk: [row][quarter q][pixel parity][pixel / 2]
    constexpr int kDma = (kSlots + 255) / 256;         // LDS-DMA instructions per wave and chunk
    static_assert(kSlots <= kB3BufSlots, "chunk buffer too small");
    const int tid = threadIdx.x, lane = tid & 63, xi = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hl = lane >> 5, li = lane & 31;
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1;
    const int n_rows = B * TH;
    const int row_blocks = (n_rows + TR - 1) / TR, col_blocks = (TW + TC - 1) / TC;
    const int cout_blocks = Cout / 64, n_chunks = Cin / kB3KC;
    const int n_work = row_blocks * col_blocks * cout_blocks;
    const size_t plane = (size_t)H * W * 8;            // floats of one 8-channel plane of one image (chunked layout)
    const unsigned lds0 = wino_lds_addr(b3_lds);
    // B^T row xi combines patch rows (ra, rb): xi 0: d0 - d2, 1: d1 + d2, 2: d2 - d1, 3: d1 - d3
    const int ra = xi == 0 ? 0 : (xi == 2 ? 2 : 1), rb = xi == 0 ? 2 : (xi == 1 ? 2 : (xi == 2 ? 1 : 3));
    const float row_sign = xi == 1 ? 1.f : -1.f;

    int item = 0;
    for (int work = blockIdx.x; work < n_work; work += gridDim.x, ++item) {
        const int cb = work % cout_blocks, blk = work / cout_blocks;
        const int cbk = blk % col_blocks, rbk = blk / col_blocks;
        const int R0 = rbk * TR, tc0 = cbk * TC;
        B3_STAMP(item, 0);
        // ---- this lane's LDS-DMA pieces: slot i = 64 (4 k + xi) + lane of a chunk buffer, the same for every chunk -------------
        // per strip (uniform): image, tile row, does the strip exist - ONE integer division per strip and work item
        int s_b[TR], s_th[TR];
        bool s_ok[TR];
#pragma unroll
        for (int st = 0; st < TR; ++st) {
            const int R = R0 + st;
            s_b[st] = R / TH;
            s_th[st] = R - s_b[st] * TH;
            s_ok[st] = R < n_rows;
        }
        unsigned poff[kDma];
        unsigned long long pmask[kDma];
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        __syncthreads();   // every wave has left the previous work item's exchange
#pragma unroll
        for (int k = 0; k < kDma; ++k) {
            // slot i = 64 (4 k + xi) + lane of a chunk buffer -> (pixel column, patch row, strip, channel quarter): re-derived per
            // work item from an OPAQUE lane id - the first version let the compiler hoist the decode out of the work loop, where the
            // K loop's register demand spilled it: 119 scratch reloads, each waited for, = 16 us per work item
            const int i = 64 * (4 * k + xi) + lane_o;
            const int pxh = i % PH, t1 = i / PH, par = t1 & 1, q = (t1 >> 1) & 3, r = t1 >> 3;
            const int px = 2 * pxh + par, pr = r & 3, strip = (r >> 2) & 3;
            const unsigned d = i < kSlots ? 1u << 14 : 0u;
            int b_ = s_b[0], th = s_th[0];
            bool sok = s_ok[0];
#pragma unroll
            for (int st = 1; st < TR; ++st)
                if (strip == st) b_ = s_b[st], th = s_th[st], sok = s_ok[st];
            const int hh = 2 * th - 1 + pr, ww = 2 * tc0 - 1 + px;
            const bool ok = (d >> 14) && sok && hh >= 0 && hh < H && ww >= 0 && ww < W;
            // channel 4 q + {0..3} of the chunk's 16: plane q >> 1, floats 4 (q & 1) .. + 3 of the pixel's 8 (32-bit arithmetic:
            // the host side keeps the tensor below 2^30 elements)
            const unsigned pix = (unsigned)hh * (unsigned)W + (unsigned)ww;
            poff[k] = !ok ? 0u : IN_NHWC ? (((unsigned)b_ * (unsigned)H * (unsigned)W + pix) * (unsigned)Cin + 4u * q) * 4u
                                         : ((((unsigned)b_ * (unsigned)(Cin / 8) + (unsigned)(q >> 1)) * (unsigned)H * (unsigned)W + pix) * 8u + 4u * (q & 1)) * 4u;
            pmask[k] = __ballot(ok);
            if (!ok && i < kB3BufSlots) {   // out-of-image pixels: zero in all three buffers, never requested
#pragma unroll
                for (int bf = 0; bf < kB3Bufs; ++bf) reinterpret_cast<float4*>(b3_lds)[bf * kB3BufSlots + i] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        auto dma_into = [&](int bufi, int chunk) {   // this wave's pieces of `chunk` into chunk buffer `bufi`
            const float* xc = x + (size_t)chunk * (IN_NHWC ? (size_t)kB3KC : 2 * plane);   // uniform
            const unsigned base = lds0 + (unsigned)(bufi * kB3BufSlots * 16);
#pragma unroll
            for (int k = 0; k < kDma; ++k) wino_dma16_gather(xc, poff[k], pmask[k], base + 1024u * (4 * k + xi));
        };
        // this lane's tiles: block tb -> (strip, tile column)
        int t_strip[2], t_col[2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            t_strip[tb] = TC >= 64 ? 0 : (TC == 32 ? tb : 2 * tb + (li >> 4));
            t_col[tb] = TC >= 64 ? 32 * tb + li : (TC == 32 ? li : (li & 15));
        }
        // ---- this wave's U: positions 4 xi + nu
        const uint4* const ubase = u3 + ((size_t)cb * n_chunks * 16 + 4 * xi) * 3 * 2 * 64;   // uniform
        const unsigned uoff0 = (unsigned)lane * 16u, uoff1 = uoff0 + 4096u;
        auto load_u = [&](b3_u32x4 (&dst)[2][3], int chunk, int nu) {   // six rows of 1 KiB: (term, channel block)
            const uint4* p = ubase + ((size_t)chunk * 16 + nu) * 3 * 2 * 64;
            dst[0][0] = b3_gload16<0>(p, uoff0);
            dst[1][0] = b3_gload16<1024>(p, uoff0);
            dst[0][1] = b3_gload16<2048>(p, uoff0);
            dst[1][1] = b3_gload16<3072>(p, uoff0);
            dst[0][2] = b3_gload16<0>(p, uoff1);
            dst[1][2] = b3_gload16<1024>(p, uoff1);
        };

        f32x16 acc[4][2][2];
#pragma unroll
        for (int nu = 0; nu < 4; ++nu)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[nu][tb][nb][r] = 0.f;

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the zeros are in place before a request can land beside them
        __syncthreads();
        B3_STAMP(item, 6);
        // ================= the K loop: a software pipeline written out as SLOTS ==================================================
        // One wave per SIMD: nothing but this wave's own instruction order hides anything.  A chunk (16 input channels) is four
        // BLOCKS, one per position of this wave's row in the order nu = 0, 3, 1, 2; a block is 24 slots, each ONE MFMA of the
        // current position (operands av[cur], ub[cur]) followed by its share of the NEXT position's preparation - column stage and
        // three-term split into av[cur ^ 1], 13 instructions per channel pair - fenced by a scheduling barrier:
        //   block 0 (nu 0): + row stage of this chunk's patch columns 1, 3 (16 LDS reads, 32 FMAs), prepares nu 3 (columns 1, 3)
        //   block 1 (nu 3): prepares nu 1 (columns 1, 2)          block 2 (nu 1): prepares nu 2 (columns 1, 2)
        //   block 3 (nu 2): the workgroup barrier for chunk + 1's staged input, the LDS-DMA request for chunk + 2, the row stage
        //                   of chunk + 1's columns 0, 2 (w of this chunk is dead by now) and nu 0 of chunk + 1
        // U of the next position is requested at the top of every block, one block (~0.4 us) ahead.  Vector-memory requests
        // complete in order, every count below is static: past the last chunk the requests repeat the last chunk (valid memory,
        // unused results) instead of being skipped.
        float w[2][4][8];
        unsigned av[2][2][3][4];
        b3_u32x4 ub[2][2][3];   // [buffer][channel block][term]
        float raw[4][4];        // [(column a, row ra), (a, rb), (column b, ra), (b, rb)][4 channels]: one group's reads
        float pv0 = 0.f, pv1 = 0.f, pf0 = 0.f, pf1 = 0.f, pr0 = 0.f, pr1 = 0.f;
        // float4 index of (row ra / rb, this lane's quarter pair, tile column) in a chunk buffer, per tile block
        unsigned la[2], lb[2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            la[tb] = (unsigned)((((4 * t_strip[tb] + ra) * 4 + 2 * hl) * 2) * PH + t_col[tb]);
            lb[tb] = (unsigned)((((4 * t_strip[tb] + rb) * 4 + 2 * hl) * 2) * PH + t_col[tb]);
        }
        const float4* const lds4 = reinterpret_cast<const float4*>(b3_lds);
        auto col_value = [&](int nu, int tb, int ch) {
            return nu == 0 ? w[tb][0][ch] - w[tb][2][ch] : nu == 1 ? w[tb][1][ch] + w[tb][2][ch]
                 : nu == 2 ? w[tb][2][ch] - w[tb][1][ch] : w[tb][1][ch] - w[tb][3][ch];
        };
        // one of the 13 instructions that turn channels (2 k, 2 k + 1) of position `nu` into three packed bf16 pairs
        auto unit_op = [&](int dst, int nu, int tb, int k, int o) {
            unsigned(&t)[3][4] = av[dst][tb];
            if (o == 0) pv0 = col_value(nu, tb, 2 * k);
            else if (o == 1) pv1 = col_value(nu, tb, 2 * k + 1);
            else if (o == 2) t[0][k] = b3_cvt_pk(pv0, pv1);
            else if (o == 3) pf0 = __uint_as_float(t[0][k] << 16);
            else if (o == 4) pf1 = __uint_as_float(t[0][k] & 0xffff0000u);
            else if (o == 5) pr0 = pv0 - pf0;
            else if (o == 6) pr1 = pv1 - pf1;
            else if (o == 7) t[1][k] = b3_cvt_pk(pr0, pr1);
            else if (o == 8) pf0 = __uint_as_float(t[1][k] << 16);
            else if (o == 9) pf1 = __uint_as_float(t[1][k] & 0xffff0000u);
            else if (o == 10) pv0 = pr0 - pf0;
            else if (o == 11) pv1 = pr1 - pf1;
            else t[2][k] = b3_cvt_pk(pv0, pv1);
        };
        // group g = (tile block g >> 1, channel half g & 1) of a row stage over patch columns (ca, cb): four reads ...
        auto group_reads = [&](const float4* buf, int g, int ca, int cb_) {
            const int tb = g >> 1, h = g & 1;
            const int oa = (h * 2 + (ca & 1)) * PH + (ca >> 1), ob = (h * 2 + (cb_ & 1)) * PH + (cb_ >> 1);
            const float4 r0 = buf[la[tb] + oa], r1 = buf[lb[tb] + oa], r2 = buf[la[tb] + ob], r3 = buf[lb[tb] + ob];
            raw[0][0] = r0.x, raw[0][1] = r0.y, raw[0][2] = r0.z, raw[0][3] = r0.w;
            raw[1][0] = r1.x, raw[1][1] = r1.y, raw[1][2] = r1.z, raw[1][3] = r1.w;
            raw[2][0] = r2.x, raw[2][1] = r2.y, raw[2][2] = r2.z, raw[2][3] = r2.w;
            raw[3][0] = r3.x, raw[3][1] = r3.y, raw[3][2] = r3.z, raw[3][3] = r3.w;
        };
        // ... and eight FMAs: w = d[ra] +/- d[rb]
        auto group_fma = [&](int g, int ca, int cb_, int i) {
            const int tb = g >> 1, h = g & 1, col = (i >> 2) ? cb_ : ca, e = i & 3;
            w[tb][col][4 * h + e] = fmaf(raw[2 * (i >> 2) + 1][e], row_sign, raw[2 * (i >> 2)][e]);
        };
        auto mfma_slot = [&](int cur, int nu, int j) {
            const int pr = j >> 2, tb = (j >> 1) & 1, nb = j & 1;
            const int ta = pr == 0 ? 2 : (pr == 1 || pr == 3) ? 1 : 0;               // (2,0) (1,1) (0,2) (1,0) (0,1) (0,0): small terms first
            const int tu = pr == 0 ? 0 : pr == 1 ? 1 : pr == 2 ? 2 : pr == 3 ? 0 : pr == 4 ? 1 : 0;
            if (!B3_ABL(4))
                acc[nu][tb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_bit_cast(b3_bf16x8, b3_u32x4{av[cur][tb][ta][0], av[cur][tb][ta][1], av[cur][tb][ta][2], av[cur][tb][ta][3]}),
                    __builtin_bit_cast(b3_bf16x8, ub[cur][nb][tu]), acc[nu][tb][nb], 0, 0, 0);
        };
        // a HEAVY block: 24 MFMAs of (cur, nu) beside a row stage over columns (ca, cb) of `buf` + the preparation of nu_next:
        // 4 groups x (8 FMAs + 2 units x 13) = 136 instructions over slots 1 .. 23, the reads of group g one group ahead
        auto heavy_block = [&](int cur, int nu, int nu_next, const float4* buf, int ca, int cb_, auto pre_slot1) {
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                mfma_slot(cur, nu, j);
                if (j == 0) pre_slot1();
                if (j == 0 && !B3_ABL(2)) group_reads(buf, 0, ca, cb_);
                if (j == 5 && !B3_ABL(2)) group_reads(buf, 1, ca, cb_);
                if (j == 11 && !B3_ABL(2)) group_reads(buf, 2, ca, cb_);
                if (j == 17 && !B3_ABL(2)) group_reads(buf, 3, ca, cb_);
                if (j >= 1 && !B3_ABL(2)) {
                    const int lo = (136 * (j - 1)) / 23, hi = (136 * j) / 23;
#pragma unroll
                    for (int o = lo; o < hi; ++o) {
                        const int g = o / 34, q = o % 34;
                        if (q < 8) group_fma(g, ca, cb_, q);
                        else unit_op(cur ^ 1, nu_next, g >> 1, 2 * (g & 1) + (q - 8) / 13, (q - 8) % 13);
                    }
                }
                B3_SLOT_FENCE();
            }
        };
        // a LIGHT block: 24 MFMAs beside the preparation of nu_next alone: 8 units x 13 = 104 instructions over 24 slots
        auto light_block = [&](int cur, int nu, int nu_next) {
#pragma unroll
            for (int j = 0; j < 24; ++j) {
                mfma_slot(cur, nu, j);
                const int lo = (104 * j) / 24, hi = (104 * (j + 1)) / 24;
#pragma unroll
                for (int o = lo; o < hi; ++o)
                    if (!B3_ABL(2)) unit_op(cur ^ 1, nu_next, (o / 13) >> 2, (o / 13) & 3, o % 13);
                B3_SLOT_FENCE();
            }
        };
        auto chunk_buf = [&](int chunk) { return lds4 + (chunk % kB3Bufs) * kB3BufSlots; };
        auto clamp_chunk = [&](int chunk) { return chunk < n_chunks ? chunk : n_chunks - 1; };

        // ---- prologue of the work item (not overlapped): two chunks requested, chunk 0 staged, its columns 0, 2 and nu 0 prepared
        // (a chunk beyond the last repeats the last one, into a buffer nobody reads; U(0, 0) sits between the two requests so that
        // block 0 of chunk 0 finds the order every other chunk has: U(chunk, 0), one DMA, U(chunk, 3).  Issuing a request blocks
        // while the memory pipeline is full - at a kernel's start all 256 workgroups ask for their first chunks at once, 2.7 us for
        // three chunks (profiles/r6/wino_b3_stamps.log) - so only what the first chunk needs is asked for here)
        dma_into(0, 0);
        load_u(ub[0], 0, 0);
        dma_into(1, clamp_chunk(1));
        B3_STAMP(item, 7);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kDma + 6) : "memory");
        __syncthreads();
        B3_STAMP(item, 8);
        {
            const float4* buf = chunk_buf(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                group_reads(buf, g, 0, 2);
#pragma unroll
                for (int i = 0; i < 8; ++i) group_fma(g, 0, 2, i);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int o = 0; o < 13; ++o) unit_op(0, 0, u >> 2, u & 3, o);
        }
        __builtin_amdgcn_sched_barrier(0);
        B3_STAMP(item, 1);

        for (int chunk = 0; chunk < (B3_ABL(16) ? 0 : n_chunks); ++chunk) {
            const float4* const buf = chunk_buf(chunk);
            // block 0: nu 0 (av[0], ub[0]); columns 1, 3 of this chunk; prepares nu 3 into av[1]
            if (!B3_ABL(1)) load_u(ub[1], chunk, 3);
            if (!B3_ABL(1)) B3_WAIT_U(kDma + 6, ub[0]);   // younger than U(chunk, 0): the DMA of block 3 of the previous chunk (or the prologue's) + U(chunk, 3)
            __builtin_amdgcn_sched_barrier(0);
            heavy_block(0, 0, 3, buf, 1, 3, [] {});
            // block 1: nu 3; prepares nu 1 into av[0]
            if (!B3_ABL(1)) load_u(ub[0], chunk, 1);
            if (!B3_ABL(1)) B3_WAIT_U(6, ub[1]);
            __builtin_amdgcn_sched_barrier(0);
            light_block(1, 3, 1);
            // block 2: nu 1; prepares nu 2 into av[1]
            if (!B3_ABL(1)) load_u(ub[1], chunk, 2);
            if (!B3_ABL(1)) B3_WAIT_U(6, ub[0]);
            __builtin_amdgcn_sched_barrier(0);
            light_block(0, 1, 2);
            // block 3: nu 2; chunk + 1: barrier, DMA of chunk + 2, columns 0, 2, nu 0 into av[0]
            if (!B3_ABL(1)) load_u(ub[0], clamp_chunk(chunk + 1), 0);
            if (!B3_ABL(1)) B3_WAIT_U(6, ub[1]);          // (everything older has landed too: this wave's pieces of chunk + 1)
            __builtin_amdgcn_sched_barrier(0);
            const int c2 = clamp_chunk(chunk + 2);
            heavy_block(1, 2, 0, chunk_buf(chunk + 1), 0, 2, [&] {
                if (B3_ABL(8)) return;
                __syncthreads();          // chunk + 1's staged input is complete; every wave has left the buffer chunk + 2 goes into
                dma_into((chunk + 2) % kB3Bufs, c2);   // (the buffer of chunk - 1: last read in block 0 of chunk - 1)
            });
        }
        B3_STAMP(item, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the repeated requests past the last chunk have landed: the exchange overlays them

        // ---- output transform.  Column half in registers: s_j = sum_nu M[xi][nu] A[nu][j]: j 0: m0 + m1 + m2, j 1: m1 - m2 - m3
        if (B3_ABL(32)) continue;
        __syncthreads();   // every wave has finished reading the chunk buffers: the exchange area overlays them
        float* const exch = b3_lds;   // [xi][j][block = 2 nb + tb][r / 4][lane][4]
        int lane_e = lane;             // opaque: the exchange's per-lane addresses are formed here, not above the K loop (and spilled)
        asm volatile("" : "+v"(lane_e));
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    float4 s0, s1;
                    float* p0 = reinterpret_cast<float*>(&s0);
                    float* p1 = reinterpret_cast<float*>(&s1);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * rq + e;
                        const float m0 = acc[0][tb][nb][r], m1 = acc[1][tb][nb][r], m2 = acc[2][tb][nb][r], m3 = acc[3][tb][nb][r];
                        p0[e] = m0 + m1 + m2;
                        p1[e] = m1 - m2 - m3;
                    }
                    const int blkid = 2 * nb + tb;
                    reinterpret_cast<float4*>(exch)[(((xi * 2 + 0) * 4 + blkid) * 4 + rq) * 64 + lane_e] = s0;
                    reinterpret_cast<float4*>(exch)[(((xi * 2 + 1) * 4 + blkid) * 4 + rq) * 64 + lane_e] = s1;
                }
        B3_STAMP(item, 3);
        __syncthreads();
        B3_STAMP(item, 4);
        // ---- wave w finishes block w: row half Y[0][j] = s0 + s1 + s2, Y[1][j] = s1 - s2 - s3; lane = output channel, r = tile
        {
            int li_o = li, hl_o = hl;   // opaque: keeps the epilogue's lane-dependent offsets from being hoisted above the K loop (and spilled)
            asm volatile("" : "+v"(li_o), "+v"(hl_o));
            const int tb = xi & 1, nb = xi >> 1, blkid = xi;
            const int co = cb * 64 + 32 * nb + li_o;
            const float bj = bias ? bias[co] : 0.f;
            const float floor_ = relu ? 0.f : -INFINITY;
            const int Ho = POOL ? TH : H, Wo = POOL ? TW : W;
            const int ps = out_nhwc ? Cout : 8;   // floats between pixels
            [[maybe_unused]] BnEpilogue bn = {0.f, 0.f, 0.f, 0.f};   // BN: the statistics of the BatchNorm behind this convolution (bn_epilogue.h)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                float4 sv[4][2];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int j = 0; j < 2; ++j) sv[a][j] = reinterpret_cast<const float4*>(exch)[(((a * 2 + j) * 4 + blkid) * 4 + rq) * 64 + lane_e];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * rq + e;
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * hl_o;       // tile li' of the block held by register r of this lane
                    float o[2][2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const float a0 = reinterpret_cast<const float*>(&sv[0][j])[e], a1 = reinterpret_cast<const float*>(&sv[1][j])[e];
                        const float a2 = reinterpret_cast<const float*>(&sv[2][j])[e], a3 = reinterpret_cast<const float*>(&sv[3][j])[e];
                        o[0][j] = a0 + a1 + a2;
                        o[1][j] = a1 - a2 - a3;
                    }
                    // (row >> 4 = r >> 3 for every lane: the strip of register r is uniform, only the tile column depends on hl)
                    const int strip = TC >= 64 ? 0 : (TC == 32 ? tb : 2 * tb + (r >> 3));
                    const int tcol = TC >= 64 ? 32 * tb + row : (TC == 32 ? row : (row & 15));
                    const int b_ = s_b[strip], th_ = s_th[strip];
                    const int tw_ = tc0 + tcol;
                    const int ow = 2 * tw_;
                    const bool col1 = ow + 1 < W, row1 = 2 * th_ + 1 < H;
                    if constexpr (BN && !POOL) {
                        {
                            if (r == 0) bn.k = o[0][0];
                            const float in_tile = (s_ok[strip] && tw_ < TW) ? 1.f : 0.f, c1 = col1 ? in_tile : 0.f, r1 = row1 ? 1.f : 0.f;
                            bn_epilogue_add(bn, o[0][0], in_tile);
                            bn_epilogue_add(bn, o[0][1], c1);
                            bn_epilogue_add(bn, o[1][0], in_tile * r1);
                            bn_epilogue_add(bn, o[1][1], c1 * r1);
                        }
                    }
                    if (!s_ok[strip] || tw_ >= TW || (B3_ABL(64) && B != -12345)) continue;
                    float* const yb = y + (size_t)b_ * Ho * Wo * Cout + (out_nhwc ? (size_t)co : ((size_t)(co >> 3) * Ho * Wo) * 8 + (co & 7));
                    if constexpr (POOL) {
                        float pooled = o[0][0];
                        if (col1) pooled = fmaxf(pooled, o[0][1]);
                        if (row1) {
                            pooled = fmaxf(pooled, o[1][0]);
                            if (col1) pooled = fmaxf(pooled, o[1][1]);
                        }
                        yb[((size_t)th_ * Wo + tw_) * ps] = fmaxf(pooled + bj, floor_);
                    } else {
                        float* const yp = yb + ((size_t)(2 * th_) * Wo + ow) * ps;
                        const size_t rowp = (size_t)Wo * ps;
                        yp[0] = fmaxf(o[0][0] + bj, floor_);
                        if (col1) yp[ps] = fmaxf(o[0][1] + bj, floor_);
                        if (row1) {
                            yp[rowp] = fmaxf(o[1][0] + bj, floor_);
                            if (col1) yp[rowp + ps] = fmaxf(o[1][1] + bj, floor_);
                        }
                    }
                }
            }
            if constexpr (BN && !POOL) bn_epilogue_flush(bn, bn_sums, Cout, co, (int)(blockIdx.x % (unsigned)bn_slots(Cout)));
        }
        B3_STAMP(item, 5);
    }
}

template <bool POOL, bool IN_NHWC, bool BN = false>
static hipError_t wino_b3_launch(int tc, unsigned grid, hipStream_t s, const float* x, const uint4* packed, const float* bias, float* y,
                                 int batch, int height, int width, int cin, int cout, int out_nhwc, int relu, double* bn_sums) {
    if (tc >= 64) k_conv3x3_wino_b3<POOL, 64, IN_NHWC, BN><<<grid, 256, kB3LdsBytes, s>>>(x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, bn_sums);
    else if (tc >= 32) k_conv3x3_wino_b3<POOL, 32, IN_NHWC, BN><<<grid, 256, kB3LdsBytes, s>>>(x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, bn_sums);
    else k_conv3x3_wino_b3<POOL, 16, IN_NHWC, BN><<<grid, 256, kB3LdsBytes, s>>>(x, packed, bias, y, batch, height, width, cin, cout, out_nhwc, relu, bn_sums);
    return hipGetLastError();
}

template <bool POOL, bool IN_NHWC, bool BN = false>
static hipError_t wino_b3_set_lds_limit() {
    const void* ks[3] = {(const void*)k_conv3x3_wino_b3<POOL, 64, IN_NHWC, BN>, (const void*)k_conv3x3_wino_b3<POOL, 32, IN_NHWC, BN>,
                         (const void*)k_conv3x3_wino_b3<POOL, 16, IN_NHWC, BN>};
    for (const void* k : ks) {
        hipError_t e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kB3LdsBytes);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Same contract as iris_conv3x3_wino (k_conv_wino.h) with `packed` from iris_wino_b3_pack_weights_device and cin % 16 == 0.
static int conv3x3_wino_b3_impl(const float* x, const float* packed, const float* bias, float* y, int batch, int height, int width,
                                int cin, int cout, int flags, double* bn_sums, void* stream) {
    if (!x || !packed || !y) return fail(IRIS_E_INVALID, "iris_conv3x3_wino_b3: NULL argument");
    if (bn_sums && (bias || (flags & (IRIS_WINO_POOL | IRIS_WINO_RELU)) || !(flags & IRIS_WINO_OUT_NHWC)))
        return fail(IRIS_E_INVALID, "iris_conv3x3_wino_b3_bn: the statistics are those of the bare convolution, channels-last out (no bias / ReLU / pooling)");
    if (batch <= 0 || height <= 0 || width <= 0) return fail(IRIS_E_INVALID, "iris_conv3x3_wino_b3: empty tensor");
    if (flags & ~(IRIS_WINO_POOL | IRIS_WINO_OUT_NHWC | IRIS_WINO_IN_NHWC | IRIS_WINO_RELU)) return fail(IRIS_E_INVALID, "iris_conv3x3_wino_b3: flags 0x%x", flags);
    if (cin <= 0 || cout <= 0 || (cin % kB3KC) || (cout % 64))
        return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_wino_b3: cin %d must be a multiple of %d, cout %d of 64", cin, kB3KC, cout);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(packed)) & 15)
        return fail(IRIS_E_INVALID, "iris_conv3x3_wino_b3: x and the packed weights must be 16-byte aligned");
    if ((long long)batch * height * width * cin >= 1073741824LL)
        return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_wino_b3: tensor too large for 32-bit byte offsets (>= 2^30 elements)");
    int dev = 0, n_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    static std::atomic<unsigned> attr_set[64];
    if (dev < 0 || dev >= 64 || !attr_set[dev].load(std::memory_order_acquire)) {
        HIP_TRY((wino_b3_set_lds_limit<false, false>()));
        HIP_TRY((wino_b3_set_lds_limit<false, true>()));
        HIP_TRY((wino_b3_set_lds_limit<true, false>()));
        HIP_TRY((wino_b3_set_lds_limit<true, true>()));
        HIP_TRY((wino_b3_set_lds_limit<false, false, true>()));
        HIP_TRY((wino_b3_set_lds_limit<false, true, true>()));
        if (dev >= 0 && dev < 64) attr_set[dev].store(1u, std::memory_order_release);
    }
    const int pool = (flags & IRIS_WINO_POOL) != 0, out_nhwc = (flags & IRIS_WINO_OUT_NHWC) != 0;
    const int in_nhwc = (flags & IRIS_WINO_IN_NHWC) != 0, relu = (flags & IRIS_WINO_RELU) != 0;
    const int th = (height + 1) / 2, tw = (width + 1) / 2;
    const int tc = tw > 32 ? 64 : (tw > 16 ? 32 : 16), tr = 64 / tc;
    const long long n_work = (((long long)batch * th + tr - 1) / tr) * ((tw + tc - 1) / tc) * (cout / 64);
    if (n_work >= 2147483647LL) return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_wino_b3: too many tiles");
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
    const unsigned grid = (unsigned)std::min<long long>(n_work, n_cu);
    const hipStream_t st = (hipStream_t)stream;
    const uint4* pk = reinterpret_cast<const uint4*>(packed);
    hipError_t e;
    if (pool) e = in_nhwc ? wino_b3_launch<true, true>(tc, grid, st, x, pk, bias, y, batch, height, width, cin, cout, out_nhwc, relu, nullptr)
                          : wino_b3_launch<true, false>(tc, grid, st, x, pk, bias, y, batch, height, width, cin, cout, out_nhwc, relu, nullptr);
    else if (bn_sums) e = in_nhwc ? wino_b3_launch<false, true, true>(tc, grid, st, x, pk, bias, y, batch, height, width, cin, cout, out_nhwc, relu, bn_sums)
                                  : wino_b3_launch<false, false, true>(tc, grid, st, x, pk, bias, y, batch, height, width, cin, cout, out_nhwc, relu, bn_sums);
    else e = in_nhwc ? wino_b3_launch<false, true>(tc, grid, st, x, pk, bias, y, batch, height, width, cin, cout, out_nhwc, relu, nullptr)
                     : wino_b3_launch<false, false>(tc, grid, st, x, pk, bias, y, batch, height, width, cin, cout, out_nhwc, relu, nullptr);
    HIP_TRY(e);
    return IRIS_OK;
}

extern "C" int iris_conv3x3_wino_b3(const float* x, const float* packed, const float* bias, float* y, int batch, int height, int width,
                                    int cin, int cout, int flags, void* stream) {
    return conv3x3_wino_b3_impl(x, packed, bias, y, batch, height, width, cin, cout, flags, nullptr, stream);
}
// as iris_conv3x3_wino_bn: the bare convolution + the statistics of the BatchNorm behind it
extern "C" int iris_conv3x3_wino_b3_bn(const float* x, const float* packed, float* y, int batch, int height, int width, int cin, int cout,
                                       int flags, double* bn_sums_zeroed, void* stream) {
    if (!bn_sums_zeroed) return fail(IRIS_E_INVALID, "iris_conv3x3_wino_b3_bn: NULL argument");
    return conv3x3_wino_b3_impl(x, packed, nullptr, y, batch, height, width, cin, cout, flags, bn_sums_zeroed, stream);
}
