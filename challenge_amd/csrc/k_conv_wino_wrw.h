// k_conv_wino_wrw.h -- the WEIGHT GRADIENT of Conv2D(3x3 'same', Cin -> Cout) as a Winograd F(2x2, 3x3) transform on the fp32
// matrix cores: the training step's blocks 2-5 (sj_train.py:191-201, 222-242, 408: model.fit's backward pass).
// Part of the single translation unit iris_frontend.hip (and of scripts/microbench/wino_conv.hip, which builds it alone).
#pragma once
// ---------------------------------------------------------------------------
// Forward:  Y = A^T [ sum_cin U (.) V ] A  with U = G g G^T, V = B^T d B (k_conv_wino.h).  Hence
//   dM[p][tile][cout] = A dY A^T                 2 x 2 output-gradient tile -> 16 positions p
//   dU[p][cout][cin]  = sum_tiles dM[p][tile][cout] V[p][tile][cin]      16 GEMMs with K = tiles of the whole batch
//   dg[cout][cin]     = G^T dU G                 4 x 4 -> 3 x 3, once per layer
// 16 instead of 36 multiplies per 2 x 2 tile and channel pair, as in the forward pass; MIOpen runs this layer's weight
// gradient as an implicit GEMM at 82-100 TFLOP/s (profiles/r5/c4_step_kernel_stats.csv).
// The GEMM's two free dimensions are BOTH channels and its reduction runs over tiles, so with channels-last activations
// (x [B][H][W][Cin], dy [B][H][W][Cout] - the layout the training step keeps them in) the operand layout of
// v_mfma_f32_32x32x2_f32 is the memory layout: lane l holds channel l & 31 of tile l >> 5 for A (dM: 32 cout x 2 tiles) and for
// B (V: 2 tiles x 32 cin).  A lane loads ITS channel's 4 x 4 input patch and 2 x 2 gradient tile straight from memory
// (32 consecutive channels = one 128-byte line per pixel and half-wave), transforms both in registers and feeds the matrix
// core: no LDS, no cross-lane traffic, no barrier anywhere in the loop.
// Decomposition: a wave owns 32 cout x 32 cin x 8 positions = 8 accumulators (128 AGPRs, two waves per SIMD); a workgroup
// of 8 waves a 64 x 64 block x 16 positions; the tile rows of the batch are split over `n_split` workgroups per block, each writing its partial
// dU to a workspace [split][16][Cout][Cin]; k_wino_wrw_reduce sums the splits in a fixed order (deterministic, unlike MIOpen's
// atomics) and applies G^T . G.  Out-of-image pixels: buffer loads with an out-of-range offset return 0 without a branch.
// Signs: A's last row is (0, -1); the kernel accumulates with +1 there (dM' = s_a s_b dM, s = (1, 1, 1, -1): adds only) and the
// reduce kernel folds s into G.
// ---------------------------------------------------------------------------
#ifndef IRIS_WINO_WRW_DEPTH
#define IRIS_WINO_WRW_DEPTH 4   // tiles in flight per lane: a tile's loads are issued DEPTH batches ahead of its transform
#endif
// timing experiments only (results wrong when non-zero): 1 no loads in the loop, 2 no transform, 4 every tile row reads image rows 0-3 of
// image 0 (the loads stay, their data comes from the L2), 8 no partial stores
#ifndef IRIS_WRW_ABLATE
#define IRIS_WRW_ABLATE 0
#endif
#define WRW_ABL(bit) ((IRIS_WRW_ABLATE & (bit)) != 0)
typedef float wrw_f2 __attribute__((ext_vector_type(2)));
// packed fp32 adds on register pairs, written out (the compiler scalarises vector fsub): plain, and with the halves of the
// sources selected per result half (VOP3P op_sel / op_sel_hi: 0 = low register of the pair, 1 = high) and negated (neg_lo / neg_hi)
#define WRW_PK(name, mods)                                                       \
    __device__ __forceinline__ wrw_f2 name(wrw_f2 a, wrw_f2 b) {                 \
        wrw_f2 r;                                                                \
        asm("v_pk_add_f32 %0, %1, %2 " mods : "=v"(r) : "v"(a), "v"(b));         \
        return r;                                                                \
    }
WRW_PK(wrw_add, "")                                                               // (a0 + b0, a1 + b1)
WRW_PK(wrw_sub, "neg_lo:[0,1] neg_hi:[0,1]")                                      // (a0 - b0, a1 - b1)
WRW_PK(wrw_row01, "op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1]")                    // (a0 - b0, a1 + b0)
WRW_PK(wrw_row23, "op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]")       // (a0 - b1, b1 - a1)
WRW_PK(wrw_sumdiff, "op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]")                  // (a0 + b1, a0 - b1)

// 1: a lane walks its tiles left to right and keeps the column transform of the two patch columns it shares with the next
// tile (6 + 4 loads per tile instead of 12 + 4; the first tile of a tile row loads all four columns); 0: every tile loads its
// whole patch
#ifndef IRIS_WRW_SLIDE
#define IRIS_WRW_SLIDE 1
#endif
#ifndef IRIS_WRW_INTERLEAVE
#define IRIS_WRW_INTERLEAVE 1
#endif
struct WrwRaw {
    wrw_f2 x[3][2];   // the three patch rows this wave's position half needs x (columns 0 1, columns 2 3)
    wrw_f2 d[2];      // the gradient tile's rows (columns 0 1)
    int first;        // uniform: the tile is the first of its tile row (x[.][0] was loaded)
};

__device__ __forceinline__ float wrw_ld(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0));
}
// descriptor of ONE image row (`bytes` = 0: a row outside the image, every load returns 0): a column left or right of the image
// is then an offset outside the descriptor - the hardware's range check is the zero padding
__device__ __forceinline__ __amdgpu_buffer_rsrc_t wrw_row_rsrc(const float* base, long long row_floats, int bytes) {
    const uint64_t a = reinterpret_cast<uint64_t>(base + row_floats);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((uint64_t)hi << 32) | lo), 0, bytes, 0x00020000);
}

// x: [B][H][W][Cin], dy: [B][H][W][Cout] (channels-last, fp32), part: [n_split][16][Cout][Cin]
// 8 waves = (cout half, cin half, position half) of a 64 x 64 block: TWO waves per SIMD (128 accumulators each), so that one
// wave's loads, address arithmetic and transform run under the other's MFMAs and a wave waiting for memory does not idle the
// matrix core (one wave per SIMD with all 16 positions: 62-68 TFLOP/s, memory latency exposed - EXPERIMENTS.md round 5).
// PH = the wave's position half: positions 8 PH .. 8 PH + 7 = rows 2 PH, 2 PH + 1 of the 4 x 4 transforms, which read the patch
// rows PH .. PH + 2 only.  Per tile pair a wave issues 8 MFMAs, 16 loads, 11 packed adds and 6 offset increments; the row
// bookkeeping (five row descriptors) runs once per tile row under a uniform branch.
template <int PH, int MW, int NW>
__device__ __forceinline__ void wino_wrw_wave(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                              int B, int H, int W, int Cin, int Cout, int n_split, int wm, int wn, int wk) {
    constexpr int D = IRIS_WINO_WRW_DEPTH;
    const int lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;  // channel of the wave's 32, tile parity
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1, hn = (TW + 1) >> 1;  // tiles per half-wave and tile row
    const int n_rows = B * TH;
    const int cin_blocks = Cin / (32 * NW), n_bp = cin_blocks * (Cout / (32 * MW));
    const int split = wk / n_bp, bp = wk - split * n_bp;
    const int cb = bp / cin_blocks, ib = bp - cb * cin_blocks;
    const int R_lo = (int)(((long long)n_rows * split) / n_split), R_hi = (int)(((long long)n_rows * (split + 1)) / n_split);
    const int n_it = (R_hi - R_lo) * hn;
    const unsigned xpix = (unsigned)Cin * 4u, dpix = (unsigned)Cout * 4u;
    const int xrow_bytes = W * (int)xpix, drow_bytes = W * (int)dpix;

    // the load stream: tile row ld_R = (ld_b, ld_th) and column pair ld_j are uniform; a lane's tile column is kh hn + ld_j.
    // Byte offsets inside the row of the lane's channel at the patch's four columns / the tile's two columns: the first is
    // "negative" (wraps) for tile column 0, the last ones pass the row's end for the last tile - both outside the row descriptor.
    unsigned col[4], dcol[2];
    auto set_cols = [&]() {
        const int tw0 = kh * hn;
#pragma unroll
        for (int c = 0; c < 4; ++c) col[c] = (unsigned)(2 * tw0 - 1 + c) * xpix + (unsigned)(ib * 32 * NW + 32 * wn + li) * 4u;
#pragma unroll
        for (int j = 0; j < 2; ++j) dcol[j] = (unsigned)(2 * tw0 + j) * dpix + (unsigned)(cb * 32 * MW + 32 * wm + li) * 4u;
    };
    __amdgpu_buffer_rsrc_t rxr[3], rdr[2];
    int ld_j = 0, ld_R = R_lo, ld_b = R_lo / TH, ld_th = R_lo - (R_lo / TH) * TH;
    auto set_rows = [&]() {
        const int live = ld_R < R_hi;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int hh = 2 * ld_th - 1 + PH + r;
            const int ok = live & ((unsigned)hh < (unsigned)H);
            rxr[r] = wrw_row_rsrc(x, ok ? (long long)(WRW_ABL(4) ? r : ld_b * H + hh) * W * Cin : 0, ok ? xrow_bytes : 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hh = 2 * ld_th + i;
            const int ok = live & (hh < H);
            rdr[i] = wrw_row_rsrc(dy, ok ? (long long)(WRW_ABL(4) ? i : ld_b * H + hh) * W * Cout : 0, ok ? drow_bytes : 0);
        }
    };
    set_cols();
    set_rows();
    auto issue_first = [&](WrwRaw& raw) {   // the two left patch columns: only for the first tile of a tile row
        raw.first = !IRIS_WRW_SLIDE || ld_j == 0;
        if (raw.first) {   // uniform
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 2; ++c) raw.x[r][0][c] = wrw_ld(rxr[r], col[c]);
        }
    };
    auto issue_loads = [&](WrwRaw& raw) {   // 6 + 4 loads, no control flow: they share a basic block with the tile's MFMAs
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 2; c < 4; ++c) raw.x[r][1][c & 1] = wrw_ld(rxr[r], col[c]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) raw.d[i][j] = wrw_ld(rdr[i], dcol[j]);
    };
    auto advance = [&]() {
        if (++ld_j == hn) {   // uniform: next tile row
            ld_j = 0;
            ++ld_R;
            if (++ld_th == TH) {
                ld_th = 0;
                ++ld_b;
            }
            set_cols();
            set_rows();
        } else {
#pragma unroll
            for (int c = 0; c < 4; ++c) col[c] += 2u * xpix;
#pragma unroll
            for (int j = 0; j < 2; ++j) dcol[j] += 2u * dpix;
        }
    };
    auto issue = [&](WrwRaw& raw) {
        issue_first(raw);
        issue_loads(raw);
        advance();
    };
    // rows 2 PH, 2 PH + 1 of V = B^T d B and of dM' = |A| dY |A|^T (see the header for the signs): 8 + 8 values, as pairs
    // V[i][0] = (V_i0, V_i1), V[i][1] = (V_i2, V_i3); M likewise
    wrw_f2 carry[2] = {wrw_f2{0.f, 0.f}, wrw_f2{0.f, 0.f}};   // B^T d of the previous tile's columns 2 3 = this tile's columns 0 1
    auto xform = [&](const WrwRaw& raw, wrw_f2 (&V)[2][2], wrw_f2 (&M)[2][2]) {
        wrw_f2 t[2][2], s[2];
        auto colop = [&](int h, wrw_f2 (&out)[2]) {
            if (PH == 0) {   // patch rows 0 1 2: t0 = d0 - d2, t1 = d1 + d2
                out[0] = wrw_sub(raw.x[0][h], raw.x[2][h]);
                out[1] = wrw_add(raw.x[1][h], raw.x[2][h]);
            } else {         // patch rows 1 2 3: t2 = d2 - d1, t3 = d1 - d3
                out[0] = wrw_sub(raw.x[1][h], raw.x[0][h]);
                out[1] = wrw_sub(raw.x[0][h], raw.x[2][h]);
            }
        };
        if (raw.first) colop(0, carry);   // uniform
        wrw_f2 right[2];
        colop(1, right);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            t[i][0] = carry[i];
            t[i][1] = right[i];
            carry[i] = right[i];
        }
        if (PH == 0) {       // s0 = y0, s1 = y0 + y1
            s[0] = raw.d[0];
            s[1] = wrw_add(raw.d[0], raw.d[1]);
        } else {             // s2 = y0 - y1, s3 = y1
            s[0] = wrw_sub(raw.d[0], raw.d[1]);
            s[1] = raw.d[1];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            V[i][0] = wrw_row01(t[i][0], t[i][1]);   // (t0 - t2, t1 + t2)
            V[i][1] = wrw_row23(t[i][1], t[i][0]);   // (t2 - t1, t1 - t3)
            const wrw_f2 sd = wrw_sumdiff(s[i], s[i]);   // (s0 + s1, s0 - s1)
            M[i][0] = wrw_f2{s[i][0], sd[0]};
            M[i][1] = wrw_f2{sd[1], s[i][1]};
        }
    };

    f32x16 acc[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    WrwRaw raw[D];
    wrw_f2 V[2][2], M[2][2];
#pragma unroll
    for (int q = 0; q < D; ++q) issue(raw[q]);
    for (int it = 0; it < n_it; it += D) {
#pragma unroll
        for (int q = 0; q < D; ++q) {   // tile it + q: transform, reload its registers with tile it + q + D, 8 MFMAs
            if (!WRW_ABL(2) || it == 0) xform(raw[q], V, M);
            if (!WRW_ABL(1)) issue_first(raw[q]);
            if (!WRW_ABL(1)) issue_loads(raw[q]);
#pragma unroll
            for (int p = 0; p < 8; ++p)
                acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(M[p >> 2][(p >> 1) & 1][p & 1], V[p >> 2][(p >> 1) & 1][p & 1], acc[p], 0, 0, 0);
#if IRIS_WRW_INTERLEAVE
            // the tile's loads spread between its MFMAs instead of in one burst in front of them
#pragma unroll
            for (int p = 0; p < 5; ++p) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            }
#endif
            if (!WRW_ABL(1)) advance();
        }
    }
    // partial dU' of this split: [split][p][cout][cin]; D register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
    float* const out = part + ((size_t)split * 16 + 8 * PH) * Cout * Cin + (size_t)(cb * 32 * MW + 32 * wm + 4 * kh) * Cin + (ib * 32 * NW + 32 * wn + li);
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (!WRW_ABL(8) || acc[p][r] == 12345.f) out[(size_t)p * Cout * Cin + (size_t)((r & 3) + 8 * (r >> 2)) * Cin] = acc[p][r];
}

// MW x NW = halves of 32 output / input channels of a workgroup's block: 2 x 2 = 8 waves, 64 x 64 (cout % 64 == 0, cin % 64 == 0);
// 2 x 1 (1 x 2) = 4 waves, 64 cout x 32 cin (the 32 -> 64 layer) or 32 x 64; 1 x 1 = 2 waves, 32 x 32 (block 1's 32 -> 32 layer).  Several of
// the smaller workgroups share a CU, so every SIMD still holds two waves.
template <int MW, int NW>
__global__ __launch_bounds__(512, 2) void k_wino_wrw(const float* __restrict__ x, const float* __restrict__ dy,
                                                     float* __restrict__ part, int B, int H, int W, int Cin, int Cout, int n_split) {
    // Always 8 waves = two per SIMD: 4 / (MW NW) blocks of the decomposition share a workgroup (2-wave workgroups of their own
    // were placed unevenly over the SIMDs: a bare MFMA loop ran at 3/4 of its rate).  Waves w and w + 4 share a SIMD
    // (round-robin placement): the same block and channels, the two position halves - their loads of the shared patch rows
    // hit the same lines.
    constexpr int kGroups = 4 / (MW * NW);
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lg = (wv & 3) / (MW * NW), q = (wv & 3) % (MW * NW);
    const int total = (Cin / (32 * NW)) * (Cout / (32 * MW)) * n_split;
    // workgroups of one XCD (blockIdx mod 8) take neighbouring work: the same tile rows for different channel blocks
    int wg = blockIdx.x;
    if ((gridDim.x & 7) == 0) wg = (wg & 7) * (gridDim.x >> 3) + (wg >> 3);
    const int wk = wg * kGroups + lg;
    if (wk >= total) return;   // (no barrier anywhere in the kernel)
    if ((wv >> 2) == 0) wino_wrw_wave<0, MW, NW>(x, dy, part, B, H, W, Cin, Cout, n_split, q % MW, q / MW, wk);
    else wino_wrw_wave<1, MW, NW>(x, dy, part, B, H, W, Cin, Cout, n_split, q % MW, q / MW, wk);
}

// First stage of the sum over many splits (layers with few channel blocks: 256 splits of a 64 x 64 layer), in place: split g < G
// becomes the sum of the splits g, g + G, g + 2 G ... (ascending: a fixed order) - enough threads to read at memory speed, which
// the (cout, 64 cin) workgroups of k_wino_wrw_reduce alone are not when there are only 64 of them.
__global__ __launch_bounds__(256) void k_wino_wrw_fold(float4* __restrict__ part, int n_split, int groups, size_t n4 /* float4 per split */) {
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int g = blockIdx.y;
    if (e >= n4) return;
    float4 a = part[(size_t)g * n4 + e];
    for (int s = g + groups; s < n_split; s += groups) {
        const float4 b = part[(size_t)s * n4 + e];
        a.x += b.x;
        a.y += b.y;
        a.z += b.z;
        a.w += b.w;
    }
    part[(size_t)g * n4 + e] = a;
}

// dW[cout][cin][a][b] = sum_{i, j} G[i][a] G[j][b] s_i s_j sum_split part[split][4 i + j][cout][cin], written with the weight
// tensor's own element strides (a channels_last parameter's gradient as it is).  One workgroup per (cout, CB = 64 or 32 cin): thread
// (i = t / CB, cin = t % CB) sums row i of the 4 x 4 over the splits in ascending order and applies G along j; the column
// pass over i goes through LDS.
template <int CB>
__global__ __launch_bounds__(4 * CB) void k_wino_wrw_reduce(const float* __restrict__ part, int n_split, int Cin, int Cout,
                                                            float* __restrict__ dw, long so, long si, long sh, long sw, int accumulate) {
    __shared__ float rows[4][3][CB];
    const int t = threadIdx.x, i = t / CB, cl = t % CB;
    const int cin_blocks = Cin / CB;
    const int co = blockIdx.x / cin_blocks, ci = (blockIdx.x - co * cin_blocks) * CB + cl;
    const size_t plane = (size_t)Cout * Cin;
    const float* p = part + ((size_t)4 * i) * plane + (size_t)co * Cin + ci;
    float u[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < n_split; ++s, p += 16 * plane)
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] += p[(size_t)j * plane];
    const float sg = i == 3 ? -1.f : 1.f;   // s_i; s_j = -1 for j == 3 folded below
    // (u G')[b] with G' = diag(s) G: columns (1, .5, .5, 0), (0, .5, -.5, 0), (0, .5, .5, -1)
    rows[i][0][cl] = sg * (u[0] + 0.5f * (u[1] + u[2]));
    rows[i][1][cl] = sg * (0.5f * (u[1] - u[2]));
    rows[i][2][cl] = sg * (0.5f * (u[1] + u[2]) - u[3]);
    __syncthreads();
    if (i < 3) {
        float* const dst = dw + (long)co * so + (long)ci * si + (long)i * sh;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const float r0 = rows[0][b][cl], r1 = rows[1][b][cl], r2 = rows[2][b][cl], r3 = rows[3][b][cl];
            // G^T rows: a = 0: (1, .5, .5, 0); a = 1: (0, .5, -.5, 0); a = 2: (0, .5, .5, 1)   (s_i already applied to r3)
            const float v = i == 0 ? r0 + 0.5f * (r1 + r2) : (i == 1 ? 0.5f * (r1 - r2) : 0.5f * (r1 + r2) + r3);
            if (accumulate) dst[(long)b * sw] += v; else dst[(long)b * sw] = v;
        }
    }
}

// workgroups of 64 cout x (64 or 32) cin; as many splits of the tile rows as fill every SIMD with two waves
static int wino_wrw_cin_block(int cin) { return (cin % 64) ? 32 : 64; }
static int wino_wrw_cout_block(int cout) { return (cout % 64) ? 32 : 64; }
static size_t wino_wrw_splits(int batch, int height, int cin, int cout, int n_cu) {
    const int cbk = wino_wrw_cin_block(cin), obk = wino_wrw_cout_block(cout), want = n_cu * (64 / cbk) * (64 / obk);
    const int n_bp = (cin / cbk) * (cout / obk), n_rows = batch * ((height + 1) / 2);
    return (size_t)std::max(1, std::min(n_rows, want / std::max(1, std::min(n_bp, want))));
}

// floats of workspace iris_conv3x3_wino_wrw needs for this geometry (partial sums of the tile-row splits)
extern "C" size_t iris_wino_wrw_workspace_len(int batch, int height, int width, int cin, int cout) {
    (void)width;
    if (batch <= 0 || height <= 0 || cin <= 0 || cout <= 0 || (cin % 32) || (cout % 32)) return 0;
    int dev = 0, n_cu = 256;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
    return wino_wrw_splits(batch, height, cin, cout, n_cu) * 16 * (size_t)cin * cout;
}

// dw (element strides so / si / sh / sw over [cout][cin][3][3]) = (accumulate ? dw : 0) + the weight gradient of
// y = conv3x3_same(x, w) given dy; x: [B][H][W][cin], dy: [B][H][W][cout], both channels-last fp32; workspace: >=
// iris_wino_wrw_workspace_len floats
extern "C" int iris_conv3x3_wino_wrw(const float* x, const float* dy, float* dw, long stride_o, long stride_i, long stride_h,
                                     long stride_w, int batch, int height, int width, int cin, int cout, int accumulate,
                                     float* workspace, size_t workspace_len, void* stream) {
    if (!x || !dy || !dw || !workspace) return fail(IRIS_E_INVALID, "iris_conv3x3_wino_wrw: NULL argument");
    if (batch <= 0 || height <= 0 || width <= 0) return fail(IRIS_E_INVALID, "iris_conv3x3_wino_wrw: empty tensor");
    if (cin <= 0 || cout <= 0 || (cin % 32) || (cout % 32))
        return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_wino_wrw: cin %d and cout %d must be multiples of 32", cin, cout);
    if ((long long)batch * height * width * std::max(cin, cout) * 4 >= 2147483648LL)
        return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_wino_wrw: tensor of 2^31 bytes or more");
    int dev = 0, n_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
    const size_t n_split = wino_wrw_splits(batch, height, cin, cout, n_cu);
    const int cbk = wino_wrw_cin_block(cin), obk = wino_wrw_cout_block(cout), n_bp = (cin / cbk) * (cout / obk);
    if (workspace_len < n_split * 16 * (size_t)cin * cout)
        return fail(IRIS_E_INVALID, "iris_conv3x3_wino_wrw: workspace of %zu floats, %zu needed", workspace_len, n_split * 16 * (size_t)cin * cout);
    const hipStream_t st = (hipStream_t)stream;
    const unsigned groups = 4u / ((unsigned)(obk / 32) * (unsigned)(cbk / 32));   // blocks of the decomposition per 8-wave workgroup
    const unsigned grid = ((unsigned)(n_bp * n_split) + groups - 1) / groups;
    if (obk == 64 && cbk == 64) k_wino_wrw<2, 2><<<grid, 512, 0, st>>>(x, dy, workspace, batch, height, width, cin, cout, (int)n_split);
    else if (obk == 64) k_wino_wrw<2, 1><<<grid, 512, 0, st>>>(x, dy, workspace, batch, height, width, cin, cout, (int)n_split);
    else if (cbk == 64) k_wino_wrw<1, 2><<<grid, 512, 0, st>>>(x, dy, workspace, batch, height, width, cin, cout, (int)n_split);
    else k_wino_wrw<1, 1><<<grid, 512, 0, st>>>(x, dy, workspace, batch, height, width, cin, cout, (int)n_split);
    HIP_TRY(hipGetLastError());
    int n_left = (int)n_split;
    if (n_left > 16) {
        const int groups = 8;
        const size_t n4 = (size_t)4 * cin * cout;
        k_wino_wrw_fold<<<dim3((unsigned)((n4 + 255) / 256), groups), 256, 0, st>>>(reinterpret_cast<float4*>(workspace), n_left, groups, n4);
        HIP_TRY(hipGetLastError());
        n_left = groups;
    }
    if (cbk == 64)
        k_wino_wrw_reduce<64><<<(unsigned)(cout * (cin / 64)), 256, 0, st>>>(workspace, n_left, cin, cout, dw, stride_o, stride_i,
                                                                             stride_h, stride_w, accumulate);
    else
        k_wino_wrw_reduce<32><<<(unsigned)(cout * (cin / 32)), 128, 0, st>>>(workspace, n_left, cin, cout, dw, stride_o, stride_i,
                                                                             stride_h, stride_w, accumulate);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
