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
// Decomposition: a wave owns 32 cout x 32 cin x 16 positions = 16 accumulators (256 AGPRs, one wave per SIMD); a workgroup
// of 4 waves a 64 x 64 block; the tile rows of the batch are split over `n_split` workgroups per block, each writing its partial
// dU to a workspace [split][16][Cout][Cin]; k_wino_wrw_reduce sums the splits in a fixed order (deterministic, unlike MIOpen's
// atomics) and applies G^T . G.  Out-of-image pixels: buffer loads with an out-of-range offset return 0 without a branch.
// Signs: A's last row is (0, -1); the kernel accumulates with +1 there (dM' = s_a s_b dM, s = (1, 1, 1, -1): adds only) and the
// reduce kernel folds s into G.
// ---------------------------------------------------------------------------
#ifndef IRIS_WINO_WRW_DEPTH
#define IRIS_WINO_WRW_DEPTH 4   // tile sets in flight per lane (loads issued DEPTH - 1 MFMA batches ahead of their transform)
#endif
constexpr unsigned kWrwBadCol = 0x40000000u, kWrwBadRow = 0x80000000u;   // byte offsets no tensor reaches (< 2^30 bytes checked)

struct WrwRaw {
    float x[4][4];
    float d[2][2];
};

__device__ __forceinline__ float wrw_ld(__amdgpu_buffer_rsrc_t rs, unsigned voff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, 0, 0));
}

// x: [B][H][W][Cin], dy: [B][H][W][Cout] (channels-last, fp32), part: [n_split][16][Cout][Cin]
__global__ __launch_bounds__(256, 1) void k_wino_wrw(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                     int B, int H, int W, int Cin, int Cout, int n_split) {
    constexpr int D = IRIS_WINO_WRW_DEPTH;
    const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, kh = lane >> 5;  // channel of the wave's 32, tile parity
    const int wm = wv & 1, wn = wv >> 1;       // cout half / cin half of the workgroup's 64 x 64 block
    const int TH = (H + 1) >> 1, TW = (W + 1) >> 1, hn = (TW + 1) >> 1;  // tiles per half-wave and tile row
    const int n_rows = B * TH;
    const int cin_blocks = Cin >> 6, n_bp = cin_blocks * (Cout >> 6), total = n_bp * n_split;
    // workgroups of one XCD (blockIdx mod 8) take neighbouring work: the same tile rows for different channel blocks
    int wk = blockIdx.x;
    if ((total & 7) == 0) wk = (wk & 7) * (total >> 3) + (wk >> 3);
    const int split = wk / n_bp, bp = wk - split * n_bp;
    const int cb = bp / cin_blocks, ib = bp - cb * cin_blocks;
    const int R_lo = (int)(((long long)n_rows * split) / n_split), R_hi = (int)(((long long)n_rows * (split + 1)) / n_split);
    const int n_it = (R_hi - R_lo) * hn;

    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)((size_t)B * H * W * Cin * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(dy), 0, (int)((size_t)B * H * W * Cout * 4), 0x00020000);
    const unsigned xch = (unsigned)(ib * 64 + 32 * wn + li) * 4u, dch = (unsigned)(cb * 64 + 32 * wm + li) * 4u;
    const unsigned xpix = (unsigned)Cin * 4u, dpix = (unsigned)Cout * 4u;

    // running position of the load stream (uniform): tile row R_lo + it / hn, column pair it % hn
    int ld_it = 0, ld_j = 0, ld_R = R_lo;
    unsigned xrow[4], drow[2];  // byte offsets of the patch / gradient rows of the current tile row (uniform), or kWrwBadRow
    auto set_rows = [&](int R) {
        const int b_ = R / TH, th = R - b_ * TH;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int hh = 2 * th - 1 + r;
            xrow[r] = (hh >= 0 && hh < H && R < R_hi) ? (unsigned)((b_ * H + hh) * W) * xpix : kWrwBadRow;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int hh = 2 * th + i;
            drow[i] = (hh < H && R < R_hi) ? (unsigned)((b_ * H + hh) * W) * dpix : kWrwBadRow;
        }
    };
    set_rows(ld_R);
    auto issue = [&](WrwRaw& raw) {
        const int tw = kh * hn + ld_j;
        const bool t_ok = tw < TW;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int ww = 2 * tw - 1 + c;
            const unsigned col = (t_ok && ww >= 0 && ww < W) ? (unsigned)ww * xpix + xch : kWrwBadCol;
#pragma unroll
            for (int r = 0; r < 4; ++r) raw.x[r][c] = wrw_ld(rx, col + xrow[r]);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ww = 2 * tw + j;
            const unsigned col = (t_ok && ww < W) ? (unsigned)ww * dpix + dch : kWrwBadCol;
#pragma unroll
            for (int i = 0; i < 2; ++i) raw.d[i][j] = wrw_ld(rd, col + drow[i]);
        }
        ++ld_it;
        if (++ld_j == hn) {
            ld_j = 0;
            set_rows(++ld_R);
        }
    };
    // V = B^T d B (16 values) and dM' = |A| dY |A|^T (16 values, see the header for the signs)
    auto xform = [&](const WrwRaw& raw, float (&V)[16], float (&M)[16]) {
        float t[4][4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            t[0][c] = raw.x[0][c] - raw.x[2][c];
            t[1][c] = raw.x[1][c] + raw.x[2][c];
            t[2][c] = raw.x[2][c] - raw.x[1][c];
            t[3][c] = raw.x[1][c] - raw.x[3][c];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            V[4 * i + 0] = t[i][0] - t[i][2];
            V[4 * i + 1] = t[i][1] + t[i][2];
            V[4 * i + 2] = t[i][2] - t[i][1];
            V[4 * i + 3] = t[i][1] - t[i][3];
        }
        float s[4][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            s[0][j] = raw.d[0][j];
            s[1][j] = raw.d[0][j] + raw.d[1][j];
            s[2][j] = raw.d[0][j] - raw.d[1][j];
            s[3][j] = raw.d[1][j];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            M[4 * a + 0] = s[a][0];
            M[4 * a + 1] = s[a][0] + s[a][1];
            M[4 * a + 2] = s[a][0] - s[a][1];
            M[4 * a + 3] = s[a][1];
        }
    };

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;

    WrwRaw raw[D];
    float V[2][16], M[2][16];
#pragma unroll
    for (int q = 0; q < D; ++q) issue(raw[q]);
    xform(raw[0], V[0], M[0]);
    issue(raw[0]);
    // batch `it`: MFMAs on tile it (operands V / M [it & 1]), transform of tile it + 1, loads of tile it + D + 1
    for (int it = 0; it < n_it; it += D) {
#pragma unroll
        for (int q = 0; q < D; ++q) {
            const int cur = q & 1, nxt = cur ^ 1;       // D is even: (it + q) & 1 == q & 1
            xform(raw[(q + 1) % D], V[nxt], M[nxt]);
            issue(raw[(q + 1) % D]);
#pragma unroll
            for (int p = 0; p < 16; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(M[cur][p], V[cur][p], acc[p], 0, 0, 0);
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);  // the next tile's transform / addresses in its shadow
                __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);  // and the loads of the tile D batches ahead
            }
        }
    }
    // partial dU' of this split: [split][p][cout][cin]; D register r of lane l = row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
    float* const out = part + ((size_t)split * 16) * Cout * Cin + (size_t)(cb * 64 + 32 * wm + 4 * kh) * Cin + (ib * 64 + 32 * wn + li);
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            out[(size_t)p * Cout * Cin + (size_t)((r & 3) + 8 * (r >> 2)) * Cin] = acc[p][r];
}

// dW[cout][cin][a][b] = sum_{i, j} G[i][a] G[j][b] s_i s_j sum_split part[split][4 i + j][cout][cin], written with the weight
// tensor's own element strides (a channels_last parameter's gradient as it is).  One workgroup per (cout, 64 cin): thread
// (i = t >> 6, cin = t & 63) sums row i of the 4 x 4 over the splits in ascending order and applies G along j; the column
// pass over i goes through LDS.
__global__ __launch_bounds__(256) void k_wino_wrw_reduce(const float* __restrict__ part, int n_split, int Cin, int Cout,
                                                         float* __restrict__ dw, long so, long si, long sh, long sw, int accumulate) {
    __shared__ float rows[4][3][64];
    const int t = threadIdx.x, i = t >> 6, cl = t & 63;
    const int cin_blocks = Cin >> 6;
    const int co = blockIdx.x / cin_blocks, ci = (blockIdx.x - co * cin_blocks) * 64 + cl;
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

static size_t wino_wrw_splits(int batch, int height, int cin, int cout, int n_cu) {
    const int n_bp = (cin / 64) * (cout / 64), n_rows = batch * ((height + 1) / 2);
    return (size_t)std::max(1, std::min(n_rows, n_cu / std::max(1, std::min(n_bp, n_cu))));
}

// floats of workspace iris_conv3x3_wino_wrw needs for this geometry (partial sums of the tile-row splits)
extern "C" size_t iris_wino_wrw_workspace_len(int batch, int height, int width, int cin, int cout) {
    (void)width;
    if (batch <= 0 || height <= 0 || cin <= 0 || cout <= 0 || (cin % 64) || (cout % 64)) return 0;
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
    if (cin <= 0 || cout <= 0 || (cin % 64) || (cout % 64))
        return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_wino_wrw: cin %d and cout %d must be multiples of 64", cin, cout);
    if ((long long)batch * height * width * std::max(cin, cout) * 4 >= 1073741824LL)
        return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_wino_wrw: tensor too large for the out-of-range offsets (>= 2^30 bytes)");
    int dev = 0, n_cu = 256;
    HIP_TRY(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n_cu = 256;
    const size_t n_split = wino_wrw_splits(batch, height, cin, cout, n_cu);
    const int n_bp = (cin / 64) * (cout / 64);
    if (workspace_len < n_split * 16 * (size_t)cin * cout)
        return fail(IRIS_E_INVALID, "iris_conv3x3_wino_wrw: workspace of %zu floats, %zu needed", workspace_len, n_split * 16 * (size_t)cin * cout);
    const hipStream_t st = (hipStream_t)stream;
    k_wino_wrw<<<(unsigned)(n_bp * n_split), 256, 0, st>>>(x, dy, workspace, batch, height, width, cin, cout, (int)n_split);
    HIP_TRY(hipGetLastError());
    k_wino_wrw_reduce<<<(unsigned)(cout * (cin / 64)), 256, 0, st>>>(workspace, (int)n_split, cin, cout, dw, stride_o, stride_i, stride_h,
                                                                     stride_w, accumulate);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
