// k_conv0_bn.h -- the CRNN's FIRST layer in training mode: Conv2D(3x3 'same', 1-2 input channels) + BatchNorm + ReLU
// (sj_train.py:191-201, 244) with the convolution RECOMPUTED wherever its output is needed.
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// With one input channel the convolution output z is 32x its input (c4: 8 MB in, 268 MB out) and costs 9 FMAs per value:
// cheaper to recompute from x than to read back.  z is therefore never stored:
//   forward   k_conv0_stats      per-channel sum / sum of squares of z, from x alone                       (reads 8 MB)
//             k_conv0_bn_relu    y = max(z scale + shift, 0), channels-last                       (reads 8 MB, writes y)
//   backward  k_conv0_bwd<false> g = dy [y > 0]; sum g, sum g xhat                                     (reads dy + x)
//             k_conv0_bwd<true>  dz = a g + b z + d;  dW[co][ci][ky][kx] += dz x[.., h + ky - 1, w + kx - 1]   (reads dy + x)
// instead of MIOpen's convolution (writes z), the statistics pass (reads z), the apply pass (reads z, writes y), the
// backward reduction and dx passes (read z and dy twice, write dz) and the weight-gradient convolution (reads dz): 2.4 GB of
// traffic become 0.8 GB.  The input needs no gradient (it is the log-mel feature), the convolution's bias none either
// (BatchNorm removes the batch mean; it only shifts the running mean).
// Thread layout: 256 threads = (256 / C4) pixels x C4 float4 channel groups; a thread keeps the 4 x CIN x 9 weights of its
// channel group in registers; a block owns an image row; x [B, CIN, H, W] contiguous, y / dy [B, H, W, C] channels-last.
// ---------------------------------------------------------------------------
constexpr int kConv0DwSlots = 8;  // copies of the weight-gradient staging buffer the blocks' atomics are spread over
constexpr int kConv0MaxW = 2048;  // the staged rows must fit the LDS next to nothing else: 3 x (W + 2) x CIN floats

// A block owns ONE image row (b, h): it stages rows h-1, h, h+1 of x (zero outside the image, one zero column either side)
// in LDS once - the 256 / C4 pixel threads of an iteration and their C4 channel-group threads all read their taps from
// there (the taps of a pixel are the same for its C4 threads: fetching them from global memory made the texture-address
// unit the limit, 9 x 64-lane loads for 8 distinct pixels per wave) - and walks the row 256 / C4 pixels at a time.
template <int CIN>
__device__ __forceinline__ void conv0_stage_rows(const float* __restrict__ x, float* tile, unsigned row, int H, int W) {
    const unsigned b = row / (unsigned)H, h = row - b * (unsigned)H;
    const int pitch = W + 2;
    for (int i = threadIdx.x; i < CIN * 3 * pitch; i += blockDim.x) {
        const int c = i / (3 * pitch), r = (i - c * 3 * pitch) / pitch, col = i - (c * 3 + r) * pitch;  // col 0 = w -1
        const int hh = (int)h + r - 1, ww = col - 1;
        const bool ok = hh >= 0 && hh < H && ww >= 0 && ww < W;
        tile[i] = ok ? x[(((size_t)b * CIN + c) * H + hh) * W + ww] : 0.f;
    }
}
template <int CIN>
struct Conv0Taps {
    float v[CIN][9];
};
template <int CIN>
__device__ __forceinline__ Conv0Taps<CIN> conv0_taps(const float* tile, int w, int W) {
    const int pitch = W + 2;
    Conv0Taps<CIN> t;
#pragma unroll
    for (int c = 0; c < CIN; ++c)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) t.v[c][ky * 3 + kx] = tile[(c * 3 + ky) * pitch + w + kx];
    return t;
}
// the thread's four output channels of one pixel: z[k] = sum over (ci, tap) of wreg[k][ci][tap] * tap value
template <int CIN>
__device__ __forceinline__ void conv0_z(const Conv0Taps<CIN>& t, const float (&wreg)[4][CIN][9], float (&z)[4]) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int q = 0; q < 9; ++q) a = fmaf(wreg[k][c][q], t.v[c][q], a);
        z[k] = a;
    }
}
template <int CIN>
__device__ __forceinline__ void conv0_load_weights(const float* __restrict__ w, int c4, float (&wreg)[4][CIN][9]) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int c = 0; c < CIN; ++c)
#pragma unroll
            for (int q = 0; q < 9; ++q) wreg[k][c][q] = w[((size_t)(4 * c4 + k) * CIN + c) * 9 + q];
}

// block-level fold of the threads' partials part[NVAL] over the pixel threads, then fp64 atomics: red[ty][C4][<= 36] in LDS
// (36 values at a time: 256 x 36 floats = 36 KB; `red` may alias the staged rows - the first barrier retires them);
// value j of channel group g goes to dst[index(g, j)]
constexpr int kConv0Chunk = 36;
template <int NVAL, typename Index>
__device__ __forceinline__ void conv0_block_reduce(const float (&part)[NVAL], float* red, int tx, int ty, int C4, int tys,
                                                   double* dst, Index index) {
#pragma unroll
    for (int j0 = 0; j0 < NVAL; j0 += kConv0Chunk) {
        constexpr int kAll = NVAL;
        const int n = (kAll - j0) < kConv0Chunk ? (kAll - j0) : kConv0Chunk;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < kConv0Chunk; ++j)
            if (j0 + j < NVAL) red[((size_t)ty * C4 + tx) * n + j] = part[j0 + j];
        __syncthreads();
        for (int i = threadIdx.x; i < C4 * n; i += blockDim.x) {
            double acc = 0.0;
            for (int t = 0; t < tys; ++t) acc += (double)red[(size_t)t * C4 * n + i];
            atomicAdd(dst + index(i / n, j0 + i % n), acc);
        }
    }
}

template <int CIN>
__global__ __launch_bounds__(256) void k_conv0_stats(const float* __restrict__ x, const float* __restrict__ w, int B, int H, int W,
                                                     int C4, double* sums) {
    extern __shared__ float c0mem[];
    const int tys = 256 / C4, tx = threadIdx.x % C4, ty = threadIdx.x / C4;
    float wreg[4][CIN][9];
    conv0_load_weights<CIN>(w, tx, wreg);
    conv0_stage_rows<CIN>(x, c0mem, blockIdx.x, H, W);
    __syncthreads();
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int wq = ty; wq < W; wq += tys) {
        const Conv0Taps<CIN> t = conv0_taps<CIN>(c0mem, wq, W);
        float z[4];
        conv0_z<CIN>(t, wreg, z);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            part[k] += z[k];
            part[4 + k] = fmaf(z[k], z[k], part[4 + k]);
        }
    }
    const int C = 4 * C4, slot = blockIdx.x % bn_slots(C);
    conv0_block_reduce<8>(part, c0mem, tx, ty, C4, tys, sums + (size_t)slot * 2 * C,
                          [=](int g, int j) { return (size_t)(j / 4) * C + 4 * g + (j & 3); });
}

template <int CIN>
__global__ __launch_bounds__(256) void k_conv0_bn_relu(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                       int B, int H, int W, int C4, double inv_m, double unbias, const double* sums,
                                                       const float* gamma, const float* beta, const float* conv_bias, float eps,
                                                       float momentum, float* running_mean, float* running_var, float* save_mean,
                                                       float* save_rstd) {
    extern __shared__ float c0mem[];  // [2][C] coefficients, formed exactly as k_bn_relu_apply forms them, then the staged rows
    const int C = 4 * C4;
    float* coef = c0mem;
    float* tile = c0mem + 2 * C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double m = bn_sum(sums, C, c) * inv_m, var = fmax(bn_sum(sums, C, C + c) * inv_m - m * m, 0.0);
        const float mu = (float)m, rs = (float)(1.0 / sqrt(var + (double)eps));
        coef[c] = gamma[c] * rs;
        coef[C + c] = beta[c] - mu * (gamma[c] * rs);
        if (blockIdx.x == 0) {
            save_mean[c] = mu;
            save_rstd[c] = rs;
            const float bias = conv_bias ? conv_bias[c] : 0.f;
            running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (mu + bias);
            running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
        }
    }
    const int tys = 256 / C4, tx = threadIdx.x % C4, ty = threadIdx.x / C4;
    float wreg[4][CIN][9];
    conv0_load_weights<CIN>(w, tx, wreg);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (unsigned row = blockIdx.x; row < (unsigned)B * H; row += gridDim.x) {
        __syncthreads();  // the previous row's taps are consumed (first pass: nothing)
        conv0_stage_rows<CIN>(x, tile, row, H, W);
        __syncthreads();
        const float4 sc = reinterpret_cast<const float4*>(coef)[tx], sh = reinterpret_cast<const float4*>(coef + C)[tx];
        for (int wq = ty; wq < W; wq += tys) {
            const Conv0Taps<CIN> t = conv0_taps<CIN>(tile, wq, W);
            float z[4];
            conv0_z<CIN>(t, wreg, z);
            float4 r;
            r.x = fmaxf(fmaf(z[0], sc.x, sh.x), 0.f);
            r.y = fmaxf(fmaf(z[1], sc.y, sh.y), 0.f);
            r.z = fmaxf(fmaf(z[2], sc.z, sh.z), 0.f);
            r.w = fmaxf(fmaf(z[3], sc.w, sh.w), 0.f);
            y4[((size_t)row * W + wq) * C4 + tx] = r;
        }
    }
}

// Inference form of the same layer (BatchNorm folded into weight and bias): y = max(z + bias, 0), channels-last.
template <int CIN>
__global__ __launch_bounds__(256) void k_conv0_bias_relu(const float* __restrict__ x, const float* __restrict__ w,
                                                         const float* __restrict__ bias, float* __restrict__ y, int B, int H,
                                                         int W, int C4) {
    extern __shared__ float c0mem[];
    const int tys = 256 / C4, tx = threadIdx.x % C4, ty = threadIdx.x / C4;
    float wreg[4][CIN][9];
    conv0_load_weights<CIN>(w, tx, wreg);
    const float4 bv = reinterpret_cast<const float4*>(bias)[tx];
    float4* y4 = reinterpret_cast<float4*>(y);
    for (unsigned row = blockIdx.x; row < (unsigned)B * H; row += gridDim.x) {
        __syncthreads();
        conv0_stage_rows<CIN>(x, c0mem, row, H, W);
        __syncthreads();
        for (int wq = ty; wq < W; wq += tys) {
            const Conv0Taps<CIN> t = conv0_taps<CIN>(c0mem, wq, W);
            float z[4];
            conv0_z<CIN>(t, wreg, z);
            y4[((size_t)row * W + wq) * C4 + tx] =
                make_float4(fmaxf(z[0] + bv.x, 0.f), fmaxf(z[1] + bv.y, 0.f), fmaxf(z[2] + bv.z, 0.f), fmaxf(z[3] + bv.w, 0.f));
        }
    }
}

// DW = false: sums[0][c] += g, sums[1][c] += g xhat.   DW = true: dw[co][ci][tap] += dz * tap (fp64 staging buffer).
template <int CIN, bool DW>
__global__ __launch_bounds__(256) void k_conv0_bwd(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ dy, int B, int H, int W, int C4, float inv_m,
                                                   const float* mean, const float* rstd, const float* gamma, const float* beta,
                                                   double* sums, double* dw, float* dgamma, float* dbeta) {
    extern __shared__ float c0mem[];
    const int C = 4 * C4;
    const int tys = 256 / C4, tx = threadIdx.x % C4, ty = threadIdx.x / C4;
    float wreg[4][CIN][9];
    conv0_load_weights<CIN>(w, tx, wreg);
    float mu[4], rs[4], a[4], bq[4], dq[4], hq[4];  // per channel: mean, rstd, a = gamma rstd, b, d (DW), shift
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = 4 * tx + k;
        mu[k] = mean[c];
        rs[k] = rstd[c];
        a[k] = gamma[c] * rs[k];
        hq[k] = beta[c] - mu[k] * a[k];
        bq[k] = dq[k] = 0.f;
        if constexpr (DW) {
            const float sg = (float)bn_sum(sums, C, c), sgx = (float)bn_sum(sums, C, C + c);
            bq[k] = -a[k] * rs[k] * sgx * inv_m;
            dq[k] = -a[k] * sg * inv_m - bq[k] * mu[k];
            if (blockIdx.x == 0 && ty == 0) {
                dbeta[c] = sg;
                dgamma[c] = sgx;
            }
        }
    }
    constexpr int NVAL = DW ? 4 * CIN * 9 : 8;
    float part[NVAL];
#pragma unroll
    for (int j = 0; j < NVAL; ++j) part[j] = 0.f;
    const float4* d4 = reinterpret_cast<const float4*>(dy);
    for (unsigned row = blockIdx.x; row < (unsigned)B * H; row += gridDim.x) {  // a few rows per block: fewer atomics per address
        __syncthreads();
        conv0_stage_rows<CIN>(x, c0mem, row, H, W);
        __syncthreads();
        for (int wq = ty; wq < W; wq += tys) {
            const float4 d = d4[((size_t)row * W + wq) * C4 + tx];
            const Conv0Taps<CIN> t = conv0_taps<CIN>(c0mem, wq, W);
            float z[4];
            conv0_z<CIN>(t, wreg, z);
            const float dd[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float g = fmaf(z[k], a[k], hq[k]) > 0.f ? dd[k] : 0.f;
                if constexpr (!DW) {
                    part[k] += g;
                    part[4 + k] = fmaf(g, (z[k] - mu[k]) * rs[k], part[4 + k]);
                } else {
                    const float dz = fmaf(a[k], g, fmaf(bq[k], z[k], dq[k]));
#pragma unroll
                    for (int c = 0; c < CIN; ++c)
#pragma unroll
                        for (int q = 0; q < 9; ++q) part[(k * CIN + c) * 9 + q] = fmaf(dz, t.v[c][q], part[(k * CIN + c) * 9 + q]);
                }
            }
        }
    }
    if constexpr (!DW) {
        const int slot = blockIdx.x % bn_slots(C);
        conv0_block_reduce<NVAL>(part, c0mem, tx, ty, C4, tys, sums + (size_t)slot * 2 * C,
                                 [=](int g, int j) { return (size_t)(j / 4) * C + 4 * g + (j & 3); });
    } else {
        // dw [C][CIN][9]: value j = (k CIN + ci) 9 + tap of channel group g -> ((4 g + k) CIN + ci) 9 + tap = 4 g CIN 9 + j
        // (kConv0DwSlots copies of dw, block % copies: the caller adds them up)
        conv0_block_reduce<NVAL>(part, c0mem, tx, ty, C4, tys, dw + (size_t)(blockIdx.x % kConv0DwSlots) * C * CIN * 9,
                                 [=](int g, int j) { return (size_t)g * 4 * CIN * 9 + j; });
    }
}

// ---- C ABI ----------------------------------------------------------------------------------------------------------
static int conv0_check(const void* a, const void* b, int batch, int cin, int cout, int height, int width, const char* who) {
    if (!a || !b) return fail(IRIS_E_INVALID, "%s: NULL argument", who);
    if (batch <= 0 || height <= 0 || width <= 0) return fail(IRIS_E_INVALID, "%s: empty tensor", who);
    if (cin != 1 && cin != 2) return fail(IRIS_E_UNSUPPORTED, "%s: %d input channels (1 or 2)", who, cin);
    if (cout <= 0 || (cout & 3) || cout > 256 || 256 % (cout / 4))
        return fail(IRIS_E_UNSUPPORTED, "%s: %d output channels (4, 8, 16, 32, 64, 128 or 256)", who, cout);
    if ((double)batch * height * width >= 2147483648.0) return fail(IRIS_E_UNSUPPORTED, "%s: more than 2^31 pixels", who);
    if (width > kConv0MaxW) return fail(IRIS_E_UNSUPPORTED, "%s: width %d > %d", who, width, kConv0MaxW);
    return IRIS_OK;
}
static size_t conv0_tile_bytes(int cin, int width) { return (size_t)cin * 3 * (width + 2) * sizeof(float); }

extern "C" int iris_conv3x3_small_bias_relu_nhwc(const float* x, const float* weight, const float* bias, float* y, int batch,
                                                 int in_channels, int out_channels, int height, int width, void* stream) {
    int rc = conv0_check(x, weight, batch, in_channels, out_channels, height, width, "iris_conv3x3_small_bias_relu_nhwc");
    if (rc) return rc;
    if (!bias || !y) return fail(IRIS_E_INVALID, "iris_conv3x3_small_bias_relu_nhwc: NULL argument");
    if ((reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(y)) & 15)
        return fail(IRIS_E_INVALID, "iris_conv3x3_small_bias_relu_nhwc: bias and y must be 16-byte aligned");
    const unsigned grid = (unsigned)std::min<size_t>((size_t)batch * height, 4096);
    const size_t lds = conv0_tile_bytes(in_channels, width);
    if (in_channels == 1)
        k_conv0_bias_relu<1><<<grid, 256, lds, (hipStream_t)stream>>>(x, weight, bias, y, batch, height, width, out_channels / 4);
    else
        k_conv0_bias_relu<2><<<grid, 256, lds, (hipStream_t)stream>>>(x, weight, bias, y, batch, height, width, out_channels / 4);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" size_t iris_conv0_dweight_len(int in_channels, int out_channels) {
    return (in_channels > 0 && out_channels > 0) ? (size_t)kConv0DwSlots * out_channels * in_channels * 9 : 0;
}

extern "C" int iris_conv0_stats(const float* x, const float* weight, int batch, int in_channels, int out_channels, int height,
                                int width, double* sums_zeroed, void* stream) {
    int rc = conv0_check(x, weight, batch, in_channels, out_channels, height, width, "iris_conv0_stats");
    if (rc) return rc;
    if (!sums_zeroed) return fail(IRIS_E_INVALID, "iris_conv0_stats: NULL argument");
    const int C4 = out_channels / 4;
    const size_t lds = std::max((size_t)256 * 8 * sizeof(float), conv0_tile_bytes(in_channels, width));
    const unsigned grid = (unsigned)batch * height;  // one block per image row
    if (in_channels == 1) k_conv0_stats<1><<<grid, 256, lds, (hipStream_t)stream>>>(x, weight, batch, height, width, C4, sums_zeroed);
    else k_conv0_stats<2><<<grid, 256, lds, (hipStream_t)stream>>>(x, weight, batch, height, width, C4, sums_zeroed);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_conv0_bn_relu(const float* x, const float* weight, float* y, int batch, int in_channels, int out_channels,
                                  int height, int width, const double* sums, const float* gamma, const float* beta,
                                  const float* conv_bias, float eps, float momentum, float* running_mean, float* running_var,
                                  float* save_mean, float* save_rstd, void* stream) {
    int rc = conv0_check(x, weight, batch, in_channels, out_channels, height, width, "iris_conv0_bn_relu");
    if (rc) return rc;
    if (!y || !sums || !gamma || !beta || !running_mean || !running_var || !save_mean || !save_rstd)
        return fail(IRIS_E_INVALID, "iris_conv0_bn_relu: NULL argument");
    const int C4 = out_channels / 4;
    const size_t n_pix = (size_t)batch * height * width;
    const double m = (double)n_pix;
    const unsigned grid = (unsigned)std::min<size_t>((size_t)batch * height, 4096);
    const size_t lds = 2 * (size_t)out_channels * sizeof(float) + conv0_tile_bytes(in_channels, width);
    if (in_channels == 1)
        k_conv0_bn_relu<1><<<grid, 256, lds, (hipStream_t)stream>>>(x, weight, y, batch, height, width, C4, 1.0 / m,
                                                                 m > 1.0 ? m / (m - 1.0) : 1.0, sums, gamma, beta, conv_bias, eps,
                                                                 momentum, running_mean, running_var, save_mean, save_rstd);
    else
        k_conv0_bn_relu<2><<<grid, 256, lds, (hipStream_t)stream>>>(x, weight, y, batch, height, width, C4, 1.0 / m,
                                                                 m > 1.0 ? m / (m - 1.0) : 1.0, sums, gamma, beta, conv_bias, eps,
                                                                 momentum, running_mean, running_var, save_mean, save_rstd);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_conv0_bn_relu_backward(const float* x, const float* weight, const float* dy, int batch, int in_channels,
                                           int out_channels, int height, int width, const float* save_mean, const float* save_rstd,
                                           const float* gamma, const float* beta, double* sums_zeroed, double* dweight_zeroed,
                                           float* dgamma, float* dbeta, void* stream) {
    int rc = conv0_check(x, weight, batch, in_channels, out_channels, height, width, "iris_conv0_bn_relu_backward");
    if (rc) return rc;
    if (!dy || !save_mean || !save_rstd || !gamma || !beta || !sums_zeroed || !dweight_zeroed || !dgamma || !dbeta)
        return fail(IRIS_E_INVALID, "iris_conv0_bn_relu_backward: NULL argument");
    const int C4 = out_channels / 4;
    const size_t n_pix = (size_t)batch * height * width;
    const unsigned grid = (unsigned)std::min<size_t>((size_t)batch * height, 1024);  // blocks walk image rows
    const float inv_m = (float)(1.0 / (double)n_pix);
    hipStream_t s = (hipStream_t)stream;
    const size_t tile = conv0_tile_bytes(in_channels, width);
    const size_t lds_r = std::max((size_t)256 * 8 * sizeof(float), tile), lds_w = std::max((size_t)256 * kConv0Chunk * sizeof(float), tile);
    if (in_channels == 1) {
        k_conv0_bwd<1, false><<<grid, 256, lds_r, s>>>(x, weight, dy, batch, height, width, C4, inv_m, save_mean, save_rstd, gamma, beta,
                                                     sums_zeroed, nullptr, nullptr, nullptr);
        k_conv0_bwd<1, true><<<grid, 256, lds_w, s>>>(x, weight, dy, batch, height, width, C4, inv_m, save_mean, save_rstd, gamma, beta,
                                                    sums_zeroed, dweight_zeroed, dgamma, dbeta);
    } else {
        k_conv0_bwd<2, false><<<grid, 256, lds_r, s>>>(x, weight, dy, batch, height, width, C4, inv_m, save_mean, save_rstd, gamma, beta,
                                                     sums_zeroed, nullptr, nullptr, nullptr);
        k_conv0_bwd<2, true><<<grid, 256, lds_w, s>>>(x, weight, dy, batch, height, width, C4, inv_m, save_mean, save_rstd, gamma, beta,
                                                    sums_zeroed, dweight_zeroed, dgamma, dbeta);
    }
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
