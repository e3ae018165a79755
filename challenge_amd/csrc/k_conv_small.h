// k_conv_small.h -- the FIRST convolution of the CRNN (sj_train.py:191-201 with cin = n_chan): 3x3 'same', 1 or 2 input
// channels, bias + ReLU fused, NCHW.  Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// With one or two input channels the layer is a 9 / 18-tap stencil that WRITES 16-32x what it reads (c3: 8 MB in, 268 MB
// out): bound by the output stream, not by arithmetic.  MIOpen runs it as an implicit GEMM (60 us) followed - in the
// inference engine - by a separate bias + ReLU pass over the 268 MB (77 us).  Here a thread owns four consecutive pixels of
// one row: it loads the 3 x 6 input patch per input channel once, then walks the output channels (weights and bias are
// wave-uniform: scalar loads), 36 FMAs + bias + ReLU + ONE 16-byte store per channel; consecutive threads write consecutive
// 16-byte pieces of a channel plane.  One pass, every output byte written once.
//   x [B, CIN, H, W], w [COUT, CIN, 3, 3], bias [COUT], y [B, COUT, H, W]; W a multiple of 4.
// ---------------------------------------------------------------------------
template <int CIN>
__global__ __launch_bounds__(256) void k_conv3x3_small_bias_relu(const float* __restrict__ x, const float* __restrict__ w,
                                                                 const float* __restrict__ bias, float* __restrict__ y, int B,
                                                                 int H, int W, int COUT) {
    const int W4 = W >> 2;
    const unsigned total = (unsigned)B * H * W4;  // < 2^31 (host check)
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned row = i / (unsigned)W4, w4 = i - row * (unsigned)W4;  // row = b * H + h
        const unsigned b = row / (unsigned)H, h = row - b * (unsigned)H;
        const int w0 = (int)w4 * 4;
        float p[CIN][3][6];  // input patch: rows h-1..h+1, columns w0-1..w0+4, zero outside the image
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int hh = (int)h + dy - 1;
                const bool row_ok = hh >= 0 && hh < H;
                const float* xr = x + (((size_t)b * CIN + c) * H + (row_ok ? hh : 0)) * W;
                const float4 mid = row_ok ? *reinterpret_cast<const float4*>(xr + w0) : make_float4(0.f, 0.f, 0.f, 0.f);
                p[c][dy][0] = (row_ok && w0 > 0) ? xr[w0 - 1] : 0.f;
                p[c][dy][1] = mid.x;
                p[c][dy][2] = mid.y;
                p[c][dy][3] = mid.z;
                p[c][dy][4] = mid.w;
                p[c][dy][5] = (row_ok && w0 + 4 < W) ? xr[w0 + 4] : 0.f;
            }
        }
        float* yo = y + ((size_t)b * COUT * H + h) * W + w0;
        for (int co = 0; co < COUT; ++co) {  // wave-uniform: the 9 CIN weights and the bias come through scalar loads
            const float* wk = w + (size_t)co * CIN * 9;
            const float bv = bias[co];
            float4 acc = make_float4(bv, bv, bv, bv);
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float wv = wk[(c * 3 + dy) * 3 + dx];
                        acc.x = fmaf(wv, p[c][dy][dx], acc.x);
                        acc.y = fmaf(wv, p[c][dy][dx + 1], acc.y);
                        acc.z = fmaf(wv, p[c][dy][dx + 2], acc.z);
                        acc.w = fmaf(wv, p[c][dy][dx + 3], acc.w);
                    }
                }
            }
            acc.x = fmaxf(acc.x, 0.f);
            acc.y = fmaxf(acc.y, 0.f);
            acc.z = fmaxf(acc.z, 0.f);
            acc.w = fmaxf(acc.w, 0.f);
            *reinterpret_cast<float4*>(yo + (size_t)co * H * W) = acc;
        }
    }
}

extern "C" int iris_conv3x3_small_bias_relu_nchw(const float* x, const float* weight, const float* bias, float* y, int batch,
                                                 int in_channels, int out_channels, int height, int width, void* stream) {
    if (!x || !weight || !bias || !y) return fail(IRIS_E_INVALID, "iris_conv3x3_small_bias_relu_nchw: NULL argument");
    if (batch <= 0 || height <= 0 || width <= 0 || out_channels <= 0)
        return fail(IRIS_E_INVALID, "iris_conv3x3_small_bias_relu_nchw: empty tensor");
    if (in_channels != 1 && in_channels != 2)
        return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_small_bias_relu_nchw: %d input channels (1 or 2)", in_channels);
    if (width & 3) return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_small_bias_relu_nchw: width %d must be a multiple of 4", width);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15)
        return fail(IRIS_E_INVALID, "iris_conv3x3_small_bias_relu_nchw: x and y must be 16-byte aligned");
    const size_t total = (size_t)batch * height * (width / 4);
    if (total >= 2147483648ull) return fail(IRIS_E_UNSUPPORTED, "iris_conv3x3_small_bias_relu_nchw: more than 2^31 pixel groups");
    const int grid = (int)std::min<size_t>((total + 255) / 256, 8192);
    if (in_channels == 1)
        k_conv3x3_small_bias_relu<1><<<grid, 256, 0, (hipStream_t)stream>>>(x, weight, bias, y, batch, height, width, out_channels);
    else
        k_conv3x3_small_bias_relu<2><<<grid, 256, 0, (hipStream_t)stream>>>(x, weight, bias, y, batch, height, width, out_channels);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
