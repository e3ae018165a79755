// k_agc_adam.h -- adaptive gradient clipping + clipvalue + the Adam update of a whole model in ONE launch (round 6): the tail of the
// reference's train_step (sj_train.py:145-155 adaptive_clip_grad, :434-435 Adam(clipvalue), :176-182 apply_gradients).
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// k_agc_clip (k_elementwise.h) reads a unit's parameters and gradients for the two norms and rewrites the gradients; torch's fused
// Adam then reads parameters, gradients and both moments again through three multi-tensor launches at 1.9 TB/s (143 us for the
// CRNN's 9.9 M parameters; profiles/r6/c4_split0_step_kernel_stats.csv).  Here a wave keeps going after it has the unit's clip
// factor: gradient -> scaled, clamped, written back (p.grad still holds what the reference's optimiser would have been handed),
// moments and parameter updated in the same pass - 32 bytes of traffic per parameter instead of 52, one launch instead of four.
// Arithmetic = ATen's fused Adam (adam_math, ADAM_MODE ORIGINAL, fp32 op-math):
//     m += (g - m) (1 - beta1);   v = beta2 v + (1 - beta2) g g;
//     p -= (lr / (1 - beta1^t)) m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// with t read from the optimiser's own (already incremented) device-side step counter and lr from its device tensor when it has one
// (capturable optimisers: a replayed hipGraph sees every new value).  No weight decay, no amsgrad, no maximize: the caller (FusedAGC
// in hip_autograd.py) takes torch's path for anything else.
// ---------------------------------------------------------------------------

__global__ __launch_bounds__(256) void k_agc_clip_adam(const iris_agc_adam_row* rows, size_t n_rows, float clip_factor, float eps_agc,
                                                       float clipvalue, int use_agc, const float* lr_dev, float lr_host, double beta1d,
                                                       double beta2d, float eps, const float* step_dev) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t n_waves = (size_t)gridDim.x * 4;
    const double t = (double)step_dev[0];
    // (the betas stay doubles up to here, as in ATen: 1 - 0.999f is 1.3e-5 away from 1 - 0.999)
    const float bc1 = (float)(1.0 - pow(beta1d, t)), bc2_sqrt = sqrtf((float)(1.0 - pow(beta2d, t)));
    const float lr = lr_dev ? lr_dev[0] : lr_host;
    const float step_size = lr / bc1, w1 = (float)(1.0 - beta1d), w2 = (float)(1.0 - beta2d), beta2 = (float)beta2d;
    const bool clamp = clipvalue > 0.f;
    for (size_t r = wave; r < n_rows; r += n_waves) {
        float* const p = rows[r].param;
        float* const g = rows[r].grad;
        float* const m = rows[r].exp_avg;
        float* const v = rows[r].exp_avg_sq;
        const long len = rows[r].len;
        const bool vec = ((len & 3) == 0) && (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                                                reinterpret_cast<uintptr_t>(v)) & 15) == 0);
        float scale = 1.0f;
        if (use_agc) {   // the unit's two norms (k_agc_clip's first pass)
            float sp = 0.f, sg = 0.f;
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
            const float max_norm = fmaxf(p_norm, eps_agc) * clip_factor;
            scale = g_norm < max_norm ? 1.0f : max_norm / fmaxf(g_norm, 1e-6f);
        }
        auto one = [&](float& pp, float& gg, float& mm, float& vv) {
            // (scale == 1 leaves the gradient's bits alone, as k_agc_clip does by skipping the unit)
            float x = scale == 1.0f ? gg : gg * scale;
            if (clamp) x = fminf(fmaxf(x, -clipvalue), clipvalue);
            gg = x;
            mm = mm + (x - mm) * w1;
            vv = beta2 * vv + w2 * x * x;
            const float denom = sqrtf(vv) / bc2_sqrt + eps;
            pp -= step_size * mm / denom;
        };
        if (vec) {
            for (long i = 4 * lane; i < len; i += 4 * kWave) {
                float4 a = *reinterpret_cast<float4*>(p + i), b = *reinterpret_cast<float4*>(g + i);
                float4 c = *reinterpret_cast<float4*>(m + i), d = *reinterpret_cast<float4*>(v + i);
                one(a.x, b.x, c.x, d.x);
                one(a.y, b.y, c.y, d.y);
                one(a.z, b.z, c.z, d.z);
                one(a.w, b.w, c.w, d.w);
                *reinterpret_cast<float4*>(p + i) = a;
                *reinterpret_cast<float4*>(g + i) = b;
                *reinterpret_cast<float4*>(m + i) = c;
                *reinterpret_cast<float4*>(v + i) = d;
            }
        } else {
            for (long i = lane; i < len; i += kWave) one(p[i], g[i], m[i], v[i]);
        }
    }
}

extern "C" int iris_agc_clip_adam(const iris_agc_adam_row* rows_dev, size_t n_rows, float clip_factor, float eps_agc, float clipvalue,
                                  int use_agc, const float* lr_dev, float lr_host, double beta1, double beta2, float eps,
                                  const float* step_dev, void* stream) {
    if (n_rows == 0) return IRIS_OK;
    if (!rows_dev || !step_dev) return fail(IRIS_E_INVALID, "iris_agc_clip_adam: NULL argument");
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.f))
        return fail(IRIS_E_INVALID, "iris_agc_clip_adam: betas (%g, %g) / eps %g", beta1, beta2, (double)eps);
    const size_t blocks = std::min<size_t>((n_rows + 3) / 4, 8192);
    k_agc_clip_adam<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(rows_dev, n_rows, clip_factor, eps_agc, clipvalue, use_agc ? 1 : 0, lr_dev,
                                                                     lr_host, beta1, beta2, eps, step_dev);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
