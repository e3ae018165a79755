// k_lstm.h -- the recurrent half of the v9 CRNN's bidirectional LSTM(128 -> 128) for inference (sj_train.py:252).
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// MIOpen runs an LSTM as one GEMM + one pointwise kernel per time step and direction: 16 steps x 2 directions x 2 launches
// of 2-7 us each = 0.3 ms of a 4.4 ms forward pass for 0.3 GFLOP of work.  The recurrence is independent across the batch,
// so here a workgroup owns R batch rows of one direction and walks ALL time steps by itself - no grid-wide step barrier:
//   * the recurrent matrix W_hh [512 gate rows x 128] of its direction lives in REGISTERS for the whole sequence: thread j
//     of 512 holds row j (128 VGPRs; 8 waves per workgroup leave 256 per thread, nothing spills);
//   * per step: every thread forms  h_{t-1} . W_hh[j]  for the R rows (h broadcast from LDS), adds the input
//     pre-activation gx[b, t, d, j] (= x_t W_ih^T + b_ih + b_hh, ONE GEMM for all steps and both directions, done by the
//     caller), and after a barrier R x 128 threads apply the gates (i, f, g, o in torch's order), keep c in a register,
//     write h to LDS and to out[b, t, d * 128 + u].
// fp32 throughout.  Launch: grid (ceil(B / R), 2), 512 threads.
// ---------------------------------------------------------------------------
constexpr int kLstmH = 128;

// Gate functions on the hardware exp2 / reciprocal (1 ulp each): |error| <= 3e-7 on values in [-1, 1], and no library
// call inside a kernel whose threads each hold 64 weights in registers (expf / tanhf spilled them).
__device__ __forceinline__ float lstm_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * x));
}
__device__ __forceinline__ float lstm_tanh(float x) { return fmaf(2.0f, lstm_sigmoid(2.0f * x), -1.0f); }

template <int R, bool SAVE>
__global__ __launch_bounds__(512) void k_bilstm128_fwd(const float* __restrict__ gx, const float* __restrict__ w_hh,
                                                       float* __restrict__ out, float* __restrict__ act, int B, int T) {
    static_assert(R * kLstmH <= 512, "one gate thread per (row, unit)");
    __shared__ __attribute__((aligned(16))) float h_s[R][kLstmH];
    __shared__ float pre[R][4 * kLstmH];
    const int d = blockIdx.y, b0 = blockIdx.x * R;
    const int j = threadIdx.x;  // gate row: i 0..127, f 128..255, g 256..383, o 384..511
    float w[kLstmH];
    {
        const float4* wr = reinterpret_cast<const float4*>(w_hh + ((size_t)d * 4 * kLstmH + j) * kLstmH);
#pragma unroll
        for (int k = 0; k < kLstmH / 4; ++k) {
            const float4 v = wr[k];
            w[4 * k] = v.x;
            w[4 * k + 1] = v.y;
            w[4 * k + 2] = v.z;
            w[4 * k + 3] = v.w;
        }
    }
    float c = 0.f;  // cell state of (row j / 128, unit j % 128) for j < R * 128
    if (j < R * kLstmH) h_s[j >> 7][j & 127] = 0.f;
    __syncthreads();
    // per-row streams of input pre-activations and outputs, advanced by one time step per iteration
    const long step_g = (d ? -1L : 1L) * 2 * 4 * kLstmH, step_o = (d ? -1L : 1L) * 2 * kLstmH;
    const int t_first = d ? T - 1 : 0;
    const float* gp = gx + (((size_t)b0 * T + t_first) * 2 + d) * (4 * kLstmH) + j;
    float* op = out + ((size_t)b0 * T + t_first) * (2 * kLstmH) + d * kLstmH;
    // SAVE (training): the gate activations and the cell state of every step, [B, T, 2, 5, 128] = (i, f, g, o, c)
    const long step_a = (d ? -1L : 1L) * 2 * 5 * kLstmH;
    float* ap = SAVE ? act + (((size_t)b0 * T + t_first) * 2 + d) * (5 * kLstmH) : nullptr;
    for (int s = 0; s < T; ++s, gp += step_g, op += step_o, ap += SAVE ? step_a : 0) {
        float g[R], acc[R][2];
#pragma unroll
        for (int r = 0; r < R; ++r) {  // this step's input pre-activation of gate row j: in flight behind the dot product
            g[r] = (b0 + r < B) ? gp[(size_t)r * T * 2 * 4 * kLstmH] : 0.f;
            acc[r][0] = acc[r][1] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < kLstmH / 4; ++k) {
            if ((k & 3) == 0) asm volatile("" ::: "memory");  // a bounded window of h reads in flight
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 hv = *reinterpret_cast<const float4*>(&h_s[r][4 * k]);  // same address in every lane: broadcast
                acc[r][0] = fmaf(w[4 * k], hv.x, acc[r][0]);
                acc[r][1] = fmaf(w[4 * k + 1], hv.y, acc[r][1]);
                acc[r][0] = fmaf(w[4 * k + 2], hv.z, acc[r][0]);
                acc[r][1] = fmaf(w[4 * k + 3], hv.w, acc[r][1]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) pre[r][j] = (acc[r][0] + acc[r][1]) + g[r];
        __syncthreads();
        if (j < R * kLstmH) {
            const int r = j >> 7, u = j & 127;
            const float* pr = &pre[r][u];
            const float gi = lstm_sigmoid(pr[0]), gf = lstm_sigmoid(pr[kLstmH]);
            const float gg = lstm_tanh(pr[2 * kLstmH]), go = lstm_sigmoid(pr[3 * kLstmH]);
            c = fmaf(gf, c, gi * gg);
            const float h = go * lstm_tanh(c);
            h_s[r][u] = h;
            if (b0 + r < B) {
                op[(size_t)r * T * 2 * kLstmH + u] = h;
                if constexpr (SAVE) {
                    float* a = ap + (size_t)r * T * 2 * 5 * kLstmH + u;
                    a[0] = gi;
                    a[kLstmH] = gf;
                    a[2 * kLstmH] = gg;
                    a[3 * kLstmH] = go;
                    a[4 * kLstmH] = c;
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------
// Backward through time of the same recurrence.  A workgroup owns R batch rows of one direction and walks the steps in
// the reverse of the forward order.  Per step the R x 128 gate threads turn (dout_t + recurrent dh, recurrent dc) and the
// saved (i, f, g, o, c_t, c_{t-1}) into the four pre-activation gradients - written to dgx[b, t, d, :] (from which the caller
// gets dW_ih, db, dx and dW_hh by GEMMs) and to LDS - then all 512 threads form  dh_{t-1} = dgates . W_hh : thread (q, u)
// holds column u of gate block q of W_hh in registers (128 values) and produces one of the four partial sums per unit.
// ---------------------------------------------------------------------------
template <int R>
__global__ __launch_bounds__(512) void k_bilstm128_bwd(const float* __restrict__ dout, const float* __restrict__ act,
                                                       const float* __restrict__ w_hh, float* __restrict__ dgx, int B, int T) {
    __shared__ __attribute__((aligned(16))) float dg_s[R][4 * kLstmH];
    __shared__ float part[R][4][kLstmH];
    const int d = blockIdx.y, b0 = blockIdx.x * R;
    const int tid = threadIdx.x, q = tid >> 7, u = tid & 127;
    float wt[kLstmH];  // W_hh[d][q * 128 + k][u], k = 0..127
    {
        const float* wc = w_hh + ((size_t)d * 4 * kLstmH + q * kLstmH) * kLstmH + u;
#pragma unroll
        for (int k = 0; k < kLstmH; ++k) wt[k] = wc[(size_t)k * kLstmH];
    }
#pragma unroll
    for (int r = 0; r < R; ++r) part[r][q][u] = 0.f;  // no recurrent gradient behind the last forward step
    float dc_rec = 0.f;
    __syncthreads();
    // forward walked t_first -> t_last; backward walks t_last -> t_first
    const int t_last = d ? 0 : T - 1, dir = d ? 1 : -1;  // time increment of the BACKWARD walk
    for (int s = 0; s < T; ++s) {
        const int t = t_last + dir * s;
        if (tid < R * kLstmH) {
            const int r = tid >> 7;  // (u = tid & 127 as above; q == r here is a coincidence of the layout)
            float dpi = 0.f, dpf = 0.f, dpg = 0.f, dpo = 0.f;
            if (b0 + r < B) {
                const size_t bt = (size_t)(b0 + r) * T + t;
                const float* a = act + (bt * 2 + d) * (5 * kLstmH) + u;
                const float gi = a[0], gf = a[kLstmH], gg = a[2 * kLstmH], go = a[3 * kLstmH], ct = a[4 * kLstmH];
                const bool first = (s == T - 1);  // the first forward step: c_{t-1} = 0
                const float cprev = first ? 0.f : (a + (ptrdiff_t)dir * 2 * 5 * kLstmH)[4 * kLstmH];  // the step the forward pass came from
                const float dh = dout[bt * (2 * kLstmH) + d * kLstmH + u] +
                                 ((part[r][0][u] + part[r][1][u]) + (part[r][2][u] + part[r][3][u]));
                const float tc = lstm_tanh(ct);
                const float dc = fmaf(dh * go, 1.f - tc * tc, dc_rec);
                dpo = dh * tc * go * (1.f - go);
                dpi = dc * gg * gi * (1.f - gi);
                dpf = dc * cprev * gf * (1.f - gf);
                dpg = dc * gi * (1.f - gg * gg);
                dc_rec = dc * gf;
                float* o = dgx + (bt * 2 + d) * (4 * kLstmH) + u;
                o[0] = dpi;
                o[kLstmH] = dpf;
                o[2 * kLstmH] = dpg;
                o[3 * kLstmH] = dpo;
            }
            dg_s[r][u] = dpi;
            dg_s[r][kLstmH + u] = dpf;
            dg_s[r][2 * kLstmH + u] = dpg;
            dg_s[r][3 * kLstmH + u] = dpo;
        }
        __syncthreads();
        float acc[R][2];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r][0] = acc[r][1] = 0.f;
#pragma unroll
        for (int k = 0; k < kLstmH / 4; ++k) {
            if ((k & 3) == 0) asm volatile("" ::: "memory");
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 gv = *reinterpret_cast<const float4*>(&dg_s[r][q * kLstmH + 4 * k]);  // wave-uniform address
                acc[r][0] = fmaf(wt[4 * k], gv.x, acc[r][0]);
                acc[r][1] = fmaf(wt[4 * k + 1], gv.y, acc[r][1]);
                acc[r][0] = fmaf(wt[4 * k + 2], gv.z, acc[r][0]);
                acc[r][1] = fmaf(wt[4 * k + 3], gv.w, acc[r][1]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) part[r][q][u] = acc[r][0] + acc[r][1];  // (read by the gate threads before the barrier above)
        __syncthreads();
    }
}

extern "C" int iris_bilstm128_forward(const float* gx, const float* w_hh, float* out, float* act, int batch, int steps,
                                      void* stream) {
    if (!gx || !w_hh || !out) return fail(IRIS_E_INVALID, "iris_bilstm128_forward: NULL argument");
    if (batch <= 0 || steps <= 0) return fail(IRIS_E_INVALID, "iris_bilstm128_forward: batch=%d steps=%d must be positive", batch, steps);
    if ((reinterpret_cast<uintptr_t>(w_hh) & 15) || (reinterpret_cast<uintptr_t>(gx) & 3) || (reinterpret_cast<uintptr_t>(out) & 3))
        return fail(IRIS_E_INVALID, "iris_bilstm128_forward: w_hh must be 16-byte aligned");
    if (batch > 65535 * 2) return fail(IRIS_E_UNSUPPORTED, "iris_bilstm128_forward: batch %d > 131070", batch);
    constexpr int R = 2;
    const dim3 grid((batch + R - 1) / R, 2);
    if (act) k_bilstm128_fwd<R, true><<<grid, 512, 0, (hipStream_t)stream>>>(gx, w_hh, out, act, batch, steps);
    else k_bilstm128_fwd<R, false><<<grid, 512, 0, (hipStream_t)stream>>>(gx, w_hh, out, nullptr, batch, steps);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_bilstm128_backward(const float* dout, const float* act, const float* w_hh, float* dgx, int batch, int steps,
                                       void* stream) {
    if (!dout || !act || !w_hh || !dgx) return fail(IRIS_E_INVALID, "iris_bilstm128_backward: NULL argument");
    if (batch <= 0 || steps <= 0) return fail(IRIS_E_INVALID, "iris_bilstm128_backward: batch=%d steps=%d must be positive", batch, steps);
    if (batch > 65535 * 2) return fail(IRIS_E_UNSUPPORTED, "iris_bilstm128_backward: batch %d > 131070", batch);
    constexpr int R = 2;
    k_bilstm128_bwd<R><<<dim3((batch + R - 1) / R, 2), 512, 0, (hipStream_t)stream>>>(dout, act, w_hh, dgx, batch, steps);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
