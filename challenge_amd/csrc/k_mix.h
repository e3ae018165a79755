// k_mix.h -- batched sample synthesis (merge_complex_specs) kernels and entry point.
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// batched sample synthesis (merge_complex_specs, pipeline.py:6-110)
// ---------------------------------------------------------------------------
// frame t of the output -> frame of the (virtually zero-padded) source, or -1 inside the padding
__device__ __forceinline__ int mix_frame(const iris_mix_src& s, int t) {
    const int fr = s.off + t - s.pad;
    return (fr >= 0 && fr < s.T) ? fr : -1;
}

// active[t] = 1 when max over (freq, chan2) of frame t is > 0 (pipeline.py:57).  A property of the
// source: computed once per corpus.  64 frames x 4 bin groups per block, blockIdx.y splits the bins
// further; loads are coalesced along t; groups combine through LDS, blocks through an atomic max on
// the flag's bit pattern (1.0f > 0.0f as integers; `active` is zeroed by the caller).
__global__ __launch_bounds__(256) void k_mix_frame_active(const float* src, int n_bins, int T, int chan2,
                                                          float* active) {
    __shared__ float part[4][64];
    const int tx = threadIdx.x & 63, fg = threadIdx.x >> 6;
    const int t = blockIdx.x * 64 + tx;
    const int groups = 4 * gridDim.y, g = blockIdx.y * 4 + fg;
    float mx = -INFINITY;
    if (t < T) {
        for (int f = g; f < n_bins; f += groups) {
            const float* p = src + ((size_t)f * T + t) * chan2;
            for (int c = 0; c < chan2; ++c) mx = fmaxf(mx, p[c]);
        }
    }
    part[fg][tx] = mx;
    __syncthreads();
    if (fg == 0 && t < T) {
        mx = fmaxf(fmaxf(part[0][tx], part[1][tx]), fmaxf(part[2][tx], part[3][tx]));
        if (mx > 0.f) atomicMax(reinterpret_cast<int*>(active) + t, __float_as_int(1.0f));
    }
}

// One block per sample: voices are accepted in slot order unless their labels would overlap the
// labels accepted so far (pipeline.py:72-84); writes the sample's label planes and one flag per voice.
__global__ __launch_bounds__(256) void k_mix_labels(const iris_mix_src* srcs, const int32_t* first,
                                                    const float* label_vecs, float* flags, float* labels,
                                                    int n_frame, int max_voices, int n_classes) {
    extern __shared__ float lsum[];  // [n_frame][n_classes] labels accepted so far, summed over voices
    const int b = blockIdx.x, plane = n_frame * n_classes;
    float* lab = labels + (size_t)b * max_voices * plane;
    for (int i = threadIdx.x; i < max_voices * plane; i += blockDim.x) lab[i] = 0.f;
    for (int i = threadIdx.x; i < plane; i += blockDim.x) lsum[i] = 0.f;
    __syncthreads();
    for (int si = first[b]; si < first[b + 1]; ++si) {
        const iris_mix_src s = srcs[si];
        if (s.kind != 1) continue;  // uniform (background, noise, or an unused slot of a fixed-stride table: kind -1)
        const float* lv = label_vecs + (size_t)s.label_row * n_classes;
        auto act = [&](int t) {  // frame t of the output lies in the padding, or in a silent / active frame
            const int fr = mix_frame(s, t);
            return fr >= 0 ? s.active[fr] : 0.f;
        };
        int over = 0;
        for (int i = threadIdx.x; i < plane; i += blockDim.x) {
            const int t = i / n_classes, c = i - t * n_classes;
            over |= (lsum[i] + lv[c] * act(t)) >= 2.f;
        }
        over = __syncthreads_or(over);
        if (threadIdx.x == 0) flags[si] = over ? 0.f : 1.f;
        if (!over && s.slot >= 0 && s.slot < max_voices) {
            for (int i = threadIdx.x; i < plane; i += blockDim.x) {
                const int t = i / n_classes, c = i - t * n_classes;
                const float l = lv[c] * act(t);
                lsum[i] += l;
                lab[(size_t)s.slot * plane + i] = l;
            }
        }
        __syncthreads();
    }
}

// spec_out[b, f, t, :] = background + accepted voices + noises, added in table order with
// separately rounded multiply and add (no FMA contraction: equals the op-by-op reference)
template <int C2>
__global__ __launch_bounds__(256) void k_mix_sum(const iris_mix_src* srcs, const int32_t* first, const float* flags,
                                                 float* out, int n_bins, int n_frame) {
#pragma clang fp contract(off)  // gain * x is rounded before it is added, as in the op-by-op reference
    typedef float vecT __attribute__((ext_vector_type(C2)));
    const int b = blockIdx.z, f = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n_frame) return;
    vecT acc = vecT(0.f);
    for (int si = first[b]; si < first[b + 1]; ++si) {
        const iris_mix_src s = srcs[si];  // uniform
        if (s.kind < 0) continue;  // unused slot of a fixed-stride table (iris_mix_draw)
        if (s.kind == 0) {
            const int fr = (s.off + t) % s.T;
            acc = *reinterpret_cast<const vecT*>(s.src + ((size_t)f * s.T + fr) * C2);
            continue;
        }
        const float keep = s.kind == 1 ? flags[si] : 1.f;
        const int fr = mix_frame(s, t);
        if (fr < 0 || keep == 0.f) continue;  // adds exactly zero
        const vecT v = *reinterpret_cast<const vecT*>(s.src + ((size_t)f * s.T + fr) * C2);
#pragma unroll
        for (int c = 0; c < C2; ++c) {
            const float scaled = s.gain * v[c];
            acc[c] = acc[c] + scaled;
        }
    }
    *reinterpret_cast<vecT*>(out + (((size_t)b * n_bins + f) * n_frame + t) * C2) = acc;
}

extern "C" int iris_mix_frame_active(const float* src, int n_bins, int n_frames, int chan2, float* active_out,
                                     void* stream) {
    if (!src || !active_out) return fail(IRIS_E_INVALID, "iris_mix_frame_active: NULL argument");
    if (n_bins <= 0 || n_frames <= 0 || chan2 <= 0)
        return fail(IRIS_E_INVALID, "iris_mix_frame_active: bad sizes (%d bins, %d frames, %d chan2)", n_bins, n_frames,
                    chan2);
    HIP_TRY(hipMemsetAsync(active_out, 0, (size_t)n_frames * sizeof(float), (hipStream_t)stream));
    const int ysplit = std::max(1, std::min(16, n_bins / 16));
    k_mix_frame_active<<<dim3((n_frames + 63) / 64, ysplit), 256, 0, (hipStream_t)stream>>>(src, n_bins, n_frames, chan2,
                                                                                          active_out);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" size_t iris_mix_workspace(int n_srcs, int n_frame) {
    (void)n_frame;
    return (size_t)std::max(n_srcs, 0);  // one accept flag per source
}

extern "C" int iris_mix_specs(const iris_mix_src* srcs_dev, int n_srcs, const int32_t* first_dev,
                              const float* label_vecs_dev, float* spec_out, float* labels_out, int batch,
                              int n_bins, int n_frame, int chan2, int max_voices, int n_classes, float* workspace,
                              size_t workspace_floats, void* stream) {
    if (!srcs_dev || !first_dev || !label_vecs_dev || !spec_out || !labels_out || !workspace)
        return fail(IRIS_E_INVALID, "iris_mix_specs: NULL argument");
    if (batch <= 0 || n_srcs < batch || n_bins <= 0 || n_frame <= 0 || max_voices <= 0 || n_classes <= 0)
        return fail(IRIS_E_INVALID, "iris_mix_specs: bad sizes (batch %d, %d sources, %d bins, %d frames)", batch,
                    n_srcs, n_bins, n_frame);
    if (chan2 != 1 && chan2 != 2 && chan2 != 4 && chan2 != 8)
        return fail(IRIS_E_UNSUPPORTED, "iris_mix_specs: chan2 %d (1, 2, 4 or 8)", chan2);
    if (batch > 65535 || n_bins > 65535 || n_srcs > 65535)
        return fail(IRIS_E_UNSUPPORTED, "iris_mix_specs: batch / bins / sources above 65535");
    if (workspace_floats < iris_mix_workspace(n_srcs, n_frame))
        return fail(IRIS_E_CAPACITY, "iris_mix_specs: workspace %zu floats < %zu", workspace_floats,
                    iris_mix_workspace(n_srcs, n_frame));
    const size_t lds = (size_t)n_frame * n_classes * sizeof(float);
    if (lds > 64 * 1024) return fail(IRIS_E_UNSUPPORTED, "iris_mix_specs: n_frame x n_classes too large for the LDS");
    hipStream_t s = (hipStream_t)stream;
    float* flags = workspace;  // [n_srcs]
    const unsigned tblocks = (unsigned)((n_frame + 255) / 256);
    k_mix_labels<<<batch, 256, lds, s>>>(srcs_dev, first_dev, label_vecs_dev, flags, labels_out, n_frame, max_voices,
                                         n_classes);
    const dim3 grid(tblocks, n_bins, batch);
    switch (chan2) {
        case 1: k_mix_sum<1><<<grid, 256, 0, s>>>(srcs_dev, first_dev, flags, spec_out, n_bins, n_frame); break;
        case 2: k_mix_sum<2><<<grid, 256, 0, s>>>(srcs_dev, first_dev, flags, spec_out, n_bins, n_frame); break;
        case 4: k_mix_sum<4><<<grid, 256, 0, s>>>(srcs_dev, first_dev, flags, spec_out, n_bins, n_frame); break;
        default: k_mix_sum<8><<<grid, 256, 0, s>>>(srcs_dev, first_dev, flags, spec_out, n_bins, n_frame); break;
    }
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// ---------------------------------------------------------------------------
// waveform-domain variant: same table, same label kernel, sources are [C, len]
// waveforms (len in `reserved`), T / pad / off in frames of `hop` samples
// ---------------------------------------------------------------------------
// active[t] = 1 when any sample under the support of frame t's periodic-Hann window is non-zero in any channel
__global__ __launch_bounds__(256) void k_mix_wave_frame_active(const float* wav, int channels, int len, int n_fft,
                                                               int hop, int n_frames, float* active) {
    const int t = blockIdx.x;  // one block per frame: the window is n_fft - 1 samples long
    const int lo = max(t * hop - n_fft / 2 + 1, 0), hi = min(t * hop + n_fft / 2 - 1, len - 1);
    int any = 0;
    for (int c = 0; c < channels; ++c)
        for (int i = lo + (int)threadIdx.x; i <= hi; i += blockDim.x) any |= wav[(size_t)c * len + i] != 0.f;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0 && t < n_frames) active[t] = any ? 1.f : 0.f;
}

// wav_out[b, c, s] = background + accepted voices + noises, table order, separately rounded multiply and add.
// Four consecutive samples per thread: 16-byte loads and stores where a source's four samples are contiguous,
// aligned and inside the clip (hop-granular offsets make that the rule), element by element otherwise.
__global__ __launch_bounds__(256) void k_mix_wave_sum(const iris_mix_src* srcs, const int32_t* first, const float* flags,
                                                      float* out, int channels, int hop, int out_len) {
#pragma clang fp contract(off)
    const int b = blockIdx.z, c = blockIdx.y, s = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (s >= out_len) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int si = first[b]; si < first[b + 1]; ++si) {
        const iris_mix_src r = srcs[si];  // uniform
        if (r.kind < 0) continue;  // unused slot of a fixed-stride table (iris_mix_draw)
        const int len = r.reserved;
        const float* row = r.src + (size_t)c * len;
        const bool row16 = (reinterpret_cast<uintptr_t>(row) & 15) == 0;  // (uniform) this channel's row starts on 16 bytes
        if (r.kind == 0) {
            const int p0 = (int)(((long long)r.off * hop + s) % len);
            if (row16 && (p0 & 3) == 0 && p0 + 3 < len) {
                const float4 v = *reinterpret_cast<const float4*>(row + p0);
                acc[0] = v.x; acc[1] = v.y; acc[2] = v.z; acc[3] = v.w;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = row[(p0 + j) % len];
            }
            continue;
        }
        const float keep = r.kind == 1 ? flags[si] : 1.f;
        const long long ss = (long long)s + (long long)(r.off - r.pad) * hop;
        if (ss + 3 < 0 || ss >= len || keep == 0.f) continue;  // adds exactly zero
        if (row16 && ss >= 0 && ss + 3 < len && (ss & 3) == 0) {
            const float4 v = *reinterpret_cast<const float4*>(row + ss);
            const float sc[4] = {r.gain * v.x, r.gain * v.y, r.gain * v.z, r.gain * v.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] + sc[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (ss + j >= 0 && ss + j < len) {
                    const float scaled = r.gain * row[ss + j];
                    acc[j] = acc[j] + scaled;
                }
        }
    }
    float* o = out + ((size_t)b * channels + c) * out_len + s;
    if (s + 3 < out_len && (reinterpret_cast<uintptr_t>(o) & 15) == 0) {
        *reinterpret_cast<float4*>(o) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
        for (int j = 0; j < 4 && s + j < out_len; ++j) o[j] = acc[j];
    }
}

extern "C" int iris_mix_wave_frame_active(const float* wav, int channels, int len, int n_fft, int hop, float* active_out,
                                          void* stream) {
    if (!wav || !active_out) return fail(IRIS_E_INVALID, "iris_mix_wave_frame_active: NULL argument");
    if (channels <= 0 || len <= 0 || n_fft <= 1 || hop <= 0)
        return fail(IRIS_E_INVALID, "iris_mix_wave_frame_active: bad sizes (%d channels, %d samples, n_fft %d, hop %d)",
                    channels, len, n_fft, hop);
    const int n_frames = 1 + len / hop;
    k_mix_wave_frame_active<<<n_frames, 256, 0, (hipStream_t)stream>>>(wav, channels, len, n_fft, hop, n_frames, active_out);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_mix_waves(const iris_mix_src* srcs_dev, int n_srcs, const int32_t* first_dev,
                              const float* label_vecs_dev, float* wav_out, float* labels_out, int batch, int channels,
                              int hop, int n_frame, int max_voices, int n_classes, float* workspace,
                              size_t workspace_floats, void* stream) {
    if (!srcs_dev || !first_dev || !label_vecs_dev || !wav_out || !labels_out || !workspace)
        return fail(IRIS_E_INVALID, "iris_mix_waves: NULL argument");
    if (batch <= 0 || n_srcs < batch || channels <= 0 || hop <= 0 || n_frame <= 1 || max_voices <= 0 || n_classes <= 0)
        return fail(IRIS_E_INVALID, "iris_mix_waves: bad sizes (batch %d, %d sources, %d channels, hop %d, %d frames)",
                    batch, n_srcs, channels, hop, n_frame);
    if (batch > 65535 || channels > 65535 || n_srcs > 65535)
        return fail(IRIS_E_UNSUPPORTED, "iris_mix_waves: batch / channels / sources above 65535");
    if ((long long)(n_frame - 1) * hop > 0x7fffffffLL) return fail(IRIS_E_UNSUPPORTED, "iris_mix_waves: output too long");
    if (workspace_floats < iris_mix_workspace(n_srcs, n_frame))
        return fail(IRIS_E_CAPACITY, "iris_mix_waves: workspace %zu floats < %zu", workspace_floats,
                    iris_mix_workspace(n_srcs, n_frame));
    const size_t lds = (size_t)n_frame * n_classes * sizeof(float);
    if (lds > 64 * 1024) return fail(IRIS_E_UNSUPPORTED, "iris_mix_waves: n_frame x n_classes too large for the LDS");
    hipStream_t s = (hipStream_t)stream;
    float* flags = workspace;  // [n_srcs]
    const int out_len = (n_frame - 1) * hop;
    k_mix_labels<<<batch, 256, lds, s>>>(srcs_dev, first_dev, label_vecs_dev, flags, labels_out, n_frame, max_voices,
                                         n_classes);
    k_mix_wave_sum<<<dim3((unsigned)((out_len + 1023) / 1024), channels, batch), 256, 0, s>>>(srcs_dev, first_dev, flags,
                                                                                          wav_out, channels, hop, out_len);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
