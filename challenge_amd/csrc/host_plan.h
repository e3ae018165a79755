// host_plan.h -- host side: mel weight matrix, constant tables, plan create / destroy.
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// host: mel matrix (fp32 recipe of tf.signal.linear_to_mel_weight_matrix)
// ---------------------------------------------------------------------------
// All fp32, one rounding per operation (no FMA contraction); the logarithm is the
// correctly rounded fp32 one (evaluated in double, rounded once).
#pragma clang fp contract(off)
static inline float hz_to_mel(float hz) {
    const float arg = 1.0f + hz / 700.0f;
    const float ln = (float)log((double)arg);
    return 1127.0f * ln;
}

static void linspace_f32(float start, float stop, int num, std::vector<float>& out) {
    out.resize(num);
    if (num == 1) {
        out[0] = start;
        return;
    }
    const float step = (stop - start) / (float)(num - 1);
    for (int i = 0; i < num; ++i) out[i] = start + step * (float)i;
    out[num - 1] = stop;
}

extern "C" int iris_mel_weight_matrix(int n_mel, int n_bins, float sample_rate, float lower_hz, float upper_hz,
                                      float* out) {
    if (!out) return fail(IRIS_E_INVALID, "iris_mel_weight_matrix: out is NULL");
    if (n_mel <= 0) return fail(IRIS_E_INVALID, "num_mel_bins must be positive");
    if (n_bins < 2) return fail(IRIS_E_INVALID, "num_spectrogram_bins must be >= 2");
    if (!(sample_rate > 0.f)) return fail(IRIS_E_INVALID, "sample_rate must be positive");
    if (lower_hz < 0.f) return fail(IRIS_E_INVALID, "lower_edge_hertz must be non-negative");
    if (!(lower_hz < upper_hz)) return fail(IRIS_E_INVALID, "lower_edge_hertz must be < upper_edge_hertz");
    if (upper_hz > sample_rate / 2.f) return fail(IRIS_E_INVALID, "upper_edge_hertz must not exceed Nyquist");
    std::vector<float> lin, edges;
    linspace_f32(0.f, sample_rate / 2.0f, n_bins, lin);
    linspace_f32(hz_to_mel(lower_hz), hz_to_mel(upper_hz), n_mel + 2, edges);
    for (int m = 0; m < n_mel; ++m) out[m] = 0.f;  // DC bin
    for (int f = 1; f < n_bins; ++f) {
        const float mel = hz_to_mel(lin[f]);
        for (int m = 0; m < n_mel; ++m) {
            const float lo = edges[m], ctr = edges[m + 1], hi = edges[m + 2];
            const float up = (mel - lo) / (ctr - lo);
            const float dn = (hi - mel) / (hi - ctr);
            out[(size_t)f * n_mel + m] = fmaxf(0.f, fminf(up, dn));
        }
    }
    return IRIS_OK;
}

// ---------------------------------------------------------------------------
// host: plan
// ---------------------------------------------------------------------------
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && hipSetDevice(dev) == hipSuccess) ok = true;
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

static int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

static int fft_ntw(int log2n) {
    switch (log2n) {
        case 11: return FftCfg<11>::NTW;
        case 10: return FftCfg<10>::NTW;
        case 9: return FftCfg<9>::NTW;
        default: return FftCfg<8>::NTW;
    }
}
static int fft_p(int log2n) { return (1 << log2n) / 2 / 64; }
static int const_nv4(int log2n) {
    switch (log2n) {
        case 11: return ConstLayout<11>::NV4;
        case 10: return ConstLayout<10>::NV4;
        case 9: return ConstLayout<9>::NV4;
        default: return ConstLayout<8>::NV4;
    }
}
static size_t wave_buf_bytes(int log2n) {
    const int NC = (1 << log2n) / 2;
    switch (log2n) {
        case 11: return (size_t)lds_padded(NC, FftCfg<11>::PMMAX) * 8;
        case 10: return (size_t)lds_padded(NC, FftCfg<10>::PMMAX) * 8;
        case 9: return (size_t)lds_padded(NC, FftCfg<9>::PMMAX) * 8;
        default: return (size_t)lds_padded(NC, FftCfg<8>::PMMAX) * 8;
    }
}

static void build_tables(int log2n, std::vector<float2>& tw, std::vector<float2>& post, std::vector<float2>& win) {
    const int N = 1 << log2n, NC = N / 2, P = fft_p(log2n);
    const double two_pi = 6.283185307179586476925286766559;
    std::vector<int> radices;
    if (log2n == 11) radices = {16, 16, 4};
    else if (log2n == 10) radices = {8, 8, 8};
    else if (log2n == 9) radices = {4, 4, 4, 4};
    else radices = {2, 2, 2, 2, 2, 2, 2};
    tw.clear();
    int ns = 1;
    for (size_t s = 0; s < radices.size(); ++s) {
        const int R = radices[s], U = P / R;
        if (s > 0) {
            for (int u = 0; u < U; ++u)
                for (int t = 1; t < R; ++t)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int bfly = lane + 64 * u;
                        const double ang = -two_pi * (double)((bfly % ns) * t) / (double)(ns * R);
                        tw.push_back(make_float2((float)cos(ang), (float)sin(ang)));
                    }
        }
        ns *= R;
    }
    post.clear();
    for (int q = 0; q < P / 2; ++q)
        for (int lane = 0; lane < 64; ++lane) {
            const double ang = -two_pi * (double)(lane + 64 * q) / (double)N;
            post.push_back(make_float2((float)cos(ang), (float)sin(ang)));
        }
    win.clear();
    for (int q = 0; q < P; ++q)
        for (int lane = 0; lane < 64; ++lane) {
            const int n = 2 * (lane + 64 * q);
            const double w0 = 0.5 - 0.5 * cos(two_pi * (double)n / (double)N);
            const double w1 = 0.5 - 0.5 * cos(two_pi * (double)(n + 1) / (double)N);
            win.push_back(make_float2((float)w0, (float)w1));
        }
    (void)NC;
}

template <typename T>
static int upload(T** dst, const std::vector<T>& src) {
    HIP_TRY(hipMalloc((void**)dst, std::max<size_t>(src.size(), 1) * sizeof(T)));
    if (!src.empty()) HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return IRIS_OK;
}

// two frames in flight per wave (IRIS_STREAMS=2) was measured slower twice: it exists in diagnostic builds only
static int plan_streams(const iris_plan* p) {
#if IRIS_DIAG
    return (p->streams == 2 && (p->log2n == 9 || p->log2n == 10)) ? 2 : 1;
#else
    (void)p;
    return 1;
#endif
}

// LDS of the fused kernel: landing + exchange buffers of every wave (the constant block is
// staged through the exchange area once), the frame queue, the MELMODE 1 tables
static size_t fused_lds_bytes(const iris_plan* p, int streams, bool bands, int chunk_frames = 0, int fuse = 0) {
    const size_t xbuf = (wave_buf_bytes(p->log2n) + 15) & ~(size_t)15;
    const size_t waves = (size_t)fused_waves(p->log2n, streams, bands, p->need_hi != 0, fuse, p->mel_mode) * streams;
    const size_t stage = (size_t)const_nv4(p->log2n) * 64 * 16;
    const size_t land = fused_direct(p->log2n) ? ((stage + 15) & ~(size_t)15) : waves * (size_t)p->n_fft * 4;
    size_t bytes = land + std::max(waves * xbuf, stage) + 16;
    if (p->mel_mode == 1) bytes += ((size_t)p->rows * p->n_mel + p->n_mel) * 4;
    // time-band bitmap of a chunk, written 2 words per wave per pass over the chunk's frames
    const size_t pass = (size_t)fused_waves(p->log2n, streams, bands, p->need_hi != 0, fuse, p->mel_mode) * 64;
    bytes += (((size_t)chunk_frames + pass - 1) / pass * pass / 32 + 2) * 4;
    return bytes;
}
// fused epilogue: mel tile [M][pitch] (+ the workgroup's reduction scratch) behind everything else, 16-byte aligned
static int fused_tile_pitch(const iris_plan* p, int chunk_frames) { return (chunk_frames * p->channels) | 1; }
static size_t fused_tile_off(size_t lds_without_tile) { return (lds_without_tile + 15) & ~(size_t)15; }
// (fuse 2 - the in-place form - keeps only the reduction scratch there)
static size_t fused_tile_bytes(const iris_plan* p, int streams, bool bands, int chunk_frames, int fuse = 1) {
    const size_t waves = (size_t)fused_waves(p->log2n, streams, bands, p->need_hi != 0, fuse, p->mel_mode);
    return ((fuse == 1 ? (size_t)p->n_mel * fused_tile_pitch(p, chunk_frames) : 0) + 2 * waves + 4) * 4;
}

typedef void (*fused_kernel_t)(const FusedArgs);

template <int LOG2N, int MELMODE, int S, int FUSE>
static fused_kernel_t fused_kernel_hb(bool hi, bool bands) {
    if (hi)
        return bands ? k_wav_to_mel<LOG2N, MELMODE, true, true, S, FUSE> : k_wav_to_mel<LOG2N, MELMODE, true, false, S, FUSE>;
    return bands ? k_wav_to_mel<LOG2N, MELMODE, false, true, S, FUSE> : k_wav_to_mel<LOG2N, MELMODE, false, false, S, FUSE>;
}
template <int LOG2N, int S, int FUSE>
static fused_kernel_t fused_kernel_mm(int mel_mode, bool hi, bool bands) {
    if constexpr (LOG2N <= 10) {
        if (mel_mode == 0)  // register weights exist only for the half-spectrum variant up to n_fft 1024
            return bands ? k_wav_to_mel<LOG2N, 0, false, true, S, FUSE> : k_wav_to_mel<LOG2N, 0, false, false, S, FUSE>;
        if (mel_mode == 3)
            return bands ? k_wav_to_mel<LOG2N, 3, false, true, S, FUSE> : k_wav_to_mel<LOG2N, 3, false, false, S, FUSE>;
    }
    if (mel_mode == 1) return fused_kernel_hb<LOG2N, 1, S, FUSE>(hi, bands);
    return fused_kernel_hb<LOG2N, 2, S, FUSE>(hi, bands);
}
// two frame streams per wave exist for n_fft 512 / 1024, in diagnostic builds only
template <int LOG2N>
static fused_kernel_t fused_kernel_m(int mel_mode, bool hi, bool bands, int streams, int fuse) {
#if IRIS_DIAG
    if constexpr (LOG2N == 9 || LOG2N == 10) {
        if (streams == 2) return fused_kernel_mm<LOG2N, 2, 0>(mel_mode, hi, bands);
    }
#endif
    (void)streams;
    if (fuse == 2) return fused_kernel_mm<LOG2N, 1, 2>(mel_mode, hi, bands);
    return fuse ? fused_kernel_mm<LOG2N, 1, 1>(mel_mode, hi, bands) : fused_kernel_mm<LOG2N, 1, 0>(mel_mode, hi, bands);
}
// fuse: 0 = raw mel + per-wave partials (two-kernel form), 1 = epilogue from an LDS tile, 2 = epilogue in place through `out`
static fused_kernel_t fused_kernel(int log2n, int mel_mode, bool hi, bool bands, int streams, int fuse = 0) {
    switch (log2n) {
        case 11: return fused_kernel_m<11>(mel_mode, hi, bands, streams, fuse);
        case 10: return fused_kernel_m<10>(mel_mode, hi, bands, streams, fuse);
        case 9: return fused_kernel_m<9>(mel_mode, hi, bands, streams, fuse);
        default: return fused_kernel_m<8>(mel_mode, hi, bands, streams, fuse);
    }
}
static fused_kernel_t mfma_kernel(int log2n) {
    switch (log2n) {
        case 11: return k_wav_to_mel_mfma<11>;
        case 10: return k_wav_to_mel_mfma<10>;
        default: return k_wav_to_mel_mfma<9>;
    }
}
// LDS of the MFMA variant: landing + exchange buffers of its 8 waves, two fp16 magnitude tiles
static size_t mfma_lds_bytes(const iris_plan* p) {
    const size_t xbuf = (wave_buf_bytes(p->log2n) + 15) & ~(size_t)15;
    const size_t kb_pad = ((size_t)p->mfma_kb + 63) & ~(size_t)63;
    return (size_t)kMfmaWaves * p->n_fft * 4 + std::max((size_t)kMfmaWaves * xbuf, (size_t)const_nv4(p->log2n) * 64 * 16) +
           2 * (size_t)kMfmaGroup * (kb_pad + 8) * 2;
}
static const void* stft_kernel(int log2n) {
    switch (log2n) {
        case 11: return (const void*)k_stft<11>;
        case 10: return (const void*)k_stft<10>;
        case 9: return (const void*)k_stft<9>;
        default: return (const void*)k_stft<8>;
    }
}

// Dynamic LDS above the 64 KiB default must be opted into once per kernel.
static hipError_t allow_big_lds(const iris_plan* p) {
    constexpr int kMaxLds = 160 * 1024;
    hipError_t e;
    for (int v = 0; v < (IRIS_DIAG ? 4 : 2); ++v) {
        const int streams = (v & 2) ? 2 : 1;
        if (streams == 2 && p->log2n != 9 && p->log2n != 10) continue;
        for (int fuse = 0; fuse < (streams == 1 ? 3 : 1); ++fuse) {
            e = hipFuncSetAttribute((const void*)fused_kernel(p->log2n, p->mel_mode, p->need_hi != 0, (v & 1) != 0, streams, fuse),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
            if (e != hipSuccess) return e;
        }
    }
    if (p->mfma_ok) {
        e = hipFuncSetAttribute((const void*)mfma_kernel(p->log2n), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
        if (e != hipSuccess) return e;
    }
    return hipFuncSetAttribute(stft_kernel(p->log2n), hipFuncAttributeMaxDynamicSharedMemorySize, kMaxLds);
}

extern "C" int iris_abi_version(void) { return IRIS_ABI_VERSION; }
extern "C" const char* iris_last_error(void) { return g_err; }

extern "C" int iris_plan_create(iris_plan** out, int device, int n_fft, int hop, int n_mel, int n_bins,
                                float sample_rate, float lower_hz, float upper_hz, int channels, int max_batch,
                                int max_len, const float* mel_host) {
    if (!out) return fail(IRIS_E_INVALID, "iris_plan_create: out is NULL");
    *out = nullptr;
    // n_fft == 0: mel-only plan (iris_magmel on any n_bins >= 2; no FFT entry points)
    const bool mel_only = (n_fft == 0);
    int log2n = mel_only ? 8 : ilog2_exact(n_fft);
    if (!mel_only && (log2n < 8 || log2n > 11))
        return fail(IRIS_E_UNSUPPORTED, "n_fft=%d: must be a power of two in [256, 2048] (or 0 for a mel-only plan)",
                    n_fft);
    if (n_mel <= 0) return fail(IRIS_E_INVALID, "n_mel=%d must be positive", n_mel);
    if (channels <= 0 || max_batch <= 0) return fail(IRIS_E_INVALID, "channels and max_batch must be positive");
    if (mel_only) {
        if (n_bins < 2) return fail(IRIS_E_INVALID, "n_bins=%d must be >= 2", n_bins);
        hop = 1;
        max_len = std::max(max_len, 1);
    } else {
        if (hop <= 0) return fail(IRIS_E_INVALID, "hop=%d must be positive", hop);
        if (n_bins != n_fft / 2 + 1)
            return fail(IRIS_E_INVALID, "n_bins=%d must equal n_fft/2+1=%d", n_bins, n_fft / 2 + 1);
        if (max_len <= n_fft / 2)
            return fail(IRIS_E_INVALID, "max_len=%d must exceed n_fft/2 (reflect padding)", max_len);
    }

    iris_plan* p = new (std::nothrow) iris_plan();
    if (!p) return fail(IRIS_E_NOMEM, "out of host memory");
    p->device = device;
    p->n_fft = n_fft;
    p->mel_only = mel_only;
    p->log2n = log2n;
    p->hop = hop;
    p->n_mel = n_mel;
    p->n_bins = n_bins;
    p->channels = channels;
    p->max_batch = max_batch;
    p->max_len = max_len;
    p->sample_rate = sample_rate;
    p->lower_hz = lower_hz;
    p->upper_hz = upper_hz;
    p->d_consts = nullptr;
    p->d_band_lo = p->d_band_len = p->d_fband_lo = nullptr;
    p->d_bin_band = nullptr;
    p->d_bin_w = nullptr;
    p->d_wband = p->d_mel = p->d_ws = nullptr;
    p->d_wfrag = nullptr;
    p->d_tile_ks = nullptr;
    p->mfma_ok = p->mfma_kb = p->mel_precision = 0;
    p->d_dbg = nullptr;
    p->ablate = 0;
    p->streams = 1;
#if IRIS_DIAG
    if (const char* e = getenv("IRIS_STREAMS")) p->streams = atoi(e) == 2 ? 2 : 1;
#endif
    p->magmel_generic = getenv("IRIS_MAGMEL_GENERIC") != nullptr;  // test hook: read once, never per launch
    p->d_slots = nullptr;
    p->d_status = nullptr;
    p->h_status = nullptr;
    p->timeout_ticks = kEpilogueTimeoutTicks;
    p->epoch = 0;
    p->epilogue = IRIS_EPILOGUE_FUSED;
    p->last_form = -1;
    // a CU mask takes compute units away from this process without changing the device's reported CU count: the fused
    // epilogue's co-residency (grid <= CUs) would not hold, so such an environment starts on the two-kernel form
    if (getenv("ROC_GLOBAL_CU_MASK") || getenv("HSA_CU_MASK")) p->epilogue = IRIS_EPILOGUE_TWO_KERNELS;
    if (const char* e = getenv("IRIS_EPILOGUE")) {
        const int v = atoi(e);
        p->epilogue = (v == IRIS_EPILOGUE_TWO_KERNELS || v == IRIS_EPILOGUE_IN_PLACE) ? v : IRIS_EPILOGUE_FUSED;
    }
    p->timing = 0;
    p->launch_no = 0;
    p->ev_used = 0;
    p->ev2_used = 0;

    p->mel.resize((size_t)n_bins * n_mel);
    if (mel_host) {
        memcpy(p->mel.data(), mel_host, p->mel.size() * sizeof(float));
    } else {
        int rc = iris_mel_weight_matrix(n_mel, n_bins, sample_rate, lower_hz, upper_hz, p->mel.data());
        if (rc != IRIS_OK) {
            delete p;
            return rc;
        }
    }
    // band structure: per mel column the contiguous bin range holding its non-zeros
    std::vector<int> lo(n_mel, 0), len(n_mel, 0);
    p->max_band_len = 0;
    p->k_need = 0;
    for (int m = 0; m < n_mel; ++m) {
        int first = -1, last = -1;
        for (int f = 0; f < n_bins; ++f)
            if (p->mel[(size_t)f * n_mel + m] != 0.f) {
                if (first < 0) first = f;
                last = f;
            }
        if (first >= 0) {
            lo[m] = first;
            len[m] = last - first + 1;
        }
        p->max_band_len = std::max(p->max_band_len, len[m]);
        p->k_need = std::max(p->k_need, lo[m] + len[m]);
    }
    // streaming magmel tables: valid when every bin's non-zeros sit in <= 2 adjacent bands and
    // the first band index never decreases with the bin (true for triangular filterbanks)
    std::vector<int> bin_band(n_bins, -1);
    std::vector<float> bin_w((size_t)n_bins * 2, 0.f);
    p->tri_ok = 1;
    p->tri_f_lo = n_bins;
    p->tri_f_hi = 0;
    {
        int prev = -1;
        for (int f = 0; f < n_bins && p->tri_ok; ++f) {
            int first = -1, last = -1;
            for (int m = 0; m < n_mel; ++m)
                if (p->mel[(size_t)f * n_mel + m] != 0.f) {
                    if (first < 0) first = m;
                    last = m;
                }
            if (first < 0) continue;
            if (last - first > 1 || first < prev) {
                p->tri_ok = 0;
                break;
            }
            prev = first;
            bin_band[f] = first;
            bin_w[2 * (size_t)f] = p->mel[(size_t)f * n_mel + first];
            bin_w[2 * (size_t)f + 1] = last > first ? p->mel[(size_t)f * n_mel + last] : 0.f;
            p->tri_f_lo = std::min(p->tri_f_lo, f);
            p->tri_f_hi = std::max(p->tri_f_hi, f + 1);
        }
    }
    // fused kernel tables: which half of the spectrum it must produce, and per band a
    // window of `rows` bins [flo, flo + rows) inside the bins the kernel writes
    const int NC = mel_only ? 2 * (n_bins - 1) / 2 : n_fft / 2;
    p->need_hi = p->k_need > NC / 2 ? 1 : 0;
    // bins the kernel writes to its magnitude buffer: [0, limit)
    const int limit = p->need_hi ? ((n_bins + 3) & ~3) : NC / 2;
    // (n_fft 2048 keeps 16 points per lane: no registers to spare for the weights -> table modes)
    // (the full-spectrum untangle needs the registers too: need_hi -> table modes)
    int max_span = 0;  // widest band counted from the 4-bin boundary below its first bin
    for (int m = 0; m < n_mel; ++m) max_span = std::max(max_span, (lo[m] & 3) + len[m]);
#ifndef IRIS_EXP_MELMODE1
#define IRIS_EXP_MELMODE1 0  // experiment (k_fused.h): never the register band weights
#endif
    if (!IRIS_EXP_MELMODE1 && log2n <= 10 && !p->need_hi && n_mel <= 64 && p->max_band_len + 3 <= kMelRegs && limit >= kMelRegs) {
        p->mel_mode = 0;  // 16-byte aligned register window of kMelRegs bins per band
        p->rows = kMelRegs;
    } else if (!IRIS_EXP_MELMODE1 && log2n <= 10 && !p->need_hi && n_mel > 64 && n_mel <= 128 && max_span <= 8 && limit >= 8) {
        p->mel_mode = 3;  // two bands per lane, 16-byte aligned register windows of 8 bins
        p->rows = 8;
    } else {
        // LDS table, read as float4: windows start on a multiple of 4 bins and span a multiple of 4
        const int rows4 = (std::max(p->max_band_len, 1) + 3 + 3) & ~3;
        if (rows4 <= limit && ((size_t)rows4 * n_mel + n_mel) * 4 <= 40 * 1024) {
            p->mel_mode = 1;
            p->rows = rows4;
        } else {  // global table, any window
            p->mel_mode = 2;
            p->rows = std::min(std::max(p->max_band_len, 1), limit);
        }
    }
    std::vector<int> flo(n_mel, 0);
    std::vector<float> wband((size_t)p->rows * n_mel, 0.f);
    for (int m = 0; m < n_mel; ++m) {
        int first = p->mel_mode != 2 ? (lo[m] & ~3) : lo[m];
        flo[m] = std::max(0, std::min(first, limit - p->rows));
        for (int i = 0; i < p->rows; ++i) {
            const int f = flo[m] + i;
            wband[(size_t)i * n_mel + m] = f < n_bins ? 0.5f * p->mel[(size_t)f * n_mel + m] : 0.f;
        }
    }

    // fp16-MFMA mel variant (k_fused_mfma.h): per tile of 16 bands the k-steps (32 bins) that hold one of its
    // non-zeros, and the A fragments 0.5 W^T in the lane layout of v_mfma_f32_16x16x32_f16:
    // lane l, element e = 0.5 W[32 ks + 8 (l >> 4) + e][16 tile + (l & 15)]
    std::vector<_Float16> wfrag;
    std::vector<int> tile_ks;
    if (!mel_only && log2n >= 9 && log2n <= 11 && !p->need_hi && n_mel <= 16 * kMfmaWaves) {
        const int n_tiles = (n_mel + 15) / 16;
        p->mfma_ok = 1;
        tile_ks.assign(2 * kMfmaWaves, 0);
        wfrag.assign((size_t)kMfmaWaves * kMfmaKsMax * 64 * 8, (_Float16)0.f);
        for (int t = 0; t < n_tiles && p->mfma_ok; ++t) {
            int b_lo = n_bins, b_hi = 0;
            for (int m = 16 * t; m < std::min(16 * t + 16, n_mel); ++m)
                if (len[m] > 0) {
                    b_lo = std::min(b_lo, lo[m]);
                    b_hi = std::max(b_hi, lo[m] + len[m]);
                }
            if (b_hi <= b_lo) continue;  // an all-zero tile: no k-steps
            const int ks_lo = b_lo / 32, ks_hi = (b_hi + 31) / 32;
            if (ks_hi - ks_lo > kMfmaKsMax || ks_hi * 32 > NC / 2) {
                p->mfma_ok = 0;
                break;
            }
            tile_ks[2 * t] = ks_lo;
            tile_ks[2 * t + 1] = ks_hi - ks_lo;
            p->mfma_kb = std::max(p->mfma_kb, ks_hi * 32);
            for (int j = 0; j < ks_hi - ks_lo; ++j)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int f = 32 * (ks_lo + j) + 8 * (l >> 4) + e, m = 16 * t + (l & 15);
                        const float w = (f < n_bins && m < n_mel) ? 0.5f * p->mel[(size_t)f * n_mel + m] : 0.f;
                        wfrag[(((size_t)t * kMfmaKsMax + j) * 64 + l) * 8 + e] = (_Float16)w;
                    }
        }
        if (p->mfma_kb == 0) p->mfma_ok = 0;
    }

    DeviceGuard guard(device);
    if (!guard.ok) {
        delete p;
        return fail(IRIS_E_INVALID, "cannot select HIP device %d", device);
    }
    std::vector<float2> tw, post, win;
    build_tables(log2n, tw, post, win);
    // pack the per-lane constant block (ConstLayout)
    const int ntw = fft_ntw(log2n), P = fft_p(log2n);
    const int off_post = 2 * ntw, off_win = off_post + P, off_wreg = off_win + 2 * P, off_lo = off_wreg + kMelRegs;
    const int nv4 = (off_lo + 1 + 3) / 4;
    std::vector<float> consts((size_t)nv4 * 64 * 4, 0.f);
    auto put = [&](int lane, int idx, float v) { consts[((size_t)(idx / 4) * 64 + lane) * 4 + (idx % 4)] = v; };
    for (int lane = 0; lane < 64; ++lane) {
        for (int i = 0; i < ntw; ++i) {
            put(lane, 2 * i, tw[(size_t)i * 64 + lane].x);
            put(lane, 2 * i + 1, tw[(size_t)i * 64 + lane].y);
        }
        for (int i = 0; i < P / 2; ++i) {
            put(lane, off_post + 2 * i, post[(size_t)i * 64 + lane].x);
            put(lane, off_post + 2 * i + 1, post[(size_t)i * 64 + lane].y);
        }
        for (int i = 0; i < P; ++i) {
            put(lane, off_win + 2 * i, win[(size_t)i * 64 + lane].x);
            put(lane, off_win + 2 * i + 1, win[(size_t)i * 64 + lane].y);
        }
        if (p->mel_mode == 0 && lane < n_mel) {
            for (int i = 0; i < p->rows; ++i) put(lane, off_wreg + i, wband[(size_t)i * n_mel + lane]);
            float bits;
            memcpy(&bits, &flo[lane], sizeof(float));
            put(lane, off_lo, bits);
        }
        if (p->mel_mode == 3) {  // bands lane and lane + 64: 8 weights each, window starts packed 16 | 16
            const int mb = lane + 64 < n_mel ? lane + 64 : -1;
            for (int i = 0; i < 8; ++i) {
                put(lane, off_wreg + i, wband[(size_t)i * n_mel + lane]);
                put(lane, off_wreg + 8 + i, mb >= 0 ? wband[(size_t)i * n_mel + mb] : 0.f);
            }
            const int packed = flo[lane] | ((mb >= 0 ? flo[mb] : 0) << 16);
            float bits;
            memcpy(&bits, &packed, sizeof(float));
            put(lane, off_lo, bits);
        }
    }
    int rc;
    if ((rc = upload(&p->d_consts, consts)) ||
        (rc = upload(&p->d_band_lo, lo)) || (rc = upload(&p->d_band_len, len)) ||
        (rc = upload(&p->d_fband_lo, flo)) || (rc = upload(&p->d_wband, wband)) ||
        (rc = upload(&p->d_bin_band, bin_band)) || (rc = upload(&p->d_bin_w, bin_w)) ||
        (rc = upload(&p->d_mel, p->mel)) ||
        (p->mfma_ok && ((rc = upload((_Float16**)&p->d_wfrag, wfrag)) || (rc = upload(&p->d_tile_ks, tile_ks))))) {
        iris_plan_destroy(p);
        return rc;
    }

    if (!mel_only) {
        hipError_t e = allow_big_lds(p);
        if (e != hipSuccess) {
            iris_plan_destroy(p);
            return fail((int)e, "hipFuncSetAttribute(max dynamic LDS) failed: %s", hipGetErrorString(e));
        }
    }
    {
        int cu = 0;
        hipError_t e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, device);
        if (e != hipSuccess || cu <= 0) cu = 256;
        p->num_cu = cu;
    }
    p->chunk_target = 0;
    if (const char* e = getenv("IRIS_CHUNK_FRAMES")) p->chunk_target = std::max(0, atoi(e));
    if (!mel_only && std::max(fused_lds_bytes(p, 1, false), fused_lds_bytes(p, 1, true)) > 160 * 1024) {
        iris_plan_destroy(p);
        return fail(IRIS_E_UNSUPPORTED, "n_mel=%d: the band table does not fit the LDS", n_mel);
    }

    // workspace of the fused path: [B, tiles, 2] min/max partials (worst case one
    // frame per tile) + [B, chunks] sums of squares for IRIS_F_NORMALIZE
    const int t_max = 1 + max_len / hop;
    const size_t wav_row = (size_t)channels * max_len;
    p->ws_floats = 2 * 16 * (size_t)max_batch * t_max + (size_t)max_batch * ((wav_row + kChunk - 1) / kChunk) + 64;
#if IRIS_DIAG
    (void)hipMalloc((void**)&p->d_dbg, kDbgWords * sizeof(unsigned long long));
    if (p->d_dbg) (void)hipMemset(p->d_dbg, 0, kDbgWords * sizeof(unsigned long long));
    if (const char* ab = getenv("IRIS_ABLATE")) p->ablate = atoi(ab);
#endif
    hipError_t e;
    e = hipMalloc((void**)&p->d_ws, p->ws_floats * sizeof(float));
    if (e != hipSuccess) {
        iris_plan_destroy(p);
        return fail((int)e, "hipMalloc(workspace %zu floats) failed: %s", p->ws_floats, hipGetErrorString(e));
    }
    // fused epilogue: per-chunk {epoch, min} / {epoch, max} granules (worst case one frame per chunk) + a status word;
    // zeroed once - a granule is valid only when it carries the epoch of the launch that reads it
    p->n_slots = (size_t)max_batch * t_max;
    e = hipMalloc((void**)&p->d_slots, p->n_slots * 16 + 64);
    if (e == hipSuccess) e = hipMemset(p->d_slots, 0, p->n_slots * 16 + 64);
    if (e != hipSuccess) {
        iris_plan_destroy(p);
        return fail((int)e, "hipMalloc(epilogue slots) failed: %s", hipGetErrorString(e));
    }
    // the status word: pinned coherent host memory mapped into the device, so that a failed wait is visible to the
    // host on its next call without any synchronisation
    e = hipHostMalloc((void**)&p->h_status, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) {
        *p->h_status = 0;
        e = hipHostGetDevicePointer((void**)&p->d_status, p->h_status, 0);
    }
    if (e != hipSuccess) {
        iris_plan_destroy(p);
        return fail((int)e, "hipHostMalloc(status word) failed: %s", hipGetErrorString(e));
    }
    *out = p;
    return IRIS_OK;
}

extern "C" int iris_plan_destroy(iris_plan* p) {
    if (!p) return IRIS_OK;
    DeviceGuard guard(p->device);
#if IRIS_DIAG
    if (p->d_dbg && (p->ablate & 4096)) {
        std::vector<unsigned long long> h(kDbgWords, 0);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), p->d_dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sum[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        int waves = 0;
        for (int w = 0; w < 4096 * 16; ++w) {
            const unsigned long long* r = &h[kDbgPhase0 + (size_t)w * 16];
            if (!r[7]) continue;
            ++waves;
            for (int i = 0; i < 16; ++i) sum[i] += (double)r[i];
        }
        if (waves && sum[7] > 0)
            fprintf(stderr, "[iris dbg] %d waves, %.2f frames each; cycles per frame: dma-wait %.0f, frame-read %.0f, "
                    "claim+dma-issue %.0f, window+fft %.0f, untangle+mag %.0f, mel %.0f; per wave: chunk setup %.0f, chunk barrier %.0f, tile write-out %.0f, "
                    "block min/max %.0f, exit %.0f; write-out parts: setup %.0f, lds issue %.0f, lds wait %.0f\n",
                    waves, sum[7] / waves, sum[0] / sum[7], sum[1] / sum[7], sum[2] / sum[7], sum[3] / sum[7],
                    sum[4] / sum[7], sum[5] / sum[7], sum[8] / waves, sum[6] / waves, sum[9] / waves,
                    sum[10] / waves, sum[11] / waves, sum[12] / waves, sum[13] / waves, sum[14] / waves);
        if (sum[15] > 0) fprintf(stderr, "[iris dbg] dummy LDS read after the barrier: %.0f cycles\n", sum[15] / waves);
        if (sum[1] > 0 && (p->ablate & 16384)) fprintf(stderr, "[iris dbg] whole chunk, cold pass %.0f cycles, warm pass %.0f cycles\n", sum[0] / waves, sum[1] / waves);

    }
    if (p->d_dbg && (p->ablate & 512)) {
        std::vector<unsigned long long> h(4 + kDbgWg * 4096, 0);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(h.data(), p->d_dbg, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        fprintf(stderr, "[iris dbg] workgroup 0: %llu shader cycles, %llu x 10 ns -> %.3f GHz\n", h[0], h[1],
                h[1] ? (double)h[0] / ((double)h[1] * 10.0) : 0.0);
        unsigned long long e0 = ~0ull, e1 = 0, l0 = ~0ull, l1 = 0, x0 = ~0ull, x1 = 0;
        double pro = 0, loop = 0, tile_done = 0, xchg = 0, wout = 0;
        int n = 0, n_epi = 0;
        for (int i = 0; i < 4096; ++i) {
            const unsigned long long* r = &h[4 + kDbgWg * i];
            if (!r[0]) continue;
            ++n;
            if (r[3]) { tile_done += (double)(r[3] - r[1]); xchg += (double)(r[4] - r[3]); wout += (double)(r[2] - r[4]); ++n_epi; }
            e0 = std::min(e0, r[0]); e1 = std::max(e1, r[0]);
            l0 = std::min(l0, r[1]); l1 = std::max(l1, r[1]);
            x0 = std::min(x0, r[2]); x1 = std::max(x1, r[2]);
            pro += (double)(r[1] - r[0]); loop += (double)(r[2] - r[1]);
        }
        if (const char* path = getenv("IRIS_DBG_DUMP")) {
            if (FILE* fp = fopen(path, "w")) {
                for (int i = 0; i < 4096; ++i) {
                    const unsigned long long* r = &h[4 + kDbgWg * i];
                    if (r[0]) fprintf(fp, "%d %llu %llu %llu\n", i, r[0] - e0, r[1] - e0, r[2] - e0);
                }
                fclose(fp);
            }
        }
        if (n)
            fprintf(stderr, "[iris dbg] %d workgroups (last launch): entry spread %.2f us, loop-start spread %.2f us, "
                    "exit spread %.2f us, first entry -> last exit %.2f us, mean prologue %.2f us, mean loop %.2f us\n",
                    n, (e1 - e0) * 0.01, (l1 - l0) * 0.01, (x1 - x0) * 0.01, (x1 - e0) * 0.01, pro / n * 0.01,
                    loop / n * 0.01);
        if (n_epi)
            fprintf(stderr, "[iris dbg] fused epilogue (last chunk of %d workgroups): loop start -> tile complete %.2f us, "
                    "-> clip range known %.2f us, -> write-out done %.2f us\n", n_epi, tile_done / n_epi * 0.01,
                    xchg / n_epi * 0.01, wout / n_epi * 0.01);
    }
    (void)hipFree(p->d_dbg);
#endif
    for (hipEvent_t ev : p->ev) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : p->ev2) (void)hipEventDestroy(ev);
    (void)hipFree(p->d_consts);
    (void)hipFree(p->d_band_lo);
    (void)hipFree(p->d_band_len);
    (void)hipFree(p->d_fband_lo);
    (void)hipFree(p->d_bin_band);
    (void)hipFree(p->d_bin_w);
    (void)hipFree(p->d_wband);
    (void)hipFree(p->d_mel);
    (void)hipFree(p->d_wfrag);
    (void)hipFree(p->d_tile_ks);
    (void)hipFree(p->d_ws);
    (void)hipFree(p->d_slots);
    if (p->h_status) (void)hipHostFree(p->h_status);
    delete p;
    return IRIS_OK;
}

extern "C" int iris_plan_set_mel_precision(iris_plan* p, int precision) {
    if (!p) return fail(IRIS_E_INVALID, "iris_plan_set_mel_precision: NULL plan");
    if (precision == IRIS_MEL_F32) {
        p->mel_precision = 0;
        return IRIS_OK;
    }
    if (precision != IRIS_MEL_F16_MFMA) return fail(IRIS_E_INVALID, "iris_plan_set_mel_precision: unknown precision %d", precision);
    if (!p->mfma_ok || mfma_lds_bytes(p) > 160 * 1024)
        return fail(IRIS_E_UNSUPPORTED, "fp16 MFMA mel needs n_fft 512/1024/2048, n_mel <= 128, bands inside the lower half "
                                        "of the spectrum and <= 256 bins per 16 bands");
    p->mel_precision = 1;
    return IRIS_OK;
}

extern "C" int iris_plan_set_epilogue(iris_plan* p, int mode) {
    if (!p) return fail(IRIS_E_INVALID, "iris_plan_set_epilogue: NULL plan");
    if (mode != IRIS_EPILOGUE_FUSED && mode != IRIS_EPILOGUE_TWO_KERNELS && mode != IRIS_EPILOGUE_IN_PLACE)
        return fail(IRIS_E_INVALID, "iris_plan_set_epilogue: unknown mode %d", mode);
    p->epilogue = mode;
    return IRIS_OK;
}

// Reads and clears the host-visible status word.  A raised word means a fused-epilogue launch gave up a wait: the
// co-residency its clip-level exchange relies on does not hold in this process' environment, so the plan leaves the
// fused form for good.
static bool take_status(iris_plan* p) {
    if (!p->h_status) return false;
    const unsigned v = __atomic_exchange_n(p->h_status, 0u, __ATOMIC_ACQ_REL);
    if (v) p->epilogue = IRIS_EPILOGUE_TWO_KERNELS;
    return v != 0;
}

extern "C" int iris_plan_status(iris_plan* p, int* status) {
    if (!p || !status) return fail(IRIS_E_INVALID, "iris_plan_status: NULL argument");
    DeviceGuard guard(p->device);
    HIP_TRY(hipDeviceSynchronize());  // every launch of the plan so far has finished (and its system-scope store landed)
    *status = take_status(p) ? 1 : 0;
    return IRIS_OK;
}

extern "C" int iris_plan_last_epilogue(const iris_plan* p, int* form) {
    if (!p || !form) return fail(IRIS_E_INVALID, "iris_plan_last_epilogue: NULL argument");
    *form = p->last_form;
    return IRIS_OK;
}

extern "C" int iris_plan_set_epilogue_timeout(iris_plan* p, unsigned long long microseconds) {
    if (!p) return fail(IRIS_E_INVALID, "iris_plan_set_epilogue_timeout: NULL plan");
    p->timeout_ticks = microseconds * 100ull;  // s_memrealtime: 100 MHz
    return IRIS_OK;
}

extern "C" int iris_plan_get_mel(const iris_plan* p, float* out) {
    if (!p || !out) return fail(IRIS_E_INVALID, "iris_plan_get_mel: NULL argument");
    memcpy(out, p->mel.data(), p->mel.size() * sizeof(float));
    return IRIS_OK;
}

extern "C" int iris_plan_num_frames(const iris_plan* p, int len) {
    if (!p || len < 0) return fail(IRIS_E_INVALID, "iris_plan_num_frames: bad argument");
    return 1 + len / p->hop;
}

static int check_wav_args(const iris_plan* p, const void* a, const void* b, int batch, int len, const char* who) {
    if (!p || !a || !b) return fail(IRIS_E_INVALID, "%s: NULL argument", who);
    if (p->mel_only) return fail(IRIS_E_UNSUPPORTED, "%s: plan was created mel-only (n_fft = 0)", who);
    if (batch <= 0 || len <= 0) return fail(IRIS_E_INVALID, "%s: batch=%d len=%d must be positive", who, batch, len);
    if (batch > p->max_batch || len > p->max_len)
        return fail(IRIS_E_CAPACITY, "%s: batch=%d len=%d exceed plan capacity (%d, %d)", who, batch, len,
                    p->max_batch, p->max_len);
    if (len <= p->n_fft / 2)
        return fail(IRIS_E_INVALID, "%s: len=%d must exceed n_fft/2=%d (reflect padding)", who, len, p->n_fft / 2);
    return IRIS_OK;
}

static int check_bands(const int32_t* bands, int n, const char* who) {
    if (n < 0 || (n > 0 && !bands)) return fail(IRIS_E_INVALID, "%s: bands pointer/count mismatch", who);
    return IRIS_OK;
}
