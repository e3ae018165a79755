// host_ops.h -- host side: the C-ABI entry points of the operators (argument checks, geometry, launches).
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// host: ops
// ---------------------------------------------------------------------------
static size_t n_chunks_of(size_t row_len) { return (row_len + kChunk - 1) / kChunk; }

extern "C" size_t iris_normalize_workspace(int n_rows, size_t row_len) {
    return n_rows > 0 ? (size_t)n_rows * n_chunks_of(row_len) : 0;
}

extern "C" int iris_normalize(const float* wav, float* out, int n_rows, size_t row_len, float* workspace,
                              size_t workspace_floats, void* stream) {
    if (!wav || !out || !workspace) return fail(IRIS_E_INVALID, "iris_normalize: NULL argument");
    if (n_rows <= 0 || row_len == 0) return fail(IRIS_E_INVALID, "iris_normalize: empty tensor");
    if (n_rows > 65535) return fail(IRIS_E_UNSUPPORTED, "iris_normalize: n_rows %d > 65535", n_rows);
    const size_t n_part = n_chunks_of(row_len);
    if (workspace_floats < iris_normalize_workspace(n_rows, row_len))
        return fail(IRIS_E_CAPACITY, "iris_normalize: workspace %zu floats < %zu", workspace_floats,
                    iris_normalize_workspace(n_rows, row_len));
    hipStream_t s = (hipStream_t)stream;
    k_sumsq_partial<<<dim3((unsigned)n_part, n_rows), 256, 0, s>>>(wav, workspace, row_len, (int)n_part);
    k_normalize_apply<<<dim3((unsigned)n_part, n_rows), 256, 0, s>>>(wav, out, workspace, (int)n_part, row_len);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

static void fused_geometry(const iris_plan* p, int batch, int T, int per_cu, int* chunk_frames, int* chunks_per_clip);

template <int LOG2N>
static hipError_t launch_stft(const StftArgs& a, int grid, size_t lds, hipStream_t s) {
    k_stft<LOG2N><<<grid, 64 * stft_waves(LOG2N), lds, s>>>(a);
    return hipGetLastError();
}

extern "C" int iris_stft(iris_plan* p, const float* wav, float* spec, int batch, int len, int flags, void* stream) {
    int rc = check_wav_args(p, wav, spec, batch, len, "iris_stft");
    if (rc) return rc;
    if (flags & ~IRIS_F_NORMALIZE) return fail(IRIS_E_INVALID, "iris_stft: flags 0x%x (only IRIS_F_NORMALIZE applies)", flags);
    DeviceGuard guard(p->device);
    StftArgs a;
    a.wav = wav;
    a.spec = spec;
    a.consts = p->d_consts;
    a.B = batch;
    a.C = p->channels;
    a.L = len;
    a.T = 1 + len / p->hop;
    a.hop = p->hop;
    const int NC = p->n_fft / 2, F = NC + 1, waves = stft_waves(p->log2n), C2 = 2 * p->channels;
    // tile = as many frames as the LDS left by the waves' exchange buffers holds (one workgroup per
    // CU owns the whole 160 KiB); at least the constant block must fit (it is staged through the tile)
    const size_t xbufs = (size_t)waves * ((wave_buf_bytes(p->log2n) + 15) & ~(size_t)15);
    const int wgs = stft_wgs(p->log2n);  // workgroups per CU
    const size_t room = (160 * 1024 - 1024) / wgs - xbufs;
    auto tile_bytes = [&](int t) { return (size_t)F * ((size_t)t * C2 + 1) * 4; };
    int tf = 1;
    while (tf < 256 && tile_bytes(tf + 1) <= room) ++tf;
    if (tile_bytes(tf) > room) return fail(IRIS_E_UNSUPPORTED, "iris_stft: channels=%d too large", p->channels);
    // a whole number of rounds of the workgroup's waves
    const int per_round = std::max(1, waves / std::max(1, p->channels));
    if (tf > per_round) tf -= tf % per_round;
    a.tile_frames = tf;
    int chunk_frames = 0;
    fused_geometry(p, batch, a.T, wgs, &chunk_frames, &a.chunks_per_clip);  // K1's balanced chunks, one per workgroup
    a.chunk_base = a.T / a.chunks_per_clip;
    a.chunk_rem = a.T % a.chunks_per_clip;
    a.n_chunks = batch * a.chunks_per_clip;
    const size_t lds = xbufs + std::max(tile_bytes(tf), (size_t)const_nv4(p->log2n) * 64 * 16);
    if (lds > 160 * 1024) return fail(IRIS_E_UNSUPPORTED, "iris_stft: %zu B of LDS", lds);
    const int grid = std::min(a.n_chunks, p->num_cu * wgs);
    hipStream_t s = (hipStream_t)stream;
    a.sumsq = nullptr;
    a.n_sq = 0;
    if (flags & IRIS_F_NORMALIZE) {  // the clip's sum of squares as partial sums in the plan's workspace, folded by the kernel
        const size_t row = (size_t)p->channels * len;
        a.n_sq = (int)((row + kChunk - 1) / kChunk);
        if ((size_t)batch * a.n_sq > p->ws_floats) return fail(IRIS_E_CAPACITY, "iris_stft: workspace too small");
        k_sumsq_partial<<<dim3(a.n_sq, batch), 256, 0, s>>>(wav, p->d_ws, row, a.n_sq);
        a.sumsq = p->d_ws;
    }
    hipError_t e;
    switch (p->log2n) {
        case 11: e = launch_stft<11>(a, grid, lds, s); break;
        case 10: e = launch_stft<10>(a, grid, lds, s); break;
        case 9: e = launch_stft<9>(a, grid, lds, s); break;
        default: e = launch_stft<8>(a, grid, lds, s); break;
    }
    HIP_TRY(e);
    return IRIS_OK;
}

static int grid_for(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 2048 * 4); }

extern "C" int iris_complex_to_magphase(const float* in, float* out, size_t n_outer, int channels, void* stream) {
    if (!in || !out || channels <= 0) return fail(IRIS_E_INVALID, "iris_complex_to_magphase: bad argument");
    if (n_outer == 0) return IRIS_OK;
    k_complex_to_magphase<<<grid_for(n_outer * channels), 256, 0, (hipStream_t)stream>>>(in, out, n_outer, channels);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_magphase_to_complex(const float* in, float* out, size_t n_outer, int channels, void* stream) {
    if (!in || !out || channels <= 0) return fail(IRIS_E_INVALID, "iris_magphase_to_complex: bad argument");
    if (n_outer == 0) return IRIS_OK;
    k_magphase_to_complex<<<grid_for(n_outer * channels), 256, 0, (hipStream_t)stream>>>(in, out, n_outer, channels);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_magmel(iris_plan* p, const float* spec, float* mel, int batch, int n_frames, int is_magphase,
                           const int32_t* t_bands, int n_tb, const int32_t* f_bands, int n_fb, void* stream) {
    if (!p || !spec || !mel) return fail(IRIS_E_INVALID, "iris_magmel: NULL argument");
    if (batch <= 0 || n_frames <= 0) return fail(IRIS_E_INVALID, "iris_magmel: batch=%d n_frames=%d", batch, n_frames);
    if (batch > 65535) return fail(IRIS_E_UNSUPPORTED, "iris_magmel: batch %d > 65535", batch);
    int rc;
    if ((rc = check_bands(t_bands, n_tb, "iris_magmel")) || (rc = check_bands(f_bands, n_fb, "iris_magmel"))) return rc;
    DeviceGuard guard(p->device);
    MagmelArgs a;
    a.spec = spec;
    a.mel = mel;
    a.w = p->d_mel;
    a.band_lo = p->d_band_lo;
    a.band_len = p->d_band_len;
    a.t_bands = n_tb ? t_bands : nullptr;
    a.n_tb = n_tb;
    a.f_bands = n_fb ? f_bands : nullptr;
    a.n_fb = n_fb;
    a.B = batch;
    a.C = p->channels;
    a.F = p->n_bins;
    a.T = n_frames;
    a.M = p->n_mel;
    a.is_magphase = is_magphase;
    const bool aligned = (reinterpret_cast<uintptr_t>(spec) & (8 * p->channels - 1)) == 0;
    if (p->tri_ok && (p->channels == 1 || p->channels == 2) && aligned &&
        (size_t)p->n_mel * 64 * p->channels * sizeof(float) <= 64 * 1024 && !p->magmel_generic) {
        MagmelTriArgs t;
        t.spec = spec;
        t.mel = mel;
        t.bin_band = p->d_bin_band;
        t.bin_w = p->d_bin_w;
        t.t_bands = a.t_bands;
        t.n_tb = n_tb;
        t.f_bands = a.f_bands;
        t.n_fb = n_fb;
        t.B = batch;
        t.F = p->n_bins;
        t.T = n_frames;
        t.M = p->n_mel;
        t.is_magphase = is_magphase;
        t.f_lo = p->tri_f_lo;
        t.f_hi = p->tri_f_hi;
        const dim3 grid((n_frames + 63) / 64, batch);
        // split the bins over 8 waves when the grid alone cannot fill the chip
        const int threads = (size_t)grid.x * grid.y * 4 < (size_t)p->num_cu * 8 ? 512 : 256;
        const size_t lds = (size_t)p->n_mel * 64 * p->channels * sizeof(float);
        if (p->channels == 1) k_magmel_tri<1><<<grid, threads, lds, (hipStream_t)stream>>>(t);
        else k_magmel_tri<2><<<grid, threads, lds, (hipStream_t)stream>>>(t);
    } else {
        const int tc = n_frames * p->channels;
        k_magmel<<<dim3((tc + 63) / 64, batch), 256, 0, (hipStream_t)stream>>>(a);
    }
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" size_t iris_minmax_log_workspace(int n_rows, size_t row_len) {
    return n_rows > 0 ? 2 * (size_t)n_rows * n_chunks_of(row_len) : 0;
}

extern "C" int iris_minmax_log(float* x, int n_rows, size_t row_len, int do_minmax, int do_log, float eps_div,
                               float eps_log, float* workspace, size_t workspace_floats, void* stream) {
    if (!x) return fail(IRIS_E_INVALID, "iris_minmax_log: x is NULL");
    if (n_rows <= 0 || row_len == 0) return fail(IRIS_E_INVALID, "iris_minmax_log: empty tensor");
    if (n_rows > 65535) return fail(IRIS_E_UNSUPPORTED, "iris_minmax_log: n_rows %d > 65535", n_rows);
    hipStream_t s = (hipStream_t)stream;
    const size_t n_part = n_chunks_of(row_len);
    if (do_minmax) {
        if (reinterpret_cast<uintptr_t>(workspace) & 7)
            return fail(IRIS_E_INVALID, "iris_minmax_log: workspace must be 8-byte aligned");
        if (!workspace || workspace_floats < iris_minmax_log_workspace(n_rows, row_len))
            return fail(IRIS_E_CAPACITY, "iris_minmax_log: workspace %zu floats < %zu", workspace_floats,
                        iris_minmax_log_workspace(n_rows, row_len));
        k_minmax_partial<<<dim3((unsigned)n_part, n_rows), 256, 0, s>>>(x, workspace, row_len, (int)n_part);
    }
    k_minmax_log_apply<<<dim3((unsigned)((row_len + kApply - 1) / kApply), n_rows), 256, 0, s>>>(x, workspace, (int)n_part, row_len, do_minmax,
                                                                     do_log, eps_div, eps_log);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

constexpr int kMaxChunkFrames = 16384;
// Chunk geometry of the fused kernel for `per_cu` workgroups per CU: every workgroup one
// chunk when the problem is large enough, chunks never span clips.
static void fused_geometry(const iris_plan* p, int batch, int T, int per_cu, int* chunk_frames, int* chunks_per_clip) {
    const int slots = p->num_cu * per_cu;
    const long total = (long)batch * T;
    int target = p->chunk_target > 0 ? p->chunk_target : (int)((total + slots - 1) / slots);
    target = std::max(target, std::min(8, T));
    int cpc = (T + target - 1) / target;
    // rounding up per clip can overshoot the slots by a few chunks, which would cost a whole
    // second round: prefer slightly larger chunks that fit one round
    if (p->chunk_target == 0 && (long)batch * cpc > slots && batch <= slots) cpc = std::max(1, slots / batch);
    cpc = std::max(cpc, (T + kMaxChunkFrames - 1) / kMaxChunkFrames);  // bounds the time-band bitmap in LDS
    *chunks_per_clip = cpc;
    *chunk_frames = (T + cpc - 1) / cpc;
}

// Geometry + grid of the fused kernel: one workgroup per CU (every wave the registers allow), checked once per
// (kernel, batch, frames) against the occupancy the hardware really grants and cached in the plan - the hot launch
// path does no runtime query.
static int fused_config(iris_plan* p, fused_kernel_t kernel, int batch, int T, int streams, bool bands, int fuse,
                        int* chunk_frames, int* chunks_per_clip, int* grid, size_t* lds) {
    for (const iris_plan::FusedGeom& g : p->geom_cache)
        if (g.kernel == (const void*)kernel && g.batch == batch && g.T == T) {
            *chunk_frames = g.chunk_frames;
            *chunks_per_clip = g.chunks_per_clip;
            *grid = g.grid;
            *lds = g.lds;
            return g.lds == 0 ? IRIS_E_UNSUPPORTED : IRIS_OK;  // lds 0: this shape does not fit (remembered)
        }
    const int per_cu = fuse ? 1 : IRIS_WGS_PER_CU;  // (the epilogue forms need every workgroup resident: one per CU)
    fused_geometry(p, batch, T, per_cu, chunk_frames, chunks_per_clip);
    *lds = fused_lds_bytes(p, streams, bands, *chunk_frames, fuse);
    if (fuse) *lds = fused_tile_off(*lds) + fused_tile_bytes(p, streams, bands, *chunk_frames, fuse);
    *grid = std::min(batch * *chunks_per_clip, p->num_cu * per_cu);
    if (p->geom_cache.size() >= 64) p->geom_cache.clear();
    if (*lds * per_cu > 160 * 1024) {
        p->geom_cache.push_back({(const void*)kernel, batch, T, *chunk_frames, *chunks_per_clip, *grid, 0});
        return fail(IRIS_E_UNSUPPORTED, "fused kernel needs %zu B of LDS", *lds);
    }
    int resident = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&resident, (const void*)kernel,
                                                                64 * fused_waves(p->log2n, streams, bands, p->need_hi != 0, fuse, p->mel_mode), *lds);
    if (e != hipSuccess) return fail((int)e, "hipOccupancyMaxActiveBlocksPerMultiprocessor: %s", hipGetErrorString(e));
    if (resident < per_cu) return fail(IRIS_E_UNSUPPORTED, "fused kernel does not fit %d workgroup(s) per CU (LDS %zu B)", per_cu, *lds);
    p->geom_cache.push_back({(const void*)kernel, batch, T, *chunk_frames, *chunks_per_clip, *grid, *lds});
    return IRIS_OK;
}

// bench hook: launch with the kernel's own start/stop timestamps attached to an event pair by the AMD launch
// extension (no hipEventRecord packets of our own on the stream)
static hipError_t launch_timed(std::vector<hipEvent_t>& ev, int& used, const void* kernel, dim3 grid, dim3 block,
                               void** kargs, size_t lds, hipStream_t s) {
    while ((int)ev.size() < 2 * (used + 1)) {
        hipEvent_t e;
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return rc;
        ev.push_back(e);
    }
    hipError_t rc = hipExtLaunchKernel(kernel, grid, block, kargs, lds, s, ev[2 * used], ev[2 * used + 1], 0);
    if (rc == hipSuccess) ++used;
    return rc;
}

extern "C" int iris_wav_to_logmel(iris_plan* p, const float* wav, float* out, int batch, int len, int flags,
                                  const int32_t* t_bands, int n_tb, const int32_t* f_bands, int n_fb,
                                  void* stream) {
    int rc = check_wav_args(p, wav, out, batch, len, "iris_wav_to_logmel");
    if (rc) return rc;
    if ((rc = check_bands(t_bands, n_tb, "iris_wav_to_logmel")) ||
        (rc = check_bands(f_bands, n_fb, "iris_wav_to_logmel")))
        return rc;
    if (take_status(p))  // raised by an earlier launch (host-visible word: no synchronisation here)
        return fail(IRIS_E_EPILOGUE_TIMEOUT,
                    "iris_wav_to_logmel: an earlier fused-epilogue launch of this plan gave up waiting for its clip's other "
                    "workgroups (they were not co-resident: concurrent kernels, a CU mask or another process on the device) and "
                    "wrote NaN; nothing was enqueued by this call, the plan now uses the two-kernel form");
    DeviceGuard guard(p->device);
    hipStream_t s = (hipStream_t)stream;
    FusedArgs a;
    a.wav = wav;
    a.out = out;
    a.consts = p->d_consts;
    a.band_lo = p->d_fband_lo;
    a.wband = p->d_wband;
    a.rows = p->rows;
    a.t_bands = n_tb ? t_bands : nullptr;
    a.n_tb = n_tb;
    a.f_bands = n_fb ? f_bands : nullptr;
    a.n_fb = n_fb;
    a.B = batch;
    a.C = p->channels;
    a.L = len;
    a.T = 1 + len / p->hop;
    a.hop = p->hop;
    a.M = p->n_mel;
    const int do_minmax = (flags & IRIS_F_MINMAX) ? 1 : 0, do_log = (flags & IRIS_F_LOG) ? 1 : 0;
    a.ablate = p->ablate;  // 0 unless this is an IRIS_DIAG build
    a.dbg = p->d_dbg;
    const bool bands = (n_tb > 0) || (n_fb > 0);
    const bool mfma = p->mel_precision == 1 && !bands;  // calls with bands always take the fp32 kernel
    const int streams = mfma ? 1 : plan_streams(p);
    // min-max / log inside the kernel (one launch) unless: nothing to apply, the MFMA variant, two frame streams
    // (diag), the plan says two kernels, the stream is being captured (the epoch is a host counter: it would be
    // frozen in the graph), or the chunk's mel tile does not fit the LDS
    // fuse: 0 = two kernels, 1 = epilogue from the chunk's LDS tile, 2 = epilogue in place through `out` (only when the plan
    // asks for it, IRIS_EPILOGUE_IN_PLACE: an A/B form - see below)
    int fuse = ((do_minmax || do_log) && !mfma && streams == 1 && p->epilogue != IRIS_EPILOGUE_TWO_KERNELS)
                   ? (p->epilogue == IRIS_EPILOGUE_IN_PLACE ? 2 : 1) : 0;
    if (fuse) {
        hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) fuse = 0;
    }
    fused_kernel_t kernel = nullptr;
    int grid = 0;
    size_t lds = 0;
    a.wfrag = p->d_wfrag;
    a.tile_ks = p->d_tile_ks;
    a.kb = p->mfma_kb;
    if (mfma) {
        kernel = mfma_kernel(p->log2n);
        fused_geometry(p, batch, a.T, 1, &a.chunk_frames, &a.chunks_per_clip);
        lds = mfma_lds_bytes(p);
        grid = std::min(batch * a.chunks_per_clip, p->num_cu);
    } else {
        if (fuse) {
            kernel = fused_kernel(p->log2n, p->mel_mode, p->need_hi != 0, bands, streams, fuse);
            rc = fused_config(p, kernel, batch, a.T, streams, bands, fuse, &a.chunk_frames, &a.chunks_per_clip, &grid, &lds);
            // no fused epilogue when the LDS tile does not fit, or when a clip has more chunks than the grid has workgroups (a
            // workgroup would then wait for a chunk it has yet to process itself): the two-kernel form.  (Round 5 measured the
            // in-place form as the fallback for tiles beyond the LDS: same bytes through the same fabric as the second kernel,
            // moved by every CU at the same moment - 3-5 % slower at B = 128 and 512, equal at 256; it stays selectable.)
            if (rc == IRIS_E_UNSUPPORTED || (size_t)batch * a.chunks_per_clip > p->n_slots || a.chunks_per_clip > grid) fuse = 0;
            else if (rc) return rc;
        }
        if (!fuse) {
            kernel = fused_kernel(p->log2n, p->mel_mode, p->need_hi != 0, bands, streams, 0);
            if ((rc = fused_config(p, kernel, batch, a.T, streams, bands, 0, &a.chunk_frames, &a.chunks_per_clip, &grid, &lds)))
                return rc;
        }
    }
    a.slots = p->d_slots;
    a.status = p->d_status;
    a.timeout_ticks = p->timeout_ticks;
    a.epoch = 0;
    a.tile_off = a.pitch = 0;
    a.do_minmax = do_minmax;
    a.do_log = do_log;
    p->last_form = (do_minmax || do_log) ? (fuse == 1 ? IRIS_EPILOGUE_FUSED : (fuse == 2 ? IRIS_EPILOGUE_IN_PLACE : IRIS_EPILOGUE_TWO_KERNELS)) : -1;
    if (fuse) {
        if (++p->epoch == 0) p->epoch = 1;
        a.epoch = p->epoch;
        a.pitch = fuse == 1 ? fused_tile_pitch(p, a.chunk_frames) : 0;
        a.tile_off = (int)fused_tile_off(fused_lds_bytes(p, streams, bands, a.chunk_frames, fuse));
    }
    if ((size_t)p->n_mel * a.T * p->channels * 4 > 0xffffffffull || (size_t)a.T * p->channels * 4 >= (1u << 24))
        return fail(IRIS_E_UNSUPPORTED, "iris_wav_to_logmel: clip too long (%d frames x %d channels)", a.T, p->channels);
    a.n_chunks = batch * a.chunks_per_clip;
    a.chunk_base = a.T / a.chunks_per_clip;
    a.chunk_rem = a.T % a.chunks_per_clip;
    const int waves = mfma ? kMfmaWaves : fused_waves(p->log2n, streams, bands, p->need_hi != 0, fuse, p->mel_mode);
    const int parts_per_chunk = waves;
    const size_t n_partial = 2 * (size_t)a.n_chunks * parts_per_chunk;
    a.partial = p->d_ws;
    a.sumsq = nullptr;
    a.n_sq = 0;
    if (flags & IRIS_F_NORMALIZE) {
        const size_t row = (size_t)p->channels * len;
        a.n_sq = (int)((row + kChunk - 1) / kChunk);
        float* sq = p->d_ws + n_partial;
        if (n_partial + (size_t)batch * a.n_sq > p->ws_floats)
            return fail(IRIS_E_CAPACITY, "iris_wav_to_logmel: workspace too small");
        k_sumsq_partial<<<dim3(a.n_sq, batch), 256, 0, s>>>(wav, sq, row, a.n_sq);
        a.sumsq = sq;
    }
    if (n_partial > p->ws_floats) return fail(IRIS_E_CAPACITY, "iris_wav_to_logmel: workspace too small");

    // bench hook (iris_timing_enable): every n-th call carries event pairs around both kernels; the first
    // kTimingSkip calls after enabling are never sampled (first dispatch on an idle GPU, clock ramp)
    const bool timed = p->timing > 0 && p->launch_no >= kTimingSkip && ((p->launch_no - kTimingSkip) % p->timing) == 0 &&
                       p->ev_used < kMaxTimedLaunches;
    p->launch_no++;
    hipError_t e;
    if (timed) {
        FusedArgs args = a;
        void* kargs[] = {&args};
        e = launch_timed(p->ev, p->ev_used, (const void*)kernel, dim3(grid), dim3(64 * waves), kargs, lds, s);
    } else {
        kernel<<<grid, 64 * waves, lds, s>>>(a);
        e = hipGetLastError();
    }
    HIP_TRY(e);
    if ((do_minmax || do_log) && !fuse) {
        size_t row_len = (size_t)p->n_mel * a.T * p->channels;
        const unsigned n_chunks = (unsigned)((row_len + kApply - 1) / kApply);
        float* x = out;
        const float* partial = p->d_ws;
        int n_part = a.chunks_per_clip * parts_per_chunk, mm = do_minmax, lg = do_log;
        float eps_div = 1e-8f, eps_log = 1e-8f;
        if (timed) {
            void* kargs[] = {&x, &partial, &n_part, &row_len, &mm, &lg, &eps_div, &eps_log};
            e = launch_timed(p->ev2, p->ev2_used, (const void*)k_minmax_log_apply, dim3(n_chunks, batch), dim3(256), kargs, 0, s);
        } else {
            k_minmax_log_apply<<<dim3(n_chunks, batch), 256, 0, s>>>(x, partial, n_part, row_len, mm, lg, eps_div, eps_log);
            e = hipGetLastError();
        }
        HIP_TRY(e);
    }
    return IRIS_OK;
}

extern "C" int iris_mask_apply(void* x, size_t n_outer, size_t axis_len, size_t n_inner, int elem_size,
                               const int32_t* bands, int n_bands, size_t outer_per_group, void* stream) {
    if (!x) return fail(IRIS_E_INVALID, "iris_mask_apply: x is NULL");
    if (elem_size != 4 && elem_size != 8) return fail(IRIS_E_UNSUPPORTED, "iris_mask_apply: elem_size %d", elem_size);
    int rc = check_bands(bands, n_bands, "iris_mask_apply");
    if (rc) return rc;
    if (outer_per_group == 0) return fail(IRIS_E_INVALID, "iris_mask_apply: outer_per_group must be > 0");
    const size_t total = n_outer * axis_len * n_inner;
    if (total == 0 || n_bands == 0) return IRIS_OK;
    hipStream_t s = (hipStream_t)stream;
    if (elem_size == 4)
        k_mask_apply<uint32_t><<<grid_for(total), 256, 0, s>>>((uint32_t*)x, n_outer, axis_len, n_inner, bands,
                                                              n_bands, outer_per_group);
    else
        k_mask_apply<uint64_t><<<grid_for(total), 256, 0, s>>>((uint64_t*)x, n_outer, axis_len, n_inner, bands,
                                                              n_bands, outer_per_group);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_agc_clip(const iris_agc_row* rows_dev, size_t n_rows, float clip_factor, float eps,
                             float clipvalue, void* stream) {
    if (!rows_dev) return fail(IRIS_E_INVALID, "iris_agc_clip: rows is NULL");
    if (n_rows == 0) return IRIS_OK;
    const int grid = (int)std::min<size_t>((n_rows + 3) / 4, 4096);
    k_agc_clip<<<grid, 256, 0, (hipStream_t)stream>>>(rows_dev, n_rows, clip_factor, eps, clipvalue);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_bias_relu(float* x, const float* bias, size_t n_outer, int channels, void* stream) {
    if (!x || !bias) return fail(IRIS_E_INVALID, "iris_bias_relu: NULL argument");
    if (channels <= 0 || (channels & 3)) return fail(IRIS_E_UNSUPPORTED, "iris_bias_relu: channels=%d must be a positive multiple of 4", channels);
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias)) & 15)
        return fail(IRIS_E_INVALID, "iris_bias_relu: x and bias must be 16-byte aligned");
    if (n_outer == 0) return IRIS_OK;
    const size_t n4 = n_outer * (size_t)(channels / 4);
    k_bias_relu<<<grid_for(n4), 256, 0, (hipStream_t)stream>>>(x, bias, n4, channels / 4);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_bias_relu_maxpool(const float* x, const float* bias, float* y, int batch, int height, int width, int channels,
                                      void* stream) {
    if (!x || !bias || !y) return fail(IRIS_E_INVALID, "iris_bias_relu_maxpool: NULL argument");
    if (channels <= 0 || (channels & 3)) return fail(IRIS_E_UNSUPPORTED, "iris_bias_relu_maxpool: channels=%d must be a positive multiple of 4", channels);
    if (batch <= 0 || height <= 0 || width <= 0) return fail(IRIS_E_INVALID, "iris_bias_relu_maxpool: empty tensor");
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(bias) | reinterpret_cast<uintptr_t>(y)) & 15)
        return fail(IRIS_E_INVALID, "iris_bias_relu_maxpool: x, bias and y must be 16-byte aligned");
    const size_t n4 = (size_t)batch * ((height + 1) / 2) * ((width + 1) / 2) * (channels / 4);
    if (n4 >= 2147483648ull) return fail(IRIS_E_UNSUPPORTED, "iris_bias_relu_maxpool: more than 2^31 pooled float4 elements");
    k_bias_relu_pool<<<grid_for(n4), 256, 0, (hipStream_t)stream>>>(x, bias, y, batch, height, width, channels / 4);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_bias_relu_nchw(float* x, const float* bias, size_t batch, int channels, size_t inner, void* stream) {
    if (!x || !bias) return fail(IRIS_E_INVALID, "iris_bias_relu_nchw: NULL argument");
    if (channels <= 0 || inner == 0 || (inner & 3)) return fail(IRIS_E_UNSUPPORTED, "iris_bias_relu_nchw: inner size %zu must be a positive multiple of 4", inner);
    if (reinterpret_cast<uintptr_t>(x) & 15) return fail(IRIS_E_INVALID, "iris_bias_relu_nchw: x must be 16-byte aligned");
    if (batch == 0) return IRIS_OK;
    const size_t n4 = batch * (size_t)channels * (inner / 4);
    k_bias_relu_nchw<<<grid_for(n4), 256, 0, (hipStream_t)stream>>>(x, bias, n4, inner / 4, channels);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_bias_relu_maxpool_nchw(const float* x, const float* bias, float* y, int batch, int height, int width,
                                           int channels, void* stream) {
    if (!x || !bias || !y) return fail(IRIS_E_INVALID, "iris_bias_relu_maxpool_nchw: NULL argument");
    if (batch <= 0 || height <= 0 || width <= 0 || channels <= 0) return fail(IRIS_E_INVALID, "iris_bias_relu_maxpool_nchw: empty tensor");
    int tile_w = 64;
    while (tile_w > 1 && (size_t)tile_w * (channels + 1) * sizeof(float) > 48 * 1024) tile_w >>= 1;
    const size_t lds = (size_t)tile_w * (channels + 1) * sizeof(float);
    if (lds > 48 * 1024 || batch > 65535 || (height + 1) / 2 > 65535)
        return fail(IRIS_E_UNSUPPORTED, "iris_bias_relu_maxpool_nchw: %d channels / batch %d / height %d out of range", channels, batch, height);
    const int Wo = (width + 1) / 2;
    k_bias_relu_pool_nchw<<<dim3((Wo + tile_w - 1) / tile_w, (height + 1) / 2, batch), 256, lds, (hipStream_t)stream>>>(
        x, bias, y, height, width, channels, tile_w);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

static int bn_check(const void* a, const void* b, size_t rows, int channels, const char* who) {
    if (!a || !b) return fail(IRIS_E_INVALID, "%s: NULL argument", who);
    if (rows == 0 || channels <= 0 || (channels & 3) || channels > 4096)
        return fail(IRIS_E_UNSUPPORTED, "%s: rows %zu, channels %d (a positive multiple of 4, <= 4096)", who, rows, channels);
    return IRIS_OK;
}
static bool bn_overlap(const float* z, size_t n_z, const float* y, size_t n_y) { return y < z + n_z && z < y + n_y; }
static int grid_bn(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 2048); }  // fat blocks: the per-block coefficient setup is amortised
static unsigned bn_reduce_grid(size_t rows, int C4, int rows_per_pass = kBnRows) {
    const int cols = std::min(C4, 256), tys = 256 / cols;
    return (unsigned)((rows + (size_t)rows_per_pass * tys - 1) / ((size_t)rows_per_pass * tys));
}
static size_t bn_reduce_lds(int C4) {
    const int cols = std::min(C4, 256), tys = 256 / cols;
    return (size_t)tys * 2 * cols * 4 * sizeof(float);
}

extern "C" size_t iris_bn_sums_len(int channels) { return channels > 0 ? (size_t)bn_slots(channels) * 2 * channels : 0; }

extern "C" int iris_bn_stats(const float* z, size_t rows, int channels, double* sums_zeroed, void* stream) {
    int rc = bn_check(z, sums_zeroed, rows, channels, "iris_bn_stats");
    if (rc) return rc;
    const int C4 = channels / 4;
    k_bn_reduce<false><<<bn_reduce_grid(rows, C4), 256, bn_reduce_lds(C4), (hipStream_t)stream>>>(z, nullptr, rows, C4, nullptr, nullptr,
                                                                                             nullptr, nullptr, sums_zeroed);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

static int bn_relu_apply_impl(const float* z, float* y, size_t rows, int channels, const double* sums, const float* gamma,
                              const float* beta, const float* conv_bias, float eps, float momentum, float* running_mean,
                              float* running_var, float* save_mean, float* save_rstd, int sums_about_zero, void* stream) {
    int rc = bn_check(z, y, rows, channels, "iris_bn_relu_apply");
    if (rc) return rc;
    if (!sums || !gamma || !beta || !running_mean || !running_var || !save_mean || !save_rstd)
        return fail(IRIS_E_INVALID, "iris_bn_relu_apply: NULL argument");
    // every block re-reads K = row 0 of z (the shift of the sums) while other blocks write y: in place, a block could
    // overwrite row 0 before its neighbours have read K
    if (bn_overlap(z, rows * (size_t)channels, y, rows * (size_t)channels)) return fail(IRIS_E_INVALID, "iris_bn_relu_apply: y must not overlap z (not an in-place op)");
    const size_t n4 = rows * (size_t)(channels / 4);
    const double m = (double)rows;
    k_bn_relu_apply<<<grid_bn(n4), 256, 2 * (size_t)channels * sizeof(float), (hipStream_t)stream>>>(z, y, n4, channels / 4, 1.0 / m, rows > 1 ? m / (m - 1.0) : 1.0, sums,
                                                                   gamma, beta, conv_bias, eps, momentum, running_mean, running_var,
                                                                   save_mean, save_rstd, sums_about_zero);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_bn_relu_apply(const float* z, float* y, size_t rows, int channels, const double* sums, const float* gamma,
                                  const float* beta, const float* conv_bias, float eps, float momentum, float* running_mean,
                                  float* running_var, float* save_mean, float* save_rstd, void* stream) {
    return bn_relu_apply_impl(z, y, rows, channels, sums, gamma, beta, conv_bias, eps, momentum, running_mean, running_var, save_mean,
                              save_rstd, 0, stream);
}
// the same with `sums` = (sum z, sum z^2) as a convolution's epilogue accumulated them (iris_conv3x3_*_bn): no iris_bn_stats pass
extern "C" int iris_bn_relu_apply_sums0(const float* z, float* y, size_t rows, int channels, const double* sums, const float* gamma,
                                        const float* beta, const float* conv_bias, float eps, float momentum, float* running_mean,
                                        float* running_var, float* save_mean, float* save_rstd, void* stream) {
    return bn_relu_apply_impl(z, y, rows, channels, sums, gamma, beta, conv_bias, eps, momentum, running_mean, running_var, save_mean,
                              save_rstd, 1, stream);
}

extern "C" int iris_bn_relu_bwd_reduce(const float* z, const float* dy, size_t rows, int channels, const float* save_mean,
                                       const float* save_rstd, const float* gamma, const float* beta, double* sums_zeroed,
                                       void* stream) {
    int rc = bn_check(z, sums_zeroed, rows, channels, "iris_bn_relu_bwd_reduce");
    if (rc) return rc;
    if (!dy || !save_mean || !save_rstd || !gamma || !beta) return fail(IRIS_E_INVALID, "iris_bn_relu_bwd_reduce: NULL argument");
    const int C4 = channels / 4;
    k_bn_reduce<true><<<bn_reduce_grid(rows, C4), 256, bn_reduce_lds(C4), (hipStream_t)stream>>>(z, dy, rows, C4, save_mean, save_rstd, gamma,
                                                                                            beta, sums_zeroed);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_bn_relu_bwd_dx(const float* z, const float* dy, float* dz, size_t rows, int channels, const float* save_mean,
                                   const float* save_rstd, const float* gamma, const float* beta, const double* sums,
                                   float* dgamma, float* dbeta, void* stream) {
    int rc = bn_check(z, dz, rows, channels, "iris_bn_relu_bwd_dx");
    if (rc) return rc;
    if (!dy || !save_mean || !save_rstd || !gamma || !beta || !sums || !dgamma || !dbeta)
        return fail(IRIS_E_INVALID, "iris_bn_relu_bwd_dx: NULL argument");
    const size_t n4 = rows * (size_t)(channels / 4);
    k_bn_relu_bwd_dx<<<grid_bn(n4), 256, 4 * (size_t)channels * sizeof(float), (hipStream_t)stream>>>(
        z, dy, dz, n4, channels / 4, (float)(1.0 / (double)rows), save_mean, save_rstd, gamma, beta, sums, dgamma, dbeta);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

// -- the same three passes with MaxPool 2x2 / stride 2 / 'same' folded in (z [B, H, W, C]; p, dp [B, ceil(H/2), ceil(W/2), C])
static int bn_pool_check(const void* a, const void* b, int batch, int height, int width, int channels, const char* who) {
    if (batch <= 0 || height <= 0 || width <= 0) return fail(IRIS_E_INVALID, "%s: empty tensor", who);
    if ((double)batch * ((height + 1) / 2) * ((width + 1) / 2) * (double)std::max(channels, 4) / 4 >= 2147483648.0)
        return fail(IRIS_E_UNSUPPORTED, "%s: more than 2^31 pooled float4 elements", who);
    return bn_check(a, b, (size_t)batch * height * width, channels, who);
}

static int bn_relu_pool_apply_impl(const float* z, float* p, int batch, int height, int width, int channels, const double* sums,
                                   const float* gamma, const float* beta, const float* conv_bias, float eps, float momentum,
                                   float* running_mean, float* running_var, float* save_mean, float* save_rstd, int sums_about_zero,
                                   void* stream) {
    int rc = bn_pool_check(z, p, batch, height, width, channels, "iris_bn_relu_pool_apply");
    if (rc) return rc;
    if (bn_overlap(z, (size_t)batch * height * width * channels, p, (size_t)batch * ((height + 1) / 2) * ((width + 1) / 2) * channels))
        return fail(IRIS_E_INVALID, "iris_bn_relu_pool_apply: p must not overlap z");
    if (!sums || !gamma || !beta || !running_mean || !running_var || !save_mean || !save_rstd)
        return fail(IRIS_E_INVALID, "iris_bn_relu_pool_apply: NULL argument");
    const size_t n4 = (size_t)batch * ((height + 1) / 2) * ((width + 1) / 2) * (channels / 4);
    const double m = (double)batch * height * width;  // the statistics are those of the full-size activation
    k_bn_relu_pool_apply<<<grid_bn(n4), 256, 2 * (size_t)channels * sizeof(float), (hipStream_t)stream>>>(
        z, p, batch, height, width, channels / 4, 1.0 / m, m > 1.0 ? m / (m - 1.0) : 1.0, sums, gamma, beta, conv_bias, eps, momentum,
        running_mean, running_var, save_mean, save_rstd, sums_about_zero);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_bn_relu_pool_apply(const float* z, float* p, int batch, int height, int width, int channels, const double* sums,
                                       const float* gamma, const float* beta, const float* conv_bias, float eps, float momentum,
                                       float* running_mean, float* running_var, float* save_mean, float* save_rstd, void* stream) {
    return bn_relu_pool_apply_impl(z, p, batch, height, width, channels, sums, gamma, beta, conv_bias, eps, momentum, running_mean,
                                   running_var, save_mean, save_rstd, 0, stream);
}
extern "C" int iris_bn_relu_pool_apply_sums0(const float* z, float* p, int batch, int height, int width, int channels, const double* sums,
                                             const float* gamma, const float* beta, const float* conv_bias, float eps, float momentum,
                                             float* running_mean, float* running_var, float* save_mean, float* save_rstd, void* stream) {
    return bn_relu_pool_apply_impl(z, p, batch, height, width, channels, sums, gamma, beta, conv_bias, eps, momentum, running_mean,
                                   running_var, save_mean, save_rstd, 1, stream);
}

extern "C" int iris_bn_relu_pool_bwd_reduce(const float* z, const float* dp, int batch, int height, int width, int channels,
                                            const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                                            double* sums_zeroed, void* stream) {
    int rc = bn_pool_check(z, sums_zeroed, batch, height, width, channels, "iris_bn_relu_pool_bwd_reduce");
    if (rc) return rc;
    if (!dp || !save_mean || !save_rstd || !gamma || !beta) return fail(IRIS_E_INVALID, "iris_bn_relu_pool_bwd_reduce: NULL argument");
    const int C4 = channels / 4;
    const size_t prow = (size_t)batch * ((height + 1) / 2) * ((width + 1) / 2);
    k_bn_pool_bwd_reduce<<<bn_reduce_grid(prow, C4, kBnPoolRows), 256, bn_reduce_lds(C4), (hipStream_t)stream>>>(
        z, dp, batch, height, width, C4, save_mean, save_rstd, gamma, beta, sums_zeroed);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_bn_relu_pool_bwd_dx(const float* z, const float* dp, float* dz, int batch, int height, int width, int channels,
                                        const float* save_mean, const float* save_rstd, const float* gamma, const float* beta,
                                        const double* sums, float* dgamma, float* dbeta, void* stream) {
    int rc = bn_pool_check(z, dz, batch, height, width, channels, "iris_bn_relu_pool_bwd_dx");
    if (rc) return rc;
    if (!dp || !save_mean || !save_rstd || !gamma || !beta || !sums || !dgamma || !dbeta)
        return fail(IRIS_E_INVALID, "iris_bn_relu_pool_bwd_dx: NULL argument");
    const size_t n4 = (size_t)batch * ((height + 1) / 2) * ((width + 1) / 2) * (channels / 4);
    const double m = (double)batch * height * width;
    k_bn_relu_pool_bwd_dx<<<grid_bn(n4), 256, 4 * (size_t)channels * sizeof(float), (hipStream_t)stream>>>(
        z, dp, dz, batch, height, width, channels / 4, (float)(1.0 / m), save_mean, save_rstd, gamma, beta, sums, dgamma, dbeta);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_plan_kernel_name(const iris_plan* p, int with_bands, char* out, int capacity) {
    if (!p || !out || capacity <= 0) return fail(IRIS_E_INVALID, "iris_plan_kernel_name: bad argument");
    if (p->mel_only) return fail(IRIS_E_UNSUPPORTED, "iris_plan_kernel_name: mel-only plan");
    if (p->mel_precision == 1 && !with_bands)
        snprintf(out, (size_t)capacity, "k_wav_to_mel_mfma<%d>", p->log2n);
    else
        snprintf(out, (size_t)capacity, "k_wav_to_mel<%d,%d,%s,%s,1,%d>", p->log2n, p->mel_mode,
                 (p->need_hi && p->mel_mode != 0 && p->mel_mode != 3) ? "true" : "false", with_bands ? "true" : "false",
                 p->epilogue == IRIS_EPILOGUE_FUSED ? 1 : (p->epilogue == IRIS_EPILOGUE_IN_PLACE ? 2 : 0));
    return IRIS_OK;
}

extern "C" int iris_timing_enable(iris_plan* p, int enable) {
    if (!p) return fail(IRIS_E_INVALID, "iris_timing_enable: NULL plan");
    p->timing = enable > 0 ? enable : 0;
    p->launch_no = 0;
    p->ev_used = 0;
    p->ev2_used = 0;
    return IRIS_OK;
}

static int read_events(const std::vector<hipEvent_t>& ev, int used, float* out_ms, int capacity, double* total) {
    *total = 0.0;
    for (int i = 0; i < used; ++i) {
        HIP_TRY(hipEventSynchronize(ev[2 * i + 1]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
        *total += ms;
        if (out_ms && i < capacity) out_ms[i] = ms;
    }
    return IRIS_OK;
}

extern "C" int iris_timing_read(iris_plan* p, int* n_launches, float* mean_ms) {
    if (!p || !n_launches || !mean_ms) return fail(IRIS_E_INVALID, "iris_timing_read: NULL argument");
    DeviceGuard guard(p->device);
    double total = 0.0;
    int rc = read_events(p->ev, p->ev_used, nullptr, 0, &total);
    if (rc) return rc;
    *n_launches = p->ev_used;
    *mean_ms = p->ev_used ? (float)(total / p->ev_used) : 0.f;
    p->ev_used = 0;
    p->ev2_used = 0;
    return IRIS_OK;
}

extern "C" int iris_timing_samples(iris_plan* p, int kernel, float* out_ms, int capacity, int* n_samples) {
    if (!p || !n_samples || (capacity > 0 && !out_ms)) return fail(IRIS_E_INVALID, "iris_timing_samples: NULL argument");
    if (kernel != 0 && kernel != 1) return fail(IRIS_E_INVALID, "iris_timing_samples: kernel must be 0 or 1");
    DeviceGuard guard(p->device);
    double total = 0.0;
    const int used = kernel == 0 ? p->ev_used : p->ev2_used;
    int rc = read_events(kernel == 0 ? p->ev : p->ev2, used, out_ms, capacity, &total);
    if (rc) return rc;
    *n_samples = used;
    return IRIS_OK;
}
