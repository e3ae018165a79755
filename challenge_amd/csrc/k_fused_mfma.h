// k_fused_mfma.h -- the fused hot path with the mel step on the matrix cores (fp16 inputs, fp32 accumulate).
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// K1m: waveform -> mel magnitudes, mel contraction as v_mfma_f32_16x16x32_f16   (BASELINE configs[4])
//   mel[m, frame] = sum_f W[f, m] |X[f, frame]|   (transforms.py:65, tf.tensordot over the bin axis)
//   as D[16 mel x 16 frames] += A[16 mel x 32 bins] . B[32 bins x 16 frames] per MFMA:
//     A = 0.5 W^T in fp16, one 16-band tile per wave, held in REGISTERS for the whole kernel: only the
//         k-steps (32 bins) that hold a non-zero of the tile are kept (a triangular filterbank touches
//         2-8 of them per tile; with the default 3800 Hz edge two thirds of the bins feed no band at all);
//     B = 2 |X| in fp16, written by the waves that transformed the frames into an LDS tile [8 frames][bins];
//     D = fp32, lane (n = lane & 15, r = lane >> 4) holds mel 4 r + i of frame n.
//   Workgroup = 8 waves, walking its chunk in groups of 8 wave-frames: every wave transforms one frame
//   (same FFT core as K1; next frame in flight by LDS-DMA), converts 2|X| to fp16 into the group's tile
//   (double-buffered: ONE workgroup barrier per group), then runs the MFMAs of its band tile over all 8
//   frames of the group (columns 8..15 of B repeat 0..7 and are dropped) and stores 4 bands x 8 frames.
//   Supported: n_fft 512 / 1024 / 2048, n_mel <= 128, bands within the lower half of the spectrum, <= 8
//   k-steps per tile, no SpecAugment bands (those calls take the fp32 kernel).  fp16 carries 11 bits: the
//   stated tolerance of this variant is 2e-3 relative (tests/test_frontend_gpu.py), not north_star's 1e-5 -
//   which is why the banded fp32 kernel stays the default.
// ---------------------------------------------------------------------------
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int kMfmaWaves = 8, kMfmaKsMax = 8, kMfmaGroup = 8;

// untangle (lower half only) + magnitude, stored as fp16 into `row` for bins k < kb_pad (a multiple of 64)
template <int LOG2N>
__device__ __forceinline__ void untangle_mag_half(const cf (&x)[FftCfg<LOG2N>::P], const cf* post, cf* lds, _Float16* row,
                                                  int kb_pad, int lane) {
    constexpr int P = FftCfg<LOG2N>::P;
    cf* wp = lds + lds_pad<untangle_pm(LOG2N)>(lane);
#pragma unroll
    for (int q = P / 2; q < P; ++q) wp[lds_pad<untangle_pm(LOG2N)>(kWave * q)] = x[q];
    wave_sync_lds();
    const cf* rp = lds + lds_pad<untangle_pm(LOG2N)>(kWave - lane);
    cf zp[P / 2];
#pragma unroll
    for (int q = 0; q < P / 2; ++q) zp[q] = rp[lds_pad<untangle_pm(LOG2N)>(kWave * (P - 1 - q))];
    if (lane == 0) zp[0] = x[0];  // k = 0 pairs with itself
#pragma unroll
    for (int q = 0; q < P / 2; ++q) {
        if (kWave * q < kb_pad) {  // wave-uniform
            const cf e = __builtin_elementwise_fma(zp[q], mk(1.0f, -1.0f), x[q]);  // 2 E
            const cf d = __builtin_elementwise_fma(zp[q], mk(-1.0f, 1.0f), x[q]);  // 2 i O
            const cf lo = e + cmul_mi_tw(d, post[q]);
            row[lane + kWave * q] = (_Float16)cabs_rn(lo);  // 2 |X[k]|
        }
    }
    wave_sync_lds();
}

template <int LOG2N>
__global__ __launch_bounds__(64 * kMfmaWaves, 2) void k_wav_to_mel_mfma(const FusedArgs a) {
    constexpr int N = 1 << LOG2N, NC = N / 2, P = FftCfg<LOG2N>::P, NTW = FftCfg<LOG2N>::NTW;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // LDS: landing buffers [8][N floats] | exchange buffers [8] | fp16 magnitude tiles [2][8][kb_pad + 8]
    constexpr int kXBufBytes = (lds_padded(NC, FftCfg<LOG2N>::PMMAX) * 8 + 15) & ~15;
    constexpr int kLandBytes = kMfmaWaves * N * 4;
    const float* fbuf = reinterpret_cast<const float*>(smem + wv * (N * 4));
    const unsigned fbuf_lds =
        __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(smem + wv * (N * 4)));
    cf* lds = reinterpret_cast<cf*>(smem + kLandBytes + wv * kXBufBytes);
    const int kb_pad = (a.kb + 63) & ~63, row_h = kb_pad + 8;  // +8 halfs: rows start 16 bytes apart modulo the banks
    _Float16* magh = reinterpret_cast<_Float16*>(smem + kLandBytes + kMfmaWaves * kXBufBytes);

    auto chunk_clip = [&](int chunk) { return chunk / a.chunks_per_clip; };
    auto chunk_t0 = [&](int chunk, int b) {
        const int ci = chunk - b * a.chunks_per_clip;
        return ci * a.chunk_base + min(ci, a.chunk_rem);
    };
    auto chunk_nt = [&](int chunk, int b) { return a.chunk_base + ((chunk - b * a.chunks_per_clip) < a.chunk_rem ? 1 : 0); };
    auto issue = [&](int b, int t0, int f) {  // LDS-DMA of wave-frame f = tl * C + c into this wave's landing buffer
        const int tl = (a.C == 1) ? f : f / a.C, c = f - tl * a.C;
        dma_frame<LOG2N>(a.wav + ((size_t)b * a.C + c) * a.L, a.L, (t0 + tl) * a.hop - N / 2, fbuf_lds, lane);
    };

    const int g0 = xcd_remap(blockIdx.x, gridDim.x);
    if (g0 < a.n_chunks) {  // the first frame: in flight while the constants are fetched
        const int b = chunk_clip(g0);
        if (wv < chunk_nt(g0, b) * a.C) issue(b, chunk_t0(g0, b), wv);
    }
    cf tw[NTW], post[P / 2], win[P];
    {
        float wreg_unused[kMelRegs];
        int lo_unused;
        float4* stage = reinterpret_cast<float4*>(smem + kLandBytes);  // through the exchange buffers
        const float4* g = reinterpret_cast<const float4*>(a.consts);
        for (int i = threadIdx.x; i < ConstLayout<LOG2N>::NV4 * kWave; i += blockDim.x) stage[i] = g[i];
        __syncthreads();
        load_consts<LOG2N>(reinterpret_cast<const float*>(stage), lane, tw, post, win, wreg_unused, lo_unused);
        __syncthreads();  // the exchange buffers are free again
    }
    // this wave's band tile: A fragments (0.5 W^T, fp16) of its k-steps, in registers for the whole kernel
    const int tile = wv, n_tiles = (a.M + 15) >> 4;
    int ks_lo = 0, nks = 0;
    h8 afrag[kMfmaKsMax];
    if (tile < n_tiles) {
        ks_lo = a.tile_ks[2 * tile];
        nks = a.tile_ks[2 * tile + 1];
    }
    {
        const h8* wf = reinterpret_cast<const h8*>(a.wfrag) + ((size_t)tile * kMfmaKsMax) * kWave + lane;
#pragma unroll
        for (int j = 0; j < kMfmaKsMax; ++j) {
            h8 z = {0, 0, 0, 0, 0, 0, 0, 0};
            afrag[j] = (tile < n_tiles && j < nks) ? wf[(size_t)j * kWave] : z;
        }
    }

    for (int chunk = g0; chunk < a.n_chunks; chunk += gridDim.x) {
        const int b = chunk_clip(chunk);
        const int t0 = chunk_t0(chunk, b), nt = chunk_nt(chunk, b);
        const int nwf = nt * a.C;
        if (chunk != g0 && wv < nwf) issue(b, t0, wv);
        float scale = 1.0f;
        if (a.sumsq != nullptr) {
            float sq = 0.f;
            const float* ssq = a.sumsq + (size_t)b * a.n_sq;
            for (int i = lane; i < a.n_sq; i += kWave) sq += ssq[i];
            sq = wave_sum(sq);
            scale = 1.0f / (sqrtf(sq / ((float)a.C * (float)a.L)) * 10.0f);
        }
        const unsigned rowpitch_b = (unsigned)a.T * (unsigned)a.C * 4u;
        float* const chunk_out = a.out + ((size_t)b * a.M * a.T + t0) * a.C;
        float mn = INFINITY, mx = -INFINITY;
        const int n_groups = (nwf + kMfmaGroup - 1) / kMfmaGroup;
        for (int g = 0; g < n_groups; ++g) {
            const int f = g * kMfmaGroup + wv;
            _Float16* tile_h = magh + (size_t)(g & 1) * kMfmaGroup * row_h;
            if (f < nwf) {  // wave-uniform
                cf x[P];
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this frame's LDS-DMA has landed
                const cf* fb2 = reinterpret_cast<const cf*>(fbuf) + lane;
#pragma unroll
                for (int q = 0; q < P; ++q) x[q] = fb2[kWave * q];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (f + kMfmaGroup < nwf) issue(b, t0, f + kMfmaGroup);  // next frame: behind this one's transform
#pragma unroll
                for (int q = 0; q < P; ++q) x[q] *= win[q];
                // normalize: 1 / (10 rms) goes onto the samples BEFORE the transform - the magnitudes are cast to fp16
                // (max 65504) in front of the contraction, and un-normalised PCM-range input would overflow there
                if (a.sumsq != nullptr) {  // wave-uniform
#pragma unroll
                    for (int q = 0; q < P; ++q) x[q] *= scale;
                }
                fft_frame<LOG2N>(x, tw, lds, lane);
                untangle_mag_half<LOG2N>(x, post, lds, tile_h + (size_t)wv * row_h, kb_pad, lane);
            }
            __syncthreads();  // the group's magnitudes are in the tile (the other tile is free for the next group)
            if (tile < n_tiles) {
                f4 acc = {0.f, 0.f, 0.f, 0.f};
                // B fragment: bins 8 (lane >> 4) .. + 7 of k-step ks, frame slot lane & 7 (columns 8..15 repeat 0..7)
                const _Float16* brow = tile_h + (size_t)(lane & 7) * row_h + 8 * (lane >> 4);
#pragma unroll
                for (int j = 0; j < kMfmaKsMax; ++j) {
                    if (j < nks) {  // wave-uniform
                        const h8 bf = *reinterpret_cast<const h8*>(brow + (ks_lo + j) * 32);
                        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(afrag[j], bf, acc, 0, 0, 0);
                    }
                }
                const int n = lane & 15, fo = g * kMfmaGroup + n;
                if (n < kMfmaGroup && fo < nwf) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int m = tile * 16 + 4 * (lane >> 4) + i;
                        if (m < a.M) {
                            const float v = acc[i];
                            chunk_out[(size_t)m * (rowpitch_b >> 2) + fo] = v;
                            mn = fminf(mn, v);
                            mx = fmaxf(mx, v);
                        }
                    }
                }
            }
        }
        mn = wave_min(mn);
        mx = wave_max(mx);
        if (lane == 0) {
            a.partial[((size_t)chunk * kMfmaWaves + wv) * 2 + 0] = mn;
            a.partial[((size_t)chunk * kMfmaWaves + wv) * 2 + 1] = mx;
        }
        __syncthreads();  // the next chunk restarts at tile 0: every wave must be done reading this chunk's last tile
    }
}
