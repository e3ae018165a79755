// k_stft.h -- STFT in the reference layout [B, F, T, 2C].
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// K3: STFT only, reference layout [B, F, T, 2C] (load_wav, data_utils.py:9-29)
//   One persistent workgroup per CU (as many waves as the registers allow, like K1) takes a chunk
//   of consecutive frames of one clip (K1's balanced geometry) and walks it in tiles of as many
//   frames as the LDS holds, a whole number of rounds of its waves.  Consecutive tiles extend the
//   same rows of the output, so their runs meet in the L2.  Per tile: every wave transforms its frames
//   (wave-per-frame FFT core, both untangle halves) into an LDS tile [F][frames * 2C (+1)], a
//   barrier, then each wave writes its share of the F rows as contiguous [t, 2C] runs - the
//   layout's time axis is the fast one, a frame's bins are a row pitch apart, so the tile is
//   what turns per-frame spectra into coalesced stores.  The next frame's samples are loaded
//   while the current frame's spectrum goes to the tile.
// ---------------------------------------------------------------------------
struct StftArgs {
    const float* wav;
    float* spec;
    const float* consts;
    int B, C, L, T, hop;
    int tile_frames;                 // capacity of the LDS tile (frames)
    int chunks_per_clip, n_chunks;   // balanced split of a clip's T frames: sizes differ by at most one
    int chunk_base, chunk_rem;
    const float* sumsq;  // nullable [B, n_sq] partial sums of squares of the clip (IRIS_F_NORMALIZE)
    int n_sq;
};

// (n_fft 2048 keeps 16 points per lane and both spectrum halves: it needs more than 256 registers)
// Workgroups per CU: with two half-size workgroups the transform phase of one overlaps the write-out phase of
// the other (inside a workgroup the two phases are separated by barriers, so all its waves are in the same one).
#ifndef IRIS_STFT_NT
#define IRIS_STFT_NT 0
#endif
#if IRIS_STFT_NT
#define STFT_ST "global_store_dword %0, %1, %2 nt"
#else
#define STFT_ST "global_store_dword %0, %1, %2"
#endif
#ifndef IRIS_STFT_WGS
#define IRIS_STFT_WGS 1
#endif
constexpr int stft_wgs(int log2n) { return log2n >= 11 ? 1 : IRIS_STFT_WGS; }
constexpr int stft_waves(int log2n) { return (log2n >= 11 ? 4 : (log2n <= 9 ? 16 : 12)) / stft_wgs(log2n); }

template <int LOG2N>
__global__ __launch_bounds__(64 * stft_waves(LOG2N), stft_waves(LOG2N) * stft_wgs(LOG2N) / 4) void k_stft(const StftArgs a) {
    constexpr int W = stft_waves(LOG2N);
    constexpr int N = 1 << LOG2N, NC = N / 2, P = FftCfg<LOG2N>::P, NTW = FftCfg<LOG2N>::NTW;
    constexpr int F = NC + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int C2 = 2 * a.C;
    const int row = a.tile_frames * C2 + 1;  // odd stride: conflict-free column writes

    constexpr int kWaveBufBytes = (lds_padded(NC, FftCfg<LOG2N>::PMMAX) * 8 + 15) & ~15;
    cf* lds = reinterpret_cast<cf*>(smem + wv * kWaveBufBytes);
    float* tile_out = reinterpret_cast<float*>(smem + W * kWaveBufBytes);  // [F][row]

    // chunk -> clip, first frame, frame count
    auto chunk_clip = [&](int ch) { return ch / a.chunks_per_clip; };
    auto chunk_t0 = [&](int ch, int b) {
        const int ci = ch - b * a.chunks_per_clip;
        return ci * a.chunk_base + min(ci, a.chunk_rem);
    };
    auto chunk_nt = [&](int ch, int b) { return a.chunk_base + ((ch - b * a.chunks_per_clip) < a.chunk_rem ? 1 : 0); };
    cf x[P];
    auto fetch = [&](int b, int t0, int f) {  // wave-frame f = tl * C + c of the tile starting at t0
        const int tl = (a.C == 1) ? f : f / a.C, c = f - tl * a.C;
        load_frame<LOG2N>(x, a.wav + ((size_t)b * a.C + c) * a.L, a.L, (t0 + tl) * a.hop - N / 2, lane);
    };

    // Walk order of a chunk's tiles (round 4): chunks with an ODD index inside their clip walk their tiles BACKWARDS.  The
    // layout's row pitch (T 2C floats) is in general no multiple of the 128-byte line, so a run of a row starts and ends
    // inside a line, and the line at a chunk edge is completed by the NEIGHBOURING chunk's workgroup.  With every chunk
    // walking forwards the two halves of that line were written a whole chunk time apart (the first tile of chunk k + 1 at
    // the start of the kernel, the last tile of chunk k at its end): the half-written line left the L2 in between and went to
    // HBM as a masked write - twice.  With alternating directions both neighbours of an edge reach it at the same end of
    // their walk (an even number of chunks per clip also pairs a row's end with the next row's start), so the halves meet in
    // the clip's L2 (a clip's chunks share an XCD) - no transform is recomputed, no data exchanged.
    // step i of a chunk of c_nt frames -> (first frame offset inside the chunk, frames)
    auto tile_of = [&](int i, int c_nt, bool backward, int& tt, int& nt) {
        const int n_tiles = (c_nt + a.tile_frames - 1) / a.tile_frames;
        const int k = backward ? n_tiles - 1 - i : i;
        tt = k * a.tile_frames;
        nt = min(a.tile_frames, c_nt - tt);
    };
    const int g0 = xcd_remap(blockIdx.x, gridDim.x);
    if (g0 < a.n_chunks) {  // first frame: in flight while the constants are fetched
        const int b = chunk_clip(g0);
        int tt, nt;
        tile_of(0, chunk_nt(g0, b), ((g0 - b * a.chunks_per_clip) & 1) != 0, tt, nt);
        if (wv < nt * a.C) fetch(b, chunk_t0(g0, b) + tt, wv);
    }
    cf tw[NTW], post[P / 2], win[P];
    {
        // the constant block is the same for every wave: global -> LDS once per workgroup (through
        // the tile, idle until the first spectrum), then every wave reads its lanes' share
        float4* stage = reinterpret_cast<float4*>(tile_out);
        const float4* g = reinterpret_cast<const float4*>(a.consts);
        for (int i = threadIdx.x; i < ConstLayout<LOG2N>::NV4 * kWave; i += blockDim.x) stage[i] = g[i];
        __syncthreads();
        float wreg_unused[kMelRegs];
        int lo_unused;
        load_consts<LOG2N>(reinterpret_cast<const float*>(stage), lane, tw, post, win, wreg_unused, lo_unused);
        __syncthreads();
    }

    for (int chunk = g0; chunk < a.n_chunks; chunk += gridDim.x) {
      const int b = chunk_clip(chunk);
      const int c_t0 = chunk_t0(chunk, b), c_nt = chunk_nt(chunk, b);
      const bool backward = ((chunk - b * a.chunks_per_clip) & 1) != 0;
      const int n_tiles = (c_nt + a.tile_frames - 1) / a.tile_frames;
      float scale = 1.0f;  // normalize: the STFT is linear, so 1 / (10 rms) scales the spectrum as it is written
      if (a.sumsq != nullptr) {
          float sq = 0.f;
          const float* ssq = opaque(a.sumsq) + (size_t)b * a.n_sq;
          for (int i = lane; i < a.n_sq; i += kWave) sq += ssq[i];
          sq = wave_sum(sq);
          scale = 1.0f / (sqrtf(sq / ((float)a.C * (float)a.L)) * 10.0f);
      }
      if (chunk != g0) {
          int tt0, nt0;
          tile_of(0, c_nt, backward, tt0, nt0);
          if (wv < nt0 * a.C) fetch(b, c_t0 + tt0, wv);
      }
      for (int ti = 0; ti < n_tiles; ++ti) {
        int tt, nt;
        tile_of(ti, c_nt, backward, tt, nt);
        const int t0 = c_t0 + tt;
        const int nwf = nt * a.C;
        for (int f = wv; f < nwf; f += W) {
#pragma unroll
            for (int q = 0; q < P; ++q) x[q] *= win[q];
            fft_frame<LOG2N>(x, tw, lds, lane);
            cf xlo[P / 2], xhi[P / 2];
            untangle<LOG2N, true, true>(x, post, lds, lane, xlo, xhi);
            const cf mid = mk(x[P / 2].x, -x[P / 2].y);  // X[NC/2] = conj(Z[NC/2]) (lane 0)
            // x is dead: the next frame (of this tile, or the first of the chunk's next tile) loads
            // behind the tile writes and the write-out
            if (f + W < nwf) {
                fetch(b, t0, f + W);
            } else if (ti + 1 < n_tiles) {
                int ttn, ntn;
                tile_of(ti + 1, c_nt, backward, ttn, ntn);
                if (wv < ntn * a.C) fetch(b, c_t0 + ttn, wv);
            }
            const int tl = (a.C == 1) ? f : f / a.C, c = f - tl * a.C;
            float* col = tile_out + tl * C2 + c;
#pragma unroll
            for (int q = 0; q < P / 2; ++q) {
                const int k = lane + kWave * q;
                col[k * row] = xlo[q].x;
                col[k * row + a.C] = xlo[q].y;
                col[(NC - k) * row] = xhi[q].x;  // k = 0 -> Nyquist bin NC
                col[(NC - k) * row + a.C] = xhi[q].y;
            }
            if (lane == 0) {
                col[(NC / 2) * row] = mid.x;
                col[(NC / 2) * row + a.C] = mid.y;
            }
        }
        if (wv >= nwf && ti + 1 < n_tiles) {  // a wave without a frame in this tile (a backward walk starts with the chunk's short
            int ttn, ntn;                      // tile) still needs its first frame of the next one
            tile_of(ti + 1, c_nt, backward, ttn, ntn);
            if (wv < ntn * a.C) fetch(b, c_t0 + ttn, wv);
        }
        __syncthreads();
        // write-out: wave w owns rows w, w + W, ...; a row is one contiguous run of nt * 2C floats.
        // Four rows' LDS reads are in flight before the first store; the stores take a uniform
        // row base (SGPR pair) and a per-lane byte offset.
        const int run = nt * C2;
        const size_t pitch = (size_t)a.T * C2;  // floats between rows of spec
        float* const out0 = a.spec + ((size_t)b * F * a.T + t0) * C2;
        for (int r0 = 0; r0 < run; r0 += kWave) {
            const int r = r0 + lane;
            if (r < run) {
                const unsigned off = (unsigned)r * 4u;
                int k = wv;
                for (; k + 3 * W < F; k += 4 * W) {
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = tile_out[(k + j * W) * row + r] * scale;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        asm volatile(STFT_ST ::"v"(off), "v"(v[j]),
                                     "s"(out0 + (size_t)(k + j * W) * pitch)
                                     : "memory");
                }
                for (; k < F; k += W) {
                    const float v = tile_out[k * row + r] * scale;
                    asm volatile(STFT_ST ::"v"(off), "v"(v), "s"(out0 + (size_t)k * pitch)
                                 : "memory");
                }
            }
        }
        __syncthreads();  // the tile is free again
      }
    }
}
