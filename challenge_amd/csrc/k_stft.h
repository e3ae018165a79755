// k_stft.h -- STFT in the reference layout [B, F, T, 2C].
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// K2: STFT only, reference layout [B, F, T, 2C]
// ---------------------------------------------------------------------------
struct StftArgs {
    const float* wav;
    float* spec;
    const float* consts;
    int B, C, L, T, hop, tile_frames, tiles_per_clip;
};

template <int LOG2N>
__global__ __launch_bounds__(256) void k_stft(const StftArgs a) {
    constexpr int N = 1 << LOG2N, NC = N / 2, P = FftCfg<LOG2N>::P, NTW = FftCfg<LOG2N>::NTW;
    constexpr int F = NC + 1;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wg = xcd_remap(blockIdx.x, gridDim.x);
    const int b = wg / a.tiles_per_clip, tile = wg - b * a.tiles_per_clip;
    const int t0 = tile * a.tile_frames;
    const int nt = min(a.tile_frames, a.T - t0);
    const int C2 = 2 * a.C;
    const int row = a.tile_frames * C2 + 1;  // odd stride: conflict-free column writes

    constexpr int kWaveBufBytes = (lds_padded(NC, FftCfg<LOG2N>::PMMAX) * 8 + 15) & ~15;
    cf* lds = reinterpret_cast<cf*>(smem + wv * kWaveBufBytes);
    float* tile_out = reinterpret_cast<float*>(smem + 4 * kWaveBufBytes);  // [F][row]

    cf tw[NTW], post[P / 2], win[P];
    float wreg_unused[kMelRegs];
    int lo_unused;
    load_consts<LOG2N>(a.consts, lane, tw, post, win, wreg_unused, lo_unused);

    const int nwf = nt * a.C;
    for (int f = wv; f < nwf; f += 4) {
        const int tl = f / a.C, c = f - tl * a.C;
        const float* clip = a.wav + ((size_t)b * a.C + c) * a.L;
        cf x[P];
        load_frame<LOG2N>(x, clip, a.L, (t0 + tl) * a.hop - N / 2, lane);
#pragma unroll
        for (int q = 0; q < P; ++q) x[q] *= win[q];
        fft_frame<LOG2N>(x, tw, lds, lane);
        cf xlo[P / 2], xhi[P / 2];
        untangle<LOG2N, true, true>(x, post, lds, lane, xlo, xhi);
        const int col = tl * C2 + c;
#pragma unroll
        for (int q = 0; q < P / 2; ++q) {
            const int k = lane + kWave * q;
            tile_out[k * row + col] = xlo[q].x;
            tile_out[k * row + col + a.C] = xlo[q].y;
            tile_out[(NC - k) * row + col] = xhi[q].x;  // k = 0 -> Nyquist bin NC
            tile_out[(NC - k) * row + col + a.C] = xhi[q].y;
        }
        if (lane == 0) {  // X[NC/2] = conj(Z[NC/2])
            tile_out[(NC / 2) * row + col] = x[P / 2].x;
            tile_out[(NC / 2) * row + col + a.C] = -x[P / 2].y;
        }
    }
    __syncthreads();
    const int run = nt * C2;
    for (int idx = threadIdx.x; idx < F * run; idx += blockDim.x) {
        const int k = idx / run, r = idx - k * run;
        a.spec[(((size_t)b * F + k) * a.T + t0) * C2 + r] = tile_out[k * row + r];
    }
}
