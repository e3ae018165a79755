// What does the STFT layout's write pattern cost with no transform in front of it?  
// k_stft writes [B, F, T, 2C]: a workgroup that owns a chunk of a clip's frames emits, per tile of `tf` frames, F runs
// of tf * 2C floats, a row pitch (T * 2C * 4 B) apart.  This kernel issues exactly those stores (constant data, the
// same wave -> row assignment, the same one-workgroup-per-CU grid, chunks of T / 8 frames) and nothing else, for
// several tile lengths and for a row pitch that is / is not a multiple of the 128-byte line.
// Build: hipcc -O3 --offload-arch=gfx950 stft_write_pattern.hip -o stft_write_pattern     Run on an MI355X.
#pragma clang diagnostic ignored "-Wunused-value"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(768) void k(float* spec, int B, int F, int T, int C2, int tf, int chunks_per_clip) {
    const int W = blockDim.x >> 6, wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n_chunks = B * chunks_per_clip;
    const int base = T / chunks_per_clip, rem = T % chunks_per_clip;
    for (int ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const int b = ch / chunks_per_clip, ci = ch % chunks_per_clip;
        const int c_t0 = ci * base + min(ci, rem), c_nt = base + (ci < rem ? 1 : 0);
        for (int tt = 0; tt < c_nt; tt += tf) {
            const int nt = min(tf, c_nt - tt), run = nt * C2;
            float* out0 = spec + ((size_t)b * F * T + c_t0 + tt) * C2;
            const size_t pitch = (size_t)T * C2;
            for (int r0 = 0; r0 < run; r0 += 64) {
                const int r = r0 + lane;
                if (r < run)
                    for (int kk = wv; kk < F; kk += W) out0[(size_t)kk * pitch + r] = (float)(kk + r);
            }
            __syncthreads();
        }
    }
}

// The same bytes, but every row flushes whole G-float lines only: after each round of `step` frames a row writes the
// lines completed so far; the partial head / tail of the chunk go with the first / last flush.  Rows sit at different
// phases of the line (the pitch is not a multiple of it), with period `per` in the row index: a wave walks its slice of
// the rows class by class (one flush window per class, scalar), two rows per store instruction when the window is one
// line, uniform row base + per-lane offset as k_stft's write-out has.
template <int G>
__global__ __launch_bounds__(768) void k_lines(float* spec, int B, int F, int T, int C2, int step, int chunks_per_clip,
                                               int per) {
    const int W = blockDim.x >> 6, wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n_chunks = B * chunks_per_clip;
    const int base = T / chunks_per_clip, rem = T % chunks_per_clip;
    const int rows_per_wave = (F + W - 1) / W;
    const int k_lo = wv * rows_per_wave, k_hi = min(F, k_lo + rows_per_wave);
    const int rowstep = T * C2, d = rowstep & (G - 1);
    for (int ch = blockIdx.x; ch < n_chunks; ch += gridDim.x) {
        const int b = ch / chunks_per_clip, ci = ch % chunks_per_clip;
        const int c_t0 = ci * base + min(ci, rem), c_nt = base + (ci < rem ? 1 : 0);
        const size_t g0 = ((size_t)b * F * T + c_t0) * C2;
        const int ph0 = (int)(g0 & (G - 1));
        for (int done0 = 0; done0 < c_nt; done0 += step) {
            const int done = min(done0 + step, c_nt);
            for (int c = 0; c < per; ++c) {
                const int k0 = k_lo + c;  // rows k0, k0 + per, ... share the phase
                if (k0 >= k_hi) break;
                const int ph = (ph0 + k0 * d) & (G - 1);
                auto upto = [&](int dd) { return dd >= c_nt ? c_nt * C2 : max(0, ((ph + dd * C2) & ~(G - 1)) - ph); };
                const int s0 = done0 == 0 ? 0 : upto(done0), s1 = upto(done), w = s1 - s0;
                if (w <= 0) continue;
                float* rowp = spec + g0 + (size_t)k0 * rowstep + s0;
                if (w <= 32) {
                    const int sub = lane >> 5, r = lane & 31;
                    const unsigned off = (unsigned)(sub * per * rowstep + r) * 4u;
                    for (int k = k0; k < k_hi; k += 2 * per) {
                        if (r < w && k + sub * per < k_hi)
                            asm volatile("global_store_dword %0, %1, %2" ::"v"(off), "v"((float)k), "s"(rowp) : "memory");
                        rowp += (size_t)2 * per * rowstep;
                    }
                } else {
                    for (int k = k0; k < k_hi; k += per) {
                        for (int r = lane; r < w; r += 64)
                            asm volatile("global_store_dword %0, %1, %2" ::"v"((unsigned)r * 4u), "v"((float)k), "s"(rowp)
                                         : "memory");
                        rowp += (size_t)per * rowstep;
                    }
                }
            }
            __syncthreads();
        }
    }
}

int main() {
    const int B = 32, F = 513, C2 = 2, NBUF = 6;
    for (int T : {626, 640}) {
        const size_t n = (size_t)B * F * T * C2;
        std::vector<float*> bufs(NBUF);
        for (auto& p : bufs) hipMalloc(&p, n * 4);
        for (int tf : {8, 16, 24, 32, 48, 64, 128}) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0);
            hipEventCreate(&e1);
            for (int i = 0; i < 12; ++i) k<<<256, 768>>>(bufs[i % NBUF], B, F, T, C2, tf, 8);
            hipEventRecord(e0);
            const int reps = 60;
            for (int i = 0; i < reps; ++i) k<<<256, 768>>>(bufs[i % NBUF], B, F, T, C2, tf, 8);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double us = ms * 1e3 / reps;
            printf("T %d (pitch %d B, %s) tile %3d frames = %4d-byte runs: %6.1f us per %5.1f MB = %5.2f TB/s\n", T, T * C2 * 4,
                   (T * C2 * 4) % 128 ? "unaligned" : "128B-aligned", tf, tf * C2 * 4, us, n * 4 / 1e6, n * 4 / us / 1e6);
        }
        for (int G : {32, 16})
            for (int step : {12, 24}) {
                hipEvent_t e0, e1;
                hipEventCreate(&e0);
                hipEventCreate(&e1);
                auto go = [&](float* p) {
                    int dd = (T * C2) & (G - 1), g = G;
                    while (dd) { const int t = g % dd; g = dd; dd = t; }  // gcd(T * C2 mod G, G)
                    const int per = G / g;
                    if (G == 32) k_lines<32><<<256, 768>>>(p, B, F, T, C2, step, 8, per);
                    else k_lines<16><<<256, 768>>>(p, B, F, T, C2, step, 8, per);
                };
                for (int i = 0; i < 12; ++i) go(bufs[i % NBUF]);
                hipEventRecord(e0);
                const int reps = 60;
                for (int i = 0; i < reps; ++i) go(bufs[i % NBUF]);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double us = ms * 1e3 / reps;
                printf("T %d whole %d-byte lines per row, flushed every %2d frames: %6.1f us = %5.2f TB/s\n", T, G * 4, step, us,
                       n * 4 / us / 1e6);
            }
        for (auto p : bufs) hipFree(p);
    }
    // reference point: the same bytes as one contiguous fill
    {
        const size_t n = (size_t)B * F * 626 * C2;
        float* p;
        hipMalloc(&p, n * 4 * NBUF);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int i = 0; i < 6; ++i) hipMemsetAsync(p + (size_t)(i % NBUF) * n, 0, n * 4);
        hipEventRecord(e0);
        for (int i = 0; i < 60; ++i) hipMemsetAsync(p + (size_t)(i % NBUF) * n, 0, n * 4);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("contiguous fill (hipMemsetAsync) of the same %5.1f MB: %6.1f us = %5.2f TB/s\n", n * 4 / 1e6, ms * 1e3 / 60,
               n * 4 / (ms * 1e3 / 60) / 1e6);
    }
    return 0;
}
