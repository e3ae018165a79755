// fft_pair.hip -- would a wave that transforms TWO frames at once, one register pair per real number (frame A in the
// low half, frame B in the high half: every v_pk_* instruction then serves two frames, and so does every LDS, scalar
// and wait instruction), beat the product's one-frame-per-wave core?  Compute + LDS only: frames come from LDS, the
// magnitudes go back to LDS, nothing touches global memory inside the loop.
//   A: the product core (iris_fft.h radix-8^3 on (re, im) pairs + untangle_mag), 16 waves per CU, 1 frame per wave
//   B: the same arithmetic on structure-of-arrays complex numbers of float2 = (frame A, frame B), 12 waves per CU
// hipcc -O3 --offload-arch=gfx950 -I../../challenge_amd/csrc fft_pair.hip -o fft_pair
#include "common.h"
#include "spectrum.h"

#include <cstdio>
#include <vector>

constexpr int NC = 512, P = 8;
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
struct cx { f2 re, im; };  // one complex point of two frames

__device__ __forceinline__ cx operator+(cx a, cx b) { return {a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cx operator-(cx a, cx b) { return {a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cx mul_mi(cx a) { return {a.im, -a.re}; }            // -i a (free: a rename and a sign)
__device__ __forceinline__ cx cmulw(cx a, float wr, float wi) {                  // a * (wr + i wi), w per lane
    return {a.re * wr - a.im * wi, a.re * wi + a.im * wr};
}
__device__ __forceinline__ void dft8(cx (&v)[8]) {
    constexpr float r = 0.70710678118654752f;
    const cx a0 = v[0] + v[4], a1 = v[1] + v[5], a2 = v[2] + v[6], a3 = v[3] + v[7];
    const cx d0 = v[0] - v[4], d1 = v[1] - v[5], d2 = v[2] - v[6], d3 = v[3] - v[7];
    {
        const cx s0 = a0 + a2, s1 = a1 + a3, t0 = a0 - a2, t1 = a1 - a3;
        v[0] = s0 + s1; v[4] = s0 - s1; v[2] = t0 + mul_mi(t1); v[6] = t0 - mul_mi(t1);
    }
    {
        const cx m1 = mul_mi(d1), m3 = mul_mi(d3);
        const cx e1 = {(d1.re + m1.re) * r, (d1.im + m1.im) * r};        // W8^1 d1
        const cx e3 = {(m3.re - d3.re) * r, (m3.im - d3.im) * r};        // W8^3 d3
        const cx s0 = d0 + mul_mi(d2), t0 = d0 - mul_mi(d2), s1 = e1 + e3, t1 = e1 - e3;
        v[1] = s0 + s1; v[5] = s0 - s1; v[3] = t0 + mul_mi(t1); v[7] = t0 - mul_mi(t1);
    }
}
__device__ __forceinline__ int pad8(int i) { return i + (i >> 3); }  // 16-byte slots: strided b128 writes conflict-free

template <int NS, bool LAST>
__device__ __forceinline__ void stage_pair(cx (&x)[8], const float* twr, const float* twi, f4* lds, int lane) {
    cx v[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) v[t] = x[t];
    if constexpr (NS > 1) {
#pragma unroll
        for (int t = 1; t < 8; ++t) v[t] = cmulw(v[t], twr[t - 1], twi[t - 1]);
    }
    dft8(v);
    if constexpr (LAST) {
#pragma unroll
        for (int t = 0; t < 8; ++t) x[t] = v[t];
    } else {
        const int base = (lane / NS) * (NS * 8) + (lane % NS);
#pragma unroll
        for (int t = 0; t < 8; ++t) lds[pad8(base + t * NS)] = (f4){v[t].re.x, v[t].re.y, v[t].im.x, v[t].im.y};
        wave_sync_lds();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const f4 p = lds[pad8(lane + 64 * q)];
            x[q] = {{p.x, p.y}, {p.z, p.w}};
        }
        wave_sync_lds();
    }
}

constexpr int kPairBuf = (NC + NC / 8 + 8) * 16;  // bytes per wave
template <int W>
__global__ __launch_bounds__(64 * W, W / 4) void k_pair(const float* consts, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f4* lds = reinterpret_cast<f4*>(smem + wv * kPairBuf);
    f2* mag = reinterpret_cast<f2*>(smem + W * kPairBuf + wv * (NC / 2 + 8) * 8);
    float twr[14], twi[14], pr[4], pi[4], win[16];
#pragma unroll
    for (int i = 0; i < 14; ++i) { twr[i] = consts[(i * 2) * 64 + lane]; twi[i] = consts[(i * 2 + 1) * 64 + lane]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { pr[i] = consts[(28 + 2 * i) * 64 + lane]; pi[i] = consts[(29 + 2 * i) * 64 + lane]; }
#pragma unroll
    for (int i = 0; i < 16; ++i) win[i] = consts[(36 + i) * 64 + lane];
    for (int q = 0; q < 8; ++q) lds[pad8(lane + 64 * q)] = (f4){0.01f * lane, 0.02f * q, -0.01f * q, 0.03f * lane};
    wave_sync_lds();
    f2 sum = {0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        cx x[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {  // "load" the two frames and window them
            const f4 p = lds[pad8(lane + 64 * q)];
            x[q] = {{p.x * win[2 * q], p.y * win[2 * q]}, {p.z * win[2 * q + 1], p.w * win[2 * q + 1]}};
        }
        wave_sync_lds();
        stage_pair<1, false>(x, nullptr, nullptr, lds, lane);
        stage_pair<8, false>(x, twr, twi, lds, lane);
        stage_pair<64, true>(x, twr + 7, twi + 7, lds, lane);
        // untangle: partners Z[NC - k] = lane 64 - l, register 7 - q (upper half exchanged), magnitudes of the lower half
#pragma unroll
        for (int q = 4; q < 8; ++q) lds[pad8(lane + 64 * q)] = (f4){x[q].re.x, x[q].re.y, x[q].im.x, x[q].im.y};
        wave_sync_lds();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f4 p = lds[pad8((64 - lane) + 64 * (7 - q))];
            const cx z = x[q], zp = {{p.x, p.y}, {p.z, p.w}};
            const f2 a = z.re + zp.re, b = z.re - zp.re, c = z.im + zp.im, d = z.im - zp.im;
            const f2 xr = a + c * pr[q] + b * pi[q], xi = d + c * pi[q] - b * pr[q];
            const f2 m2 = xr * xr + xi * xi;
            mag[lane + 64 * q] = (f2){__builtin_amdgcn_sqrtf(m2.x), __builtin_amdgcn_sqrtf(m2.y)};
        }
        wave_sync_lds();
        sum += mag[(lane * 3) & 255];
        wave_sync_lds();
    }
    out[blockIdx.x * 64 * W + threadIdx.x] = sum.x + sum.y;
}

template <int W>
__global__ __launch_bounds__(64 * W, W / 4) void k_single(const float* consts, float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kXBuf = (lds_padded(NC, FftCfg<10>::PMMAX) * 8 + 15) & ~15;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    cf* lds = reinterpret_cast<cf*>(smem + wv * kXBuf);
    cf* inbuf = reinterpret_cast<cf*>(smem + W * kXBuf + wv * NC * 8);
    cf tw[14], post[4], win[8];
#pragma unroll
    for (int i = 0; i < 14; ++i) tw[i] = mk(consts[(i * 2) * 64 + lane], consts[(i * 2 + 1) * 64 + lane]);
#pragma unroll
    for (int i = 0; i < 4; ++i) post[i] = mk(consts[(28 + 2 * i) * 64 + lane], consts[(29 + 2 * i) * 64 + lane]);
#pragma unroll
    for (int i = 0; i < 8; ++i) win[i] = mk(consts[(36 + 2 * i) * 64 + lane], consts[(37 + 2 * i) * 64 + lane]);
    for (int q = 0; q < 8; ++q) inbuf[lane + 64 * q] = mk(0.01f * lane, -0.01f * q);
    wave_sync_lds();
    float sum = 0.f;
    float* magbuf = reinterpret_cast<float*>(lds);
    for (int it = 0; it < iters; ++it) {
        cf x[1][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) x[0][q] = inbuf[lane + 64 * q] * win[q];
        cf* const l1[1] = {lds};
        float* const m1[1] = {magbuf};
        fft_frames<10, 1, true>(x, tw, l1, lane);
        untangle_mag<10, false, 1>(x, post, l1, m1, lane);
        wave_sync_lds();
        sum += magbuf[(lane * 3) & 255];
        wave_sync_lds();
    }
    out[blockIdx.x * 64 * W + threadIdx.x] = sum;
}

template <typename K>
static float time_kernel(K kern, int threads, size_t lds, const float* consts, float* out, int iters) {
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) kern<<<256, threads, lds>>>(consts, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) kern<<<256, threads, lds>>>(consts, out, iters);
    hipEventRecord(b);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms / 10;
}

int main() {
    std::vector<float> h(64 * 64);
    for (int i = 0; i < 64 * 64; ++i) h[i] = 0.3f + 0.6f * (float)((i * 2654435761u) % 1000) / 1000.f;  // O(1) factors
    float *consts, *out;
    hipMalloc(&consts, h.size() * 4);
    hipMemcpy(consts, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 200;
    constexpr int kXBuf = (lds_padded(NC, FftCfg<10>::PMMAX) * 8 + 15) & ~15;
    const float a16 = time_kernel(k_single<16>, 1024, 16 * (kXBuf + NC * 8), consts, out, iters);
    const float a12 = time_kernel(k_single<12>, 768, 12 * (kXBuf + NC * 8), consts, out, iters);
    const float b12 = time_kernel(k_pair<12>, 768, 12 * (kPairBuf + (NC / 2 + 8) * 8), consts, out, iters);
    const float b8 = time_kernel(k_pair<8>, 512, 8 * (kPairBuf + (NC / 2 + 8) * 8), consts, out, iters);
    auto rate = [&](float ms, int waves, int per) { return ms * 1e6 / ((double)iters * waves * per); };  // ns per frame per CU
    printf("A one frame per wave : 16 waves %.1f ns per frame and CU, 12 waves %.1f\n", rate(a16, 16, 1), rate(a12, 12, 1));
    printf("B two frames per wave: 12 waves %.1f ns per frame and CU,  8 waves %.1f\n", rate(b12, 12, 2), rate(b8, 8, 2));
    return 0;
}
