// Cross-lane move issue-rate microbenchmark for gfx950 (K1 closure): what does a butterfly
// exchange INSIDE the register file cost, per complex point, against the packed arithmetic it would sit beside?
//   mode 0: v_pk_fma_f32                       (the FFT's own arithmetic: 1 instruction per complex point)
//   mode 1: 2 x v_mov_b32_dpp row_ror:4        (moving one complex point to a partner lane of its 16-lane row)
//   mode 2: 2 x v_mov_b32_dpp + v_pk_fma_f32   (one radix-2 butterfly across lanes: t = dpp(x); x = t + s x)
//   mode 3: v_permlane32_swap_b32 x 2          (exchange between the wave's halves, two registers per instruction)
// Build: hipcc -O3 --offload-arch=gfx950 xlane_rate.hip -o xlane_rate       Run on an MI355X: prints cycles per complex point.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CTRL>
__device__ __forceinline__ float dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}

template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x * 0.001f + i, 1.0f + i};
    const f2 c = {1.0001f, 0.9999f};
    const f2 s = {(threadIdx.x & 4) ? -1.f : 1.f, (threadIdx.x & 4) ? -1.f : 1.f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) a[i] = __builtin_elementwise_fma(a[i], c, c);
                else if (MODE == 1) a[i] = f2{dpp<0x124>(a[i].x), dpp<0x124>(a[i].y)};   // row_ror:4
                else if (MODE == 2) {
                    const f2 t = {dpp<0x124>(a[i].x), dpp<0x124>(a[i].y)};
                    a[i] = __builtin_elementwise_fma(a[i], s, t);
                } else {
                    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i].x), "+v"(a[(i + 1) & 7].x));
                    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i].y), "+v"(a[(i + 1) & 7].y));
                }
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0;
    for (int i = 0; i < 8; ++i) sum += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int MODE>
static void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000;
    for (int wps = 1; wps <= 4; wps *= 2) {
        unsigned long long h = 0;
        for (int rep = 0; rep < 2; ++rep) {
            k<MODE><<<256, 256 * wps>>>(out, cyc, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
        const double points = (double)iters * 64;  // complex points handled per wave
        printf("%-44s waves/SIMD %d: %6.2f cycles per complex point per wave, %5.2f per SIMD\n", name, wps, h / points,
               h / points / wps);
    }
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 8);
    run<0>("v_pk_fma_f32 (1 per point)", out, cyc);
    run<1>("2 x v_mov_b32_dpp (move a point)", out, cyc);
    run<2>("2 x v_mov_b32_dpp + v_pk_fma_f32 (butterfly)", out, cyc);
    run<3>("2 x v_permlane32_swap_b32 (2 points)", out, cyc);
    return 0;
}
