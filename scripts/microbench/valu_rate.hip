// VALU issue-rate microbenchmark for gfx950: cycles per wave-instruction for scalar vs packed fp32
// ops at 1..4 waves per SIMD (one workgroup per CU).  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x * 0.001f + i, 1.0f + i};
    const f2 c = {1.0001f, 0.9999f};
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) { a[i].x = a[i].x * c.x + c.y; a[i].y = a[i].y * c.y + c.x; }      // 2 v_fma_f32
                else if (MODE == 1) a[i] = __builtin_elementwise_fma(a[i], c, c);                     // 1 v_pk_fma_f32
                else if (MODE == 2) { a[i].x = a[i].x + c.x; a[i].y = a[i].y + c.y; }               // 2 v_add_f32
                else a[i] = a[i] + c;                                                                // 1 v_pk_add_f32
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 512 * 1024 * 4); hipMalloc(&cyc, 8);
    const int iters = 2000;
    const char* names[4] = {"2x v_fma_f32", "1x v_pk_fma_f32", "2x v_add_f32", "1x v_pk_add_f32"};
    for (int mode = 0; mode < 4; ++mode)
        for (int wps = 1; wps <= 8; ++wps) {
            if (mode == 0 || mode == 2) continue;  // hipcc SLP-packs the scalar forms into the packed ones anyway
            unsigned long long h = 0;
            const int blocks = wps <= 4 ? 256 : 512, threads = wps <= 4 ? 256 * wps : 128 * wps;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (mode == 1) k<1><<<blocks, threads>>>(out, cyc, iters);
                if (mode == 3) k<3><<<blocks, threads>>>(out, cyc, iters);
                hipEventRecord(e1);
                hipDeviceSynchronize();
                hipEventElapsedTime(&ms, e0, e1);
            }
            printf("[wall %.3f ms -> %.2f TFLOP-equivalent pk-ops/s x1e12: %.2f] ", ms, 0.0, (double)blocks * (threads / 64) * iters * 64 / (ms * 1e-3) / 1e12);
            hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
            const double groups = (double)iters * 64;   // complex-sized (2 float) operations per wave
            printf("%-16s waves/SIMD %d: %.2f cycles per 2-float op per wave, %.2f per SIMD\n", names[mode], wps,
                   h / groups, h / groups / wps);
        }
    return 0;
}
