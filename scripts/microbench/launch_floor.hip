// launch_floor.hip -- what an (almost) empty kernel costs on MI355X as a function of launch geometry:
// workgroups x threads, dynamic LDS, VGPR allocation.  Two clocks: the dispatch's own begin/end
// timestamps (hipExtLaunchKernel event pair, what bench.py's roofline uses) and the stream time per
// launch of a back-to-back train.   hipcc -O3 --offload-arch=gfx950 launch_floor.hip -o launch_floor
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int THREADS, int VG>
__global__ __launch_bounds__(THREADS) void k_empty(float* out, int n) {
    extern __shared__ float sm[];
    // touch VG registers so the allocation is real
    float v[VG];
#pragma unroll
    for (int i = 0; i < VG; ++i) v[i] = (float)(threadIdx.x + i);
#pragma unroll
    for (int i = 0; i < VG; ++i) asm volatile("" : "+v"(v[i]));
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += v[i];
    if (n < 0) { sm[threadIdx.x] = s; out[blockIdx.x * THREADS + threadIdx.x] = sm[threadIdx.x ^ 1]; }
}

template <int THREADS, int VG>
static void run(const char* name, int grid, size_t lds, float* d) {
    auto kern = k_empty<THREADS, VG>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipStream_t s;
    hipStreamCreate(&s);
    const int N = 200;
    std::vector<hipEvent_t> ev(2 * N);
    for (auto& e : ev) hipEventCreate(&e);
    int n = 0;
    void* args[] = {&d, &n};
    for (int i = 0; i < 20; ++i) kern<<<grid, THREADS, lds, s>>>(d, 0);
    hipStreamSynchronize(s);
    for (int i = 0; i < N; ++i)
        hipExtLaunchKernel((const void*)kern, dim3(grid), dim3(THREADS), args, lds, s, ev[2 * i], ev[2 * i + 1], 0);
    hipStreamSynchronize(s);
    std::vector<float> ms(N);
    for (int i = 0; i < N; ++i) hipEventElapsedTime(&ms[i], ev[2 * i], ev[2 * i + 1]);
    std::sort(ms.begin(), ms.end());
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int M = 2000;
    hipEventRecord(a, s);
    for (int i = 0; i < M; ++i) kern<<<grid, THREADS, lds, s>>>(d, 0);
    hipEventRecord(b, s);
    hipStreamSynchronize(s);
    float tot = 0;
    hipEventElapsedTime(&tot, a, b);
    printf("%-44s grid %5d x %4d thr, LDS %6zu B: dispatch begin->end median %.2f us (min %.2f), train %.2f us/launch\n", name,
           grid, THREADS, lds, ms[N / 2] * 1e3, ms[0] * 1e3, tot * 1e3 / M);
}

int main() {
    float* d;
    hipMalloc(&d, 64 << 20);
    run<1024, 100>("1 WG/CU, 16 waves, ~128 VGPR, 140 KB LDS", 256, 140 * 1024, d);
    run<1024, 100>("1 WG/CU, 16 waves, ~128 VGPR, no LDS", 256, 0, d);
    run<1024, 16>("1 WG/CU, 16 waves, few VGPR, 140 KB LDS", 256, 140 * 1024, d);
    run<768, 100>("1 WG/CU, 12 waves, ~128 VGPR, 110 KB LDS", 256, 110 * 1024, d);
    run<512, 100>("2 WG/CU, 8 waves, ~128 VGPR, 70 KB LDS", 512, 70 * 1024, d);
    run<256, 100>("4 WG/CU, 4 waves, ~128 VGPR, 35 KB LDS", 1024, 35 * 1024, d);
    run<256, 100>("1 WG/CU, 4 waves, ~128 VGPR, 35 KB LDS", 256, 35 * 1024, d);
    run<64, 100>("16 WG/CU, 1 wave, ~128 VGPR, 9 KB LDS", 4096, 9 * 1024, d);
    run<256, 16>("4 WG/CU, 4 waves, few VGPR, no LDS", 1024, 0, d);
    run<256, 16>("1 WG, 4 waves, few VGPR, no LDS", 1, 0, d);
    return 0;
}
