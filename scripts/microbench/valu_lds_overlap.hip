// Do the vector ALU and the LDS data path of a CU overlap?  (K1's frame loop keeps the vector ALU ~46 % and the LDS ~52 % busy -
// together ~100 % - and got no faster when 10 % of its LDS cycles were removed.)  One 16-wave workgroup per CU; waves 0-7 are
// "V" waves (a chain-free stream of packed FMAs), waves 8-15 are "L" waves (conflict-free ds_write_b64 / ds_read_b64 through a
// private region, the exchange pattern of the FFT): waves w and w + 4 share a SIMD, so every SIMD holds 2 V and 2 L waves.
// Timed: V alone, L alone, both.  both ~ max(V, L): the pipes overlap; both ~ V + L: they exclude each other (register-file
// ports of the SIMD feeding either the ALU or the LDS data path).
// Build: hipcc -O3 --offload-arch=gfx950 valu_lds_overlap.hip -o valu_lds_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool V, bool L, int LMODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters) {
    __shared__ f2 lds[16][64 * 9];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{(float)threadIdx.x * 0.001f + i, 1.0f + i};
    const f2 c = {1.0001f, 0.9999f};
    if (wv < 8) {
        if (V) {
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int r = 0; r < 6; ++r)
#pragma unroll
                    for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], c, c);  // 48 v_pk_fma_f32
            }
        }
    } else if (L) {
        volatile f2* p = lds[wv] + lane;
        for (int it = 0; it < iters; ++it) {
            if (LMODE == 0 || LMODE == 2) {
#pragma unroll
                for (int i = 0; i < 8; ++i) p[64 * i] = a[i];   // 8 ds_write_b64 (unit stride: conflict-free)
            }
            if (LMODE == 0 || LMODE == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = p[64 * i];   // 8 ds_read_b64
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool V, bool L, int LMODE>
static float run(float* out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0, best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        k<V, L, LMODE><<<256, 1024>>>(out, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 4000;
    const char* names[3] = {"8 writes + 8 reads", "8 reads", "8 writes"};
    const float v = run<true, false, 0>(out, iters);
    printf("V alone (2 waves per SIMD, 48 v_pk_fma_f32 per iteration): %.3f ms = %.2f cycles per instruction and SIMD at 2.1 GHz\n", v,
           v * 1e-3 * 2.1e9 / ((double)iters * 48 * 2));
    for (int m = 0; m < 3; ++m) {
        float l, b;
        if (m == 0) { l = run<false, true, 0>(out, iters); b = run<true, true, 0>(out, iters); }
        else if (m == 1) { l = run<false, true, 1>(out, iters); b = run<true, true, 1>(out, iters); }
        else { l = run<false, true, 2>(out, iters); b = run<true, true, 2>(out, iters); }
        printf("L = %-18s: L alone %.3f ms, V + L together %.3f ms  (max %.3f, sum %.3f) -> overlap %.0f %%\n", names[m], l, b,
               v > l ? v : l, v + l, 100.0 * (v + l - b) / (v < l ? v : l));
    }
    return 0;
}
