// Standalone build of the Winograd convolution kernel (challenge_amd/csrc/k_conv_wino.h) for the round-5 microbenchmark:
//   hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -shared -o scripts/microbench/libwino.so scripts/microbench/wino_conv.hip
// driven by scripts/gpu_wino_bench.py (ctypes) against torch / MIOpen on the same shapes.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <type_traits>

enum { IRIS_OK = 0, IRIS_E_INVALID = -1, IRIS_E_UNSUPPORTED = -2 };
enum { IRIS_WINO_POOL = 1, IRIS_WINO_OUT_NHWC = 2, IRIS_WINO_IN_NHWC = 4, IRIS_WINO_RELU = 8 };
#include <cmath>
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* wino_last_error(void) { return g_err; }
#define HIP_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) return fail((int)e_, "%s failed: %s", #expr, hipGetErrorString(e_));        \
    } while (0)
#define IRIS_WINO_STANDALONE 1
typedef struct iris_pack_job {
    const float* weight;
    float* packed;
    long stride_o, stride_i, stride_h, stride_w;
    int cin, cout, transposed, first_block;
} iris_pack_job;
#include "../../challenge_amd/csrc/k_conv_wino.h"
#include "../../challenge_amd/csrc/k_conv_wino_wrw.h"
#include "../../challenge_amd/csrc/k_conv_wino_b3.h"
