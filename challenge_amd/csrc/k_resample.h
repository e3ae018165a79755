// k_resample.h -- band-limited resampling of a waveform to another sample rate: the step in front of normalize + STFT in the
// reference's load_wav (data_utils.py:20-21: torchaudio.compliance.kaldi.resample_waveform(wav, r, 16000)).
// Part of the single translation unit iris_frontend.hip.
#pragma once
// ---------------------------------------------------------------------------
// torchaudio (unpinned, requirements.txt:5; source not under /root/reference) implements resample_waveform as
// torchaudio.functional.resample(waveform, orig, new, lowpass_filter_width = 6) with rolloff 0.99 and the Hann-windowed sinc
// ("sinc_interp_hann"): with o = orig / gcd, n = new / gcd, f = 0.99 min(o, n), w = ceil(6 o / f), K = 2 w + o
//     tap[j][k] = (f / o) sinc(pi t) cos^2(pi t / 12),   t = clamp(f ((k - w) / o - j / n), -6, 6),   j < n, k < K
//     y[i n + j] = sum_k xpad[i o + k] tap[j][k],        xpad = x with w zeros in front and w + o behind,
//     i <= len / o, the result cut to ceil(n len / o) samples                         (a strided conv1d with n output channels)
// The taps are computed HERE in fp64 and rounded once to fp32 (torchaudio evaluates them in the waveform's own dtype: its fp32
// taps carry ~1e-7 of rounding that these do not); the sums are fp32, one output sample per thread: neighbouring threads read
// neighbouring taps of the transposed table [k][n] and the SAME input sample (a broadcast).  Offline / evaluation path
// (load_wav): 10 s of 44.1 kHz stereo -> 16 kHz is 320,000 outputs x 475 taps = 0.3 GFLOP; not a hot kernel, written plainly.
// ---------------------------------------------------------------------------
#include <map>
#include <mutex>
#include <numeric>
#include <tuple>
#include <vector>

__global__ __launch_bounds__(256) void k_resample(const float* __restrict__ x, float* __restrict__ y, const float* __restrict__ taps_t,
                                                  int channels, long long len, long long out_len, int o, int n, int w, int K) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)channels * out_len) return;
    const int c = (int)(idx / out_len);
    const long long t = idx - (long long)c * out_len;
    const long long i = t / n;
    const int j = (int)(t - i * n);
    const float* const xc = x + (long long)c * len;
    const long long first = i * o - w;                       // input index of tap 0
    const int k_lo = first < 0 ? (int)(-first) : 0;          // taps that fall on the zero padding contribute nothing
    const long long room = len - first;
    const int k_hi = room < (long long)K ? (room > 0 ? (int)room : 0) : K;
    float acc = 0.f;
    for (int k = k_lo; k < k_hi; ++k) acc = fmaf(xc[first + k], taps_t[(size_t)k * n + j], acc);
    y[idx] = acc;
}

struct ResampleTaps {
    float* dev = nullptr;
    int o = 0, n = 0, w = 0, K = 0;
};

// output samples for `len` input samples: ceil(new len / orig) on the reduced ratio; 0 for invalid arguments
extern "C" long long iris_resample_len(long long len, int orig_freq, int new_freq) {
    if (len < 0 || orig_freq <= 0 || new_freq <= 0) return 0;
    const int g = std::gcd(orig_freq, new_freq), o = orig_freq / g, n = new_freq / g;
    return ((long long)n * len + o - 1) / o;
}

// wav: DEVICE [channels][len] fp32 at orig_freq; out: DEVICE [channels][iris_resample_len(len, orig_freq, new_freq)] fp32.
// orig_freq == new_freq copies.  The tap table of a rate pair is built once per device and kept (a few hundred KB).
extern "C" int iris_resample(const float* wav, int channels, long long len, int orig_freq, int new_freq, float* out, void* stream) {
    if (!wav || !out) return fail(IRIS_E_INVALID, "iris_resample: NULL argument");
    if (channels <= 0 || len <= 0) return fail(IRIS_E_INVALID, "iris_resample: empty waveform");
    if (orig_freq <= 0 || new_freq <= 0) return fail(IRIS_E_INVALID, "iris_resample: sample rates %d -> %d", orig_freq, new_freq);
    const hipStream_t st = (hipStream_t)stream;
    if (orig_freq == new_freq) {
        HIP_TRY(hipMemcpyAsync(out, wav, (size_t)channels * (size_t)len * sizeof(float), hipMemcpyDeviceToDevice, st));
        return IRIS_OK;
    }
    const int g = std::gcd(orig_freq, new_freq), o = orig_freq / g, n = new_freq / g;
    const double base = 0.99 * (double)std::min(o, n);
    const int w = (int)std::ceil(6.0 * o / base), K = 2 * w + o;
    if ((long long)n * K > (1LL << 26)) return fail(IRIS_E_UNSUPPORTED, "iris_resample: %d -> %d needs a table of %lld taps", orig_freq, new_freq, (long long)n * K);
    const long long out_len = iris_resample_len(len, orig_freq, new_freq);
    if ((long long)channels * out_len >= (1LL << 40)) return fail(IRIS_E_UNSUPPORTED, "iris_resample: output too large");
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    static std::mutex mu;
    static std::map<std::tuple<int, int, int>, ResampleTaps> cache;
    ResampleTaps tp;
    {
        std::lock_guard<std::mutex> lock(mu);
        auto it = cache.find({dev, o, n});
        if (it == cache.end()) {
            std::vector<float> host((size_t)K * n);
            const double pi = 3.14159265358979323846, scale = base / o;
            for (int j = 0; j < n; ++j)
                for (int k = 0; k < K; ++k) {
                    double t = ((double)(k - w) / o - (double)j / n) * base;
                    t = std::min(6.0, std::max(-6.0, t));
                    const double c = std::cos(t * pi / 6.0 / 2.0), window = c * c;
                    const double a = t * pi, s = a == 0.0 ? 1.0 : std::sin(a) / a;
                    host[(size_t)k * n + j] = (float)(s * window * scale);
                }
            ResampleTaps fresh;
            fresh.o = o, fresh.n = n, fresh.w = w, fresh.K = K;
            HIP_TRY(hipMalloc(&fresh.dev, host.size() * sizeof(float)));
            // (a blocking copy on purpose: `host` dies with this scope, and the table is built once per rate pair)
            hipError_t e = hipMemcpy(fresh.dev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(fresh.dev);
                HIP_TRY(e);
            }
            it = cache.emplace(std::make_tuple(dev, o, n), fresh).first;
        }
        tp = it->second;
    }
    const long long total = (long long)channels * out_len;
    k_resample<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(wav, out, tp.dev, channels, len, out_len, tp.o, tp.n, tp.w, tp.K);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
