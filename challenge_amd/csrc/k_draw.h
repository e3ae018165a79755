// k_draw.h -- the random half of a training batch drawn ON THE DEVICE: which sources merge_complex_specs mixes and
// with which offsets / gains (pipeline.py:29-106, dataset graph :147-174), and the SpecAugment bands of `augment`
// (data_utils.py:58-61, transforms.py:25-26).  Part of the single translation unit iris_frontend.hip.
//
// The host used to draw these with NumPy and upload a 48-byte record per source use (0.38-0.41 ms per batch of 64
// against ~0.1 ms of GPU time).  Here one small kernel writes the same table from a counter-based generator:
//   * Philox4x32-10, key = the caller's 64-bit seed, counter = (call counter lo, hi, sample * 64 + column, purpose):
//     every draw is a pure function of (seed, call counter, sample, column) - reproducible, independent;
//   * the call counter and the stream positions live in DEVICE memory (`state`) and are advanced by the kernel itself:
//     a captured launch replays with fresh draws and no host input;
//   * integers come from one 32-bit word as (word * range) >> 32 (bias < range / 2^32, irrelevant at these ranges),
//     reals in [0, 1) from the top 24 bits;
//   * "the next max_voices voices of a shuffled, repeated stream" (pipeline.py:147-160) = position p of the stream:
//     epoch p / n, element perm_epoch(p % n), perm a keyed bijection of [0, 2^k) (add / odd multiply / xor-shift
//     rounds, keys from Philox) walked until it lands below n - every source exactly once per epoch.
// The table has a FIXED stride of 1 + max_voices + max_noises records per sample (first[b] = b * stride); unused
// voice / noise slots carry kind = -1 and are skipped by the mix kernels.  oracle/frontend_ref.py restates the
// generator and the integer draws in NumPy (tests compare them bit for bit).
#pragma once

struct Philox4 {
    uint32_t x, y, z, w;
};

__host__ __device__ inline Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}

// purposes (counter word 3)
enum { kDrawSample = 1, kDrawVoice = 2, kDrawNoise = 3, kDrawPerm = 4, kDrawBandT = 5, kDrawBandF = 6 };

__host__ __device__ inline uint32_t draw_below(uint32_t word, uint32_t range) {  // uniform on [0, range), range >= 1
    return (uint32_t)(((uint64_t)word * range) >> 32);
}
__host__ __device__ inline float draw_unit(uint32_t word) { return (float)(word >> 8) * (1.0f / 16777216.0f); }  // [0, 1)

// element i (< n) of epoch `epoch` of stream `stream`: a keyed bijection of [0, n)
__host__ __device__ inline uint32_t stream_perm(uint32_t i, uint32_t n, uint32_t stream, uint64_t epoch, uint32_t k0, uint32_t k1) {
    if (n <= 1) return 0;
    int bits = 1;
    while ((1u << bits) < n) ++bits;
    const uint32_t mask = bits >= 32 ? 0xffffffffu : (1u << bits) - 1;
    const int sh = bits / 2 > 0 ? bits / 2 : 1;
    const Philox4 h = philox4x32_10((uint32_t)epoch, (uint32_t)(epoch >> 32), stream, kDrawPerm, k0, k1);
    uint32_t v = i;
    do {  // every line is a bijection of [0, 2^bits); values >= n are walked on (cycle walking keeps it a bijection)
        v = (v + h.x) & mask;
        v = (v * (h.y | 1u)) & mask;
        v ^= v >> sh;
        v = (v + h.z) & mask;
        v = (v * (h.w | 1u)) & mask;
        v ^= v >> sh;
        v = (v * 0x9E3779B1u) & mask;
        v ^= v >> sh;
    } while (v >= n);
    return v;
}

struct MixCorpusDev {  // device-resident description of one corpus (backgrounds, voices or noises)
    const uint64_t* src;     // [n] device addresses of the sources
    const uint64_t* active;  // [n] device addresses of the per-frame activity flags (voices), or nullptr
    const int32_t* T;        // [n] frames per source
    const int32_t* len;      // [n] samples per channel (waveform corpora), or nullptr
    int32_t n;
};

struct MixDrawArgs {
    MixCorpusDev bg, voice, noise;  // noise.n == 0: no noise corpus
    int batch, n_frame, V, Nn;
    float min_ratio, min_noise_ratio, snr;
    uint32_t k0, k1;
    unsigned long long* state;  // [4] call counter, positions of the background / voice / noise streams
    iris_mix_src* table;        // [batch * (1 + V + Nn)]
    int32_t* first;             // [batch + 1]
};

__device__ __forceinline__ int padded_pad(int frames, float ratio, int n_frame) {  // pipeline.py:59-66: fp32 product, truncated
    return n_frame - (int)(ratio * (float)frames);
}

__global__ __launch_bounds__(256) void k_mix_draw(const MixDrawArgs a) {
    const unsigned long long ctr = a.state[0], pos_b = a.state[1], pos_v = a.state[2], pos_n = a.state[3];
    const uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32);
    const int cols = 1 + a.V + a.Nn;
    __syncthreads();  // every thread has read the state before thread 0 advances it
    for (int b = threadIdx.x; b < a.batch; b += blockDim.x) {
        iris_mix_src* row = a.table + (size_t)b * cols;
        auto pick = [&](const MixCorpusDev& c, uint32_t stream, unsigned long long pos) {
            return stream_perm((uint32_t)(pos % (unsigned long long)c.n), (uint32_t)c.n, stream, pos / (unsigned long long)c.n, a.k0, a.k1);
        };
        const Philox4 rs = philox4x32_10(c0, c1, (uint32_t)b * 64u, kDrawSample, a.k0, a.k1);
        // background: one per sample, random crop of the tiled spectrogram (pipeline.py:29-35)
        {
            const uint32_t s = pick(a.bg, 0, pos_b + b);
            const int T = a.bg.T[s];
            const int reps = (a.n_frame + T - 1) / T;
            iris_mix_src r = {};
            r.src = reinterpret_cast<const float*>(a.bg.src[s]);
            r.active = nullptr;
            r.T = T;
            r.pad = 0;
            r.off = (int)draw_below(rs.z, (uint32_t)(reps * T - a.n_frame + 1));
            r.gain = 1.0f;
            r.kind = 0;
            r.slot = 0;
            r.label_row = 0;
            r.reserved = a.bg.len ? a.bg.len[s] : 0;
            row[0] = r;
        }
        // voices: the group is padded to its longest member (padded_batch); n_voices ~ U{1..V-1} (pipeline.py:41-46)
        const int n_voices = a.V > 1 ? 1 + (int)draw_below(rs.x, (uint32_t)(a.V - 1)) : 1;
        int v_len = 0;
        for (int j = 0; j < a.V; ++j) v_len = max(v_len, a.voice.T[pick(a.voice, 1, pos_v + (unsigned long long)b * a.V + j)]);
        {
            const int pad = padded_pad(v_len, a.min_ratio, a.n_frame);
            const int length = pad > 0 ? v_len + 2 * pad : v_len, maxval = length - a.n_frame;
            for (int j = 0; j < a.V; ++j) {
                const uint32_t s = pick(a.voice, 1, pos_v + (unsigned long long)b * a.V + j);
                const Philox4 rv = philox4x32_10(c0, c1, (uint32_t)b * 64u + 1u + (uint32_t)j, kDrawVoice, a.k0, a.k1);
                iris_mix_src r = {};
                r.src = reinterpret_cast<const float*>(a.voice.src[s]);
                r.active = reinterpret_cast<const float*>(a.voice.active[s]);
                r.T = a.voice.T[s];
                r.pad = max(pad, 0);
                r.off = maxval > 0 ? (int)draw_below(rv.y, (uint32_t)maxval) : 0;                 // pipeline.py:68-69
                r.gain = exp10f(-(draw_unit(rv.x) * (-a.snr / 10.0f)));                            // pipeline.py:50
                r.kind = j < n_voices ? 1 : -1;
                r.slot = j;
                r.label_row = (int)s;
                r.reserved = a.voice.len ? a.voice.len[s] : 0;
                row[1 + j] = r;
            }
        }
        // noises: n_noises ~ U{0..Nn-1}, gain 10^-U[0,2), crop of the padded noise (pipeline.py:86-106)
        if (a.Nn > 0) {
            const int n_noises = (int)draw_below(rs.y, (uint32_t)a.Nn);
            int n_len = 0;
            for (int j = 0; j < a.Nn; ++j) n_len = max(n_len, a.noise.T[pick(a.noise, 2, pos_n + (unsigned long long)b * a.Nn + j)]);
            const int pad = padded_pad(n_len, a.min_noise_ratio, a.n_frame);
            const int length = pad > 0 ? n_len + 2 * pad : n_len;
            for (int j = 0; j < a.Nn; ++j) {
                const uint32_t s = pick(a.noise, 2, pos_n + (unsigned long long)b * a.Nn + j);
                const Philox4 rn = philox4x32_10(c0, c1, (uint32_t)b * 64u + 1u + (uint32_t)j, kDrawNoise, a.k0, a.k1);
                iris_mix_src r = {};
                r.src = reinterpret_cast<const float*>(a.noise.src[s]);
                r.active = nullptr;
                r.T = a.noise.T[s];
                r.pad = max(pad, 0);
                r.off = (int)draw_below(rn.y, (uint32_t)(max(length - a.n_frame, 0) + 1));       // pipeline.py:103
                r.gain = exp10f(-(draw_unit(rn.x) * 2.0f));                                         // pipeline.py:94
                r.kind = j < n_noises ? 2 : -1;
                r.slot = 0;
                r.label_row = 0;
                r.reserved = a.noise.len ? a.noise.len[s] : 0;
                row[1 + a.V + j] = r;
            }
        }
    }
    for (int b = threadIdx.x; b <= a.batch; b += blockDim.x) a.first[b] = b * cols;
    __syncthreads();
    if (threadIdx.x == 0) {
        a.state[0] = ctr + 1;
        a.state[1] = pos_b + (unsigned long long)a.batch;
        a.state[2] = pos_v + (unsigned long long)a.batch * a.V;
        a.state[3] = pos_n + (unsigned long long)a.batch * a.Nn;
    }
}

// SpecAugment bands of `augment` for a batch: n_t time masks (size ~ U{0..max_t-1}) and n_f frequency masks per sample,
// offset ~ U{0..total-size-1} (transforms.py:25-26).  state[0] = call counter (advanced here).
__global__ __launch_bounds__(256) void k_augment_draw(int batch, int n_time, int n_t, int max_t, int n_freq, int n_f, int max_f,
                                                      uint32_t k0, uint32_t k1, unsigned long long* state, int32_t* t_bands,
                                                      int32_t* f_bands) {
    const unsigned long long ctr = state[0];
    const uint32_t c0 = (uint32_t)ctr, c1 = (uint32_t)(ctr >> 32);
    __syncthreads();
    const int per = n_t + n_f;
    for (int i = threadIdx.x; i < batch * per; i += blockDim.x) {
        const int b = i / per, j = i - b * per;
        const bool is_t = j < n_t;
        const int total = is_t ? n_time : n_freq, mx = is_t ? max_t : max_f, jj = is_t ? j : j - n_t;
        const Philox4 r = philox4x32_10(c0, c1, (uint32_t)b * 64u + (uint32_t)jj, is_t ? kDrawBandT : kDrawBandF, k0, k1);
        const int size = (int)draw_below(r.x, (uint32_t)mx);
        const int off = (int)draw_below(r.y, (uint32_t)(total - size));
        int32_t* o = (is_t ? t_bands + ((size_t)b * n_t + jj) * 2 : f_bands + ((size_t)b * n_f + jj) * 2);
        o[0] = off;
        o[1] = size;
    }
    __syncthreads();
    if (threadIdx.x == 0) state[0] = ctr + 1;
}

extern "C" int iris_mix_draw(const iris_mix_corpus* backgrounds, const iris_mix_corpus* voices, const iris_mix_corpus* noises,
                             int batch, int n_frame, int max_voices, int max_noises, float min_ratio, float min_noise_ratio,
                             float snr, uint64_t seed, uint64_t* state_dev, iris_mix_src* table_out, int32_t* first_out,
                             void* stream) {
    if (!backgrounds || !voices || !state_dev || !table_out || !first_out)
        return fail(IRIS_E_INVALID, "iris_mix_draw: NULL argument");
    if (batch <= 0 || n_frame <= 0 || max_voices <= 0 || max_noises < 0)
        return fail(IRIS_E_INVALID, "iris_mix_draw: bad sizes (batch %d, %d frames, %d voices, %d noises)", batch, n_frame,
                    max_voices, max_noises);
    if (1 + max_voices + max_noises > 64)
        return fail(IRIS_E_UNSUPPORTED, "iris_mix_draw: at most 63 voices + noises per sample");
    if (backgrounds->n <= 0 || voices->n <= 0 || !backgrounds->src || !backgrounds->T || !voices->src || !voices->T ||
        !voices->active)
        return fail(IRIS_E_INVALID, "iris_mix_draw: empty or incomplete background / voice corpus");
    const bool has_noise = noises && noises->n > 0 && max_noises > 0;
    if (has_noise && (!noises->src || !noises->T)) return fail(IRIS_E_INVALID, "iris_mix_draw: incomplete noise corpus");
    MixDrawArgs a;
    auto conv = [](const iris_mix_corpus* c) {
        MixCorpusDev d;
        d.src = reinterpret_cast<const uint64_t*>(c->src);
        d.active = reinterpret_cast<const uint64_t*>(c->active);
        d.T = c->T;
        d.len = c->len;
        d.n = c->n;
        return d;
    };
    a.bg = conv(backgrounds);
    a.voice = conv(voices);
    if (has_noise) a.noise = conv(noises);
    else a.noise = MixCorpusDev{nullptr, nullptr, nullptr, nullptr, 0};
    a.batch = batch;
    a.n_frame = n_frame;
    a.V = max_voices;
    a.Nn = has_noise ? max_noises : 0;
    a.min_ratio = min_ratio;
    a.min_noise_ratio = min_noise_ratio;
    a.snr = snr;
    a.k0 = (uint32_t)seed;
    a.k1 = (uint32_t)(seed >> 32);
    a.state = reinterpret_cast<unsigned long long*>(state_dev);
    a.table = table_out;
    a.first = first_out;
    k_mix_draw<<<1, 256, 0, (hipStream_t)stream>>>(a);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}

extern "C" int iris_augment_draw(int batch, int n_time, int n_time_masks, int max_time_mask, int n_freq, int n_freq_masks,
                                 int max_freq_mask, uint64_t seed, uint64_t* state_dev, int32_t* t_bands_out,
                                 int32_t* f_bands_out, void* stream) {
    if (!state_dev || (n_time_masks > 0 && !t_bands_out) || (n_freq_masks > 0 && !f_bands_out))
        return fail(IRIS_E_INVALID, "iris_augment_draw: NULL argument");
    if (batch <= 0 || n_time_masks < 0 || n_freq_masks < 0 || n_time_masks + n_freq_masks <= 0 || n_time_masks > 64 ||
        n_freq_masks > 64)
        return fail(IRIS_E_INVALID, "iris_augment_draw: bad sizes (batch %d, %d + %d masks)", batch, n_time_masks, n_freq_masks);
    // transforms.py:25-26: offset ~ U{0..total - size - 1} needs total - size >= 1 for every size < max_mask_size
    if ((n_time_masks > 0 && (max_time_mask <= 0 || max_time_mask > n_time)) ||
        (n_freq_masks > 0 && (max_freq_mask <= 0 || max_freq_mask > n_freq)))
        return fail(IRIS_E_INVALID, "iris_augment_draw: mask sizes must satisfy 0 < max_mask_size <= axis length");
    k_augment_draw<<<1, 256, 0, (hipStream_t)stream>>>(batch, n_time, n_time_masks, max_time_mask, n_freq, n_freq_masks,
                                                       max_freq_mask, (uint32_t)seed, (uint32_t)(seed >> 32),
                                                       reinterpret_cast<unsigned long long*>(state_dev), t_bands_out, f_bands_out);
    HIP_TRY(hipGetLastError());
    return IRIS_OK;
}
