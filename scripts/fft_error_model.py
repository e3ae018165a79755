"""CPU model of the rounding behaviour of the HIP frame transform (iris_fft.h + spectrum.h), operation by operation in
fp32 (FMAs emulated through fp64), beside NumPy's fp32 pocketfft and an fp64 reference.  Used to find out which steps
of the wave-per-frame transform cost accuracy at small n_fft without spending GPU time:
every variant below was first costed here, then built and measured on the GPU (profiles/r4/hip_vs_fp64_sweep.log).

    python scripts/fft_error_model.py [n_fft] [frames]
"""
import sys
import numpy as np

f32 = np.float32


def fma(a, b, c):
    """fp32 fused multiply-add: the product of two fp32 is exact in fp64; one extra rounding in the fp64 add is far
    below the fp32 rounding that follows."""
    return (a.astype(np.float64) * np.asarray(b, np.float64) + np.asarray(c, np.float64)).astype(f32)


class C:
    """complex array as separate fp32 re / im (a packed-f32 register pair per point)."""

    def __init__(self, re, im):
        self.re, self.im = np.asarray(re, f32), np.asarray(im, f32)

    def __add__(self, o):
        return C(self.re + o.re, self.im + o.im)

    def __sub__(self, o):
        return C(self.re - o.re, self.im - o.im)


def add_mi(a, b):  # a + (-i) b = a + (b.im, -b.re): one packed FMA with a (+1, -1) constant
    return C(a.re + b.im, a.im - b.re)


def sub_mi(a, b):
    return C(a.re - b.im, a.im + b.re)


def cmul_tw(a, w, mode="hip"):
    """a * w.  hip: t = (a.im * -w.im, a.im * w.re) rounded, r = fma(a.re, w, t) (v_pk_mul + v_pk_fma)."""
    if mode == "hip":
        t_re, t_im = (a.im * (-w.im)).astype(f32), (a.im * w.re).astype(f32)
        return C(fma(a.re, w.re, t_re), fma(a.re, w.im, t_im))
    if mode == "plain":  # 4 multiplies + 2 adds, all rounded (a compiler without contraction)
        return C((a.re * w.re).astype(f32) - (a.im * w.im).astype(f32), (a.re * w.im).astype(f32) + (a.im * w.re).astype(f32))
    if mode == "exact":  # product rounded once (what a compensated form could reach)
        re = a.re.astype(np.float64) * w.re - a.im.astype(np.float64) * w.im
        im = a.re.astype(np.float64) * w.im + a.im.astype(np.float64) * w.re
        return C(re.astype(f32), im.astype(f32))
    raise ValueError(mode)


def dft(v, mode):
    r = len(v)
    if r == 2:
        return [v[0] + v[1], v[0] - v[1]]
    if r == 4:
        t0, t1, t2, t3 = v[0] + v[2], v[0] - v[2], v[1] + v[3], v[1] - v[3]
        return [t0 + t2, add_mi(t1, t3), t0 - t2, sub_mi(t1, t3)]
    if r == 8:
        rr = f32(0.70710678118654752)
        a = [v[i] + v[i + 4] for i in range(4)]
        d = [v[i] - v[i + 4] for i in range(4)]
        s0, s1, t0, t1 = a[0] + a[2], a[1] + a[3], a[0] - a[2], a[1] - a[3]
        out = [None] * 8
        out[0], out[4], out[2], out[6] = s0 + s1, s0 - s1, add_mi(t0, t1), sub_mi(t0, t1)
        e1 = add_mi(d[1], d[1])
        e1 = C(e1.re * rr, e1.im * rr)
        e3 = sub_mi(d[3], d[3])
        e3 = C(e3.re * -rr, e3.im * -rr)
        s0, t0 = add_mi(d[0], d[2]), sub_mi(d[0], d[2])
        s1, t1 = e1 + e3, e1 - e3
        out[1], out[5], out[3], out[7] = s0 + s1, s0 - s1, add_mi(t0, t1), sub_mi(t0, t1)
        return out
    e, o = dft(v[0::2], mode), dft(v[1::2], mode)  # radix 16: two radix 8 + a twiddled radix-2 level
    out = [None] * r
    for k in range(r // 2):
        ang = -2.0 * np.pi * k / r
        w = C(np.full_like(o[k].re, f32(np.cos(ang))), np.full_like(o[k].re, f32(np.sin(ang))))
        t = o[k] if k == 0 else cmul_tw(o[k], w, "hip" if mode != "plain" else "plain")
        out[k], out[k + r // 2] = e[k] + t, e[k] - t
    return out


RADICES = {2048: [16, 16, 4], 1024: [8, 8, 8], 512: [4, 4, 4, 4], 256: [2] * 7}


def hip_fft(z, radices, mode="hip"):
    """Stockham autosort complex FFT over the last axis, the HIP kernel's stage order and roundings."""
    nc = z.re.shape[-1]
    ns = 1
    for r in radices:
        m = nc // r
        b = np.arange(m)
        v = [C(z.re[..., b + t * m], z.im[..., b + t * m]) for t in range(r)]
        if ns > 1:
            for t in range(1, r):
                ang = -2.0 * np.pi * ((b % ns) * t) / (ns * r)
                w = C(np.cos(ang).astype(f32), np.sin(ang).astype(f32))
                v[t] = cmul_tw(v[t], w, mode)
        v = dft(v, mode)
        out_re, out_im = np.empty_like(z.re), np.empty_like(z.im)
        base = (b // ns) * (ns * r) + (b % ns)
        for t in range(r):
            out_re[..., base + t * ns], out_im[..., base + t * ns] = v[t].re, v[t].im
        z = C(out_re, out_im)
        ns *= r
    return z


def hip_rfft_mag(frames, n_fft, mode="hip", untangle="hip"):
    """frames[..., n_fft] fp32 (already windowed or not, see caller) -> 2|X[k]|, k = 0..n_fft/2 like untangle_mag."""
    nc = n_fft // 2
    z = hip_fft(C(frames[..., 0::2], frames[..., 1::2]), RADICES[n_fft], mode)
    k = np.arange(nc // 2 + 1)  # lo half, and its mirror gives the hi half
    zk = C(z.re[..., k], z.im[..., k])
    pk = (nc - k) % nc
    zp = C(z.re[..., pk], z.im[..., pk])
    ang = -2.0 * np.pi * k / n_fft
    w = C(np.cos(ang).astype(f32), np.sin(ang).astype(f32))
    e = C(zk.re + zp.re, zk.im - zp.im)  # 2 E
    d = C(zk.re - zp.re, zk.im + zp.im)  # 2 i O
    if untangle == "hip":  # (-i d) * w: t = (d.re * w.im, -d.re * w.re); r = fma(d.im, w, t)
        t_re, t_im = (d.re * w.im).astype(f32), (-(d.re * w.re)).astype(f32)
        wo = C(fma(d.im, w.re, t_re), fma(d.im, w.im, t_im))
        lo, hi = e + wo, e - wo
    elif untangle == "fused":  # lo = e + (-i d) w with the last add inside the FMA chain: one rounding fewer
        t_re, t_im = fma(d.re, w.im, e.re), fma(-d.re, w.re, e.im)
        lo = C(fma(d.im, w.re, t_re), fma(d.im, w.im, t_im))
        t_re, t_im = fma(-d.re, w.im, e.re), fma(d.re, w.re, e.im)
        hi = C(fma(-d.im, w.re, t_re), fma(-d.im, w.im, t_im))
    else:
        raise ValueError(untangle)
    mag_lo = np.sqrt(fma(lo.re, lo.re, (lo.im * lo.im).astype(f32)))
    mag_hi = np.sqrt(fma(hi.re, hi.re, (hi.im * hi.im).astype(f32)))
    mag = np.empty(frames.shape[:-1] + (nc + 1,), f32)
    mag[..., k] = mag_lo
    mag[..., nc - k] = mag_hi  # k = nc/2 written twice with the same value up to the sign of im
    return mag


def main():
    n_fft = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((n_frames, n_fft)) * 0.1).astype(f32)
    k = np.arange(n_fft)
    win64 = 0.5 - 0.5 * np.cos(2 * np.pi * k / n_fft)
    win = win64.astype(f32)
    ref = 2.0 * np.abs(np.fft.rfft(x.astype(np.float64) * win.astype(np.float64), axis=-1))
    rms = np.sqrt((ref ** 2).mean())
    u = 2.0 ** -24

    def report(name, mag):
        d = mag.astype(np.float64) - ref
        small = ref < 0.02 * rms
        print(f"{name:34s} rms err {np.sqrt((d ** 2).mean()) / rms / u:6.3f} u  max {np.abs(d).max() / rms / u:6.2f} u  "
              f"rms err on the {small.sum()} bins below 2 % of rms: {np.sqrt((d[small] ** 2).mean()) / rms / u:6.3f} u")

    xw = (x * win).astype(f32)
    report("numpy fp32 pocketfft", 2.0 * np.abs(np.fft.rfft(xw, axis=-1)).astype(f32))
    import torch
    t = torch.stft(torch.from_numpy(x), n_fft, n_fft, n_fft, window=torch.from_numpy(win), center=False, return_complex=True)
    report("torch.stft fp32", 2.0 * t.abs().numpy().T if t.dim() == 2 else 2.0 * t.abs().numpy()[:, :, 0])
    report("HIP model (as built)", hip_rfft_mag(xw, n_fft))
    report("HIP model, fused untangle add", hip_rfft_mag(xw, n_fft, untangle="fused"))
    report("HIP model, plain complex multiply", hip_rfft_mag(xw, n_fft, mode="plain"))
    report("HIP model, once-rounded twiddles", hip_rfft_mag(xw, n_fft, mode="exact"))
    for alt in ([8, 8, 4], [16, 16], [4, 8, 8], [8, 4, 8], [2, 8, 16]):
        if n_fft == 512 and np.prod(alt) == 256:
            RADICES[n_fft] = alt
            report(f"HIP model, radices {alt}", hip_rfft_mag(xw, n_fft))
    RADICES[512] = [4, 4, 4, 4]


if __name__ == "__main__":
    main()
