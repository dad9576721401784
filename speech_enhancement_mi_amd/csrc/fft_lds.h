// fft_lds.h - mixed-radix (2/4/5) Stockham autosort FFT passes over LDS-resident complex arrays.
//
// Used by the STFT / iSTFT kernels (stft.hip.h): n_fft = 400 = 4*4*5*5 (reference default,
// config.yaml:205-217) and 512 = 4*4*4*4*2.  One "task" is one radix-R butterfly of one transform; a
// workgroup runs `nfft` independent transforms side by side so every pass has nfft*N/R tasks.
//
// The pass code is plain C++ on (tid, nthreads) so the same source is unit-tested on the host
// (tests/test_fft_host.py builds it with g++); on the device the arrays live in LDS and a
// __syncthreads() separates passes.
#pragma once

#ifndef SE_HD
#ifdef __HIPCC__
#define SE_HD __host__ __device__ __forceinline__
#else
#define SE_HD inline
#endif
#endif

struct alignas(8) cf2 {
    float x, y;
};

SE_HD cf2 cmul(cf2 a, cf2 b) { return cf2{a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
SE_HD cf2 cadd(cf2 a, cf2 b) { return cf2{a.x + b.x, a.y + b.y}; }
SE_HD cf2 csub(cf2 a, cf2 b) { return cf2{a.x - b.x, a.y - b.y}; }
SE_HD cf2 cmuli_neg(cf2 a) { return cf2{a.y, -a.x}; }  // a * (-i)

SE_HD void dft2(cf2 *v) {
    cf2 a = v[0], b = v[1];
    v[0] = cadd(a, b);
    v[1] = csub(a, b);
}

SE_HD void dft4(cf2 *v) {
    cf2 t0 = cadd(v[0], v[2]), t1 = csub(v[0], v[2]);
    cf2 t2 = cadd(v[1], v[3]), t3 = cmuli_neg(csub(v[1], v[3]));
    v[0] = cadd(t0, t2);
    v[1] = cadd(t1, t3);
    v[2] = csub(t0, t2);
    v[3] = csub(t1, t3);
}

SE_HD void dft5(cf2 *v) {
    const float c1 = 0.30901699437494745f, s1 = 0.95105651629515353f;
    const float c2 = -0.80901699437494734f, s2 = 0.58778525229247314f;
    cf2 a0 = v[0];
    cf2 b1 = cadd(v[1], v[4]), b2 = cadd(v[2], v[3]);
    cf2 d1 = csub(v[1], v[4]), d2 = csub(v[2], v[3]);
    cf2 m1 = cf2{a0.x + c1 * b1.x + c2 * b2.x, a0.y + c1 * b1.y + c2 * b2.y};
    cf2 m2 = cf2{a0.x + c2 * b1.x + c1 * b2.x, a0.y + c2 * b1.y + c1 * b2.y};
    cf2 n1 = cmuli_neg(cf2{s1 * d1.x + s2 * d2.x, s1 * d1.y + s2 * d2.y});
    cf2 n2 = cmuli_neg(cf2{s2 * d1.x - s1 * d2.x, s2 * d1.y - s1 * d2.y});
    v[0] = cf2{a0.x + b1.x + b2.x, a0.y + b1.y + b2.y};
    v[1] = cadd(m1, n1);
    v[4] = csub(m1, n1);
    v[2] = cadd(m2, n2);
    v[3] = csub(m2, n2);
}

// One Stockham pass of radix R over `nfft` transforms of length N stored back to back (stride `ld`
// complex elements) in `in`, result in `out`.  Ns = product of the radices of the previous passes.
// tw[m] = exp(-2*pi*i*m/N), m in [0,N).
// tws = stride of the twiddle table (tw holds exp(-2*pi*i*m/(N*tws)); tws=2 lets an N/2-point
// transform share the n_fft-point table of the real-FFT post-processing).
template <int R>
SE_HD void fft_pass(const cf2 *in, cf2 *out, int N, int ld, int nfft, int Ns, const cf2 *tw, int tid, int nthreads, int tws = 1) {
    const int per = N / R;
    const int total = per * nfft;
    const int twstep = N / (Ns * R);
    for (int q = tid; q < total; q += nthreads) {
        const int f = q / per, j = q - f * per;
        const cf2 *src = in + f * ld;
        cf2 *dst = out + f * ld;
        const int k = j % Ns;
        cf2 v[R];
#pragma unroll
        for (int r = 0; r < R; r++) {
            cf2 a = src[j + r * per];
            if (r > 0 && Ns > 1) a = cmul(a, tw[((r * k * twstep) % N) * tws]);
            v[r] = a;
        }
        if (R == 2) dft2(v);
        if (R == 4) dft4(v);
        if (R == 5) dft5(v);
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int r = 0; r < R; r++) dst[j0 + r * Ns] = v[r];
    }
}

// ---- real <-> half-length complex packing (each frame is transformed on its own, so an all-zero frame
// gives exact zeros - the reference's arctan(im/(re+1e-8)) feature is ill-conditioned around 0 and a
// two-frames-per-transform trick would leak rounding noise of the neighbour frame into silent frames).
// x real, N = 2*N2 samples; z[n] = x[2n] + i x[2n+1]; Z = FFT_N2(z); twN[k] = exp(-2*pi*i*k/N).
// X[k] for k in [0, N2]:
SE_HD cf2 rfft_post(const cf2 *Z, int k, int N2, const cf2 *twN) {
    const cf2 zk = Z[k == N2 ? 0 : k];
    const cf2 zc = Z[(N2 - k) % N2];  // conj applied below
    const cf2 e = cf2{0.5f * (zk.x + zc.x), 0.5f * (zk.y - zc.y)};
    const cf2 o = cf2{0.5f * (zk.y + zc.y), -0.5f * (zk.x - zc.x)};  // (zk - conj zc) / (2i)
    cf2 x = cadd(e, cmul(twN[k], o));
    // DC and Nyquist of a real signal are real: return exactly +0 like torch/pocketfft (the arithmetic above can
    // leave -0.0, which atan2-based phase features of CRN_ELU.py:370 turn into a 2*pi flip)
    if (k == 0 || k == N2) x.y = 0.0f;
    return x;
}
// inverse: from the onesided spectrum X (Im of DC/Nyquist ignored by the caller) build conj(Z[k]),
// k in [0, N2), so that z = conj(FFT_N2(conj Z)) / N2 and x[2n] = Re z[n], x[2n+1] = Im z[n].
SE_HD cf2 irfft_pre(cf2 xk, cf2 xnk /* X[N2-k] */, int k, const cf2 *twN) {
    const cf2 e = cf2{0.5f * (xk.x + xnk.x), 0.5f * (xk.y - xnk.y)};
    const cf2 d = cf2{0.5f * (xk.x - xnk.x), 0.5f * (xk.y + xnk.y)};  // (X[k] - conj X[N2-k]) / 2
    const cf2 w = cf2{twN[k].x, -twN[k].y};                            // exp(+2*pi*i*k/N)
    const cf2 o = cmul(d, w);
    const cf2 z = cf2{e.x - o.y, e.y + o.x};  // E + i O
    return cf2{z.x, -z.y};
}

// Radix plan: fills radices[], returns count (0 if N has a prime factor other than 2 and 5).
inline int fft_plan(int N, int *radices) {
    int n = 0;
    while (N % 5 == 0) { radices[n++] = 5; N /= 5; }
    while (N % 4 == 0) { radices[n++] = 4; N /= 4; }
    while (N % 2 == 0) { radices[n++] = 2; N /= 2; }
    return N == 1 ? n : 0;
}
