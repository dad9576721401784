// Host check of the Stockham pass code and the real<->half-complex packing shared with the HIP STFT
// kernels (csrc/fft_lds.h).  Usage: fft_host_check N
//   prints "N npass err_complex err_rfft err_irfft" (max abs error vs double-precision DFTs);
//   also verifies that an all-zero frame transforms to exact zeros.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../speech_enhancement_mi_amd/csrc/fft_lds.h"

static void run_fft(std::vector<cf2> &a, std::vector<cf2> &b, cf2 *&res, int N, int ld, int nfft, const int *rad, int np,
                    const cf2 *tw, int tws) {
    const int nthreads = 64;
    cf2 *src = a.data(), *dst = b.data();
    int Ns = 1;
    for (int p = 0; p < np; p++) {
        for (int tid = 0; tid < nthreads; tid++) {
            if (rad[p] == 5) fft_pass<5>(src, dst, N, ld, nfft, Ns, tw, tid, nthreads, tws);
            if (rad[p] == 4) fft_pass<4>(src, dst, N, ld, nfft, Ns, tw, tid, nthreads, tws);
            if (rad[p] == 2) fft_pass<2>(src, dst, N, ld, nfft, Ns, tw, tid, nthreads, tws);
        }
        Ns *= rad[p];
        std::swap(src, dst);
    }
    res = src;
}

int main(int argc, char **argv) {
    int N = argc > 1 ? atoi(argv[1]) : 400;
    int rad[16];
    int np = fft_plan(N, rad);
    if (!np) { printf("noplan\n"); return 2; }
    const int nfft = 3, ld = N;
    std::vector<cf2> a(nfft * ld), b(nfft * ld), tw(N);
    for (int m = 0; m < N; m++) tw[m] = cf2{(float)cos(2 * M_PI * m / N), (float)-sin(2 * M_PI * m / N)};
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.0f - 1; };
    for (auto &v : a) { v.x = rnd(); v.y = rnd(); }
    std::vector<cf2> in = a;
    cf2 *res;
    run_fft(a, b, res, N, ld, nfft, rad, np, tw.data(), 1);
    double e1 = 0;
    for (int f = 0; f < nfft; f++)
        for (int k = 0; k < N; k++) {
            double re = 0, im = 0;
            for (int n = 0; n < N; n++) {
                double ang = -2 * M_PI * (double)((long)k * n % N) / N;
                re += in[f * ld + n].x * cos(ang) - in[f * ld + n].y * sin(ang);
                im += in[f * ld + n].x * sin(ang) + in[f * ld + n].y * cos(ang);
            }
            e1 = fmax(e1, fmax(fabs(re - res[f * ld + k].x), fabs(im - res[f * ld + k].y)));
        }
    // real FFT via N/2 complex; frame 1 is all zeros
    const int N2 = N / 2;
    int rad2[16];
    int np2 = fft_plan(N2, rad2);
    if (!np2) { printf("noplan2\n"); return 2; }
    std::vector<float> x(nfft * N);
    for (auto &v : x) v = rnd();
    for (int n = 0; n < N; n++) x[1 * N + n] = 0.0f;
    std::vector<cf2> za(nfft * N2), zb(nfft * N2);
    for (int f = 0; f < nfft; f++)
        for (int n = 0; n < N2; n++) za[f * N2 + n] = cf2{x[f * N + 2 * n], x[f * N + 2 * n + 1]};
    run_fft(za, zb, res, N2, N2, nfft, rad2, np2, tw.data(), 2);
    double e2 = 0;
    std::vector<cf2> X(nfft * (N2 + 1));
    bool zero_ok = true;
    for (int f = 0; f < nfft; f++)
        for (int k = 0; k <= N2; k++) {
            cf2 v = rfft_post(res + f * N2, k, N2, tw.data());
            X[f * (N2 + 1) + k] = v;
            if (f == 1 && (v.x != 0.0f || v.y != 0.0f)) zero_ok = false;
            double re = 0, im = 0;
            for (int n = 0; n < N; n++) {
                double ang = -2 * M_PI * (double)((long)k * n % N) / N;
                re += x[f * N + n] * cos(ang);
                im += x[f * N + n] * sin(ang);
            }
            e2 = fmax(e2, fmax(fabs(re - v.x), fabs(im - v.y)));
        }
    // inverse: X -> x
    for (int f = 0; f < nfft; f++)
        for (int k = 0; k < N2; k++) {
            cf2 xk = X[f * (N2 + 1) + k], xn = X[f * (N2 + 1) + N2 - k];
            if (k == 0) { xk.y = 0; xn.y = 0; }
            za[f * N2 + k] = irfft_pre(xk, xn, k, tw.data());
        }
    run_fft(za, zb, res, N2, N2, nfft, rad2, np2, tw.data(), 2);
    double e3 = 0;
    for (int f = 0; f < nfft; f++)
        for (int n = 0; n < N2; n++) {
            float x0 = res[f * N2 + n].x / N2, x1 = -res[f * N2 + n].y / N2;
            e3 = fmax(e3, fmax(fabs(x0 - x[f * N + 2 * n]), fabs(x1 - x[f * N + 2 * n + 1])));
        }
    printf("%d %d %.3e %.3e %.3e zero_exact=%d\n", N, np, e1, e2, e3, (int)zero_ok);
    return (e1 < 2e-4 && e2 < 2e-4 && e3 < 2e-6 && zero_ok) ? 0 : 1;
}
