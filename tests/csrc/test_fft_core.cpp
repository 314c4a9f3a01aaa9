// Host emulation of the ring sub-DFT kernel's algorithm (hx_fft_core.h): radix-4 DIF split,
// in-place DIF FFT / Bluestein with bit-reversed spectrum, DIT inverse.  Compares against a
// direct O(n^2) DFT of the full length-4n ring.  Build: g++ -O2 -std=c++17 test_fft_core.cpp
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../heracles_amd/csrc/hx_fft_core.h"
using namespace hxfft;
typedef std::complex<long double> cld;

static std::vector<double2> make_tw(int twN)
{
    std::vector<double2> tw(twN / 2 > 0 ? twN / 2 : 1);
    for (int k = 0; k < twN / 2; ++k) {
        long double a = -2.0L * M_PIl * k / twN;
        tw[k] = mk((double)cosl(a), (double)sinl(a));
    }
    return tw;
}
static double2 expipi(double x) /* exp(i pi x) */ { return mk(cos(M_PI * x), sin(M_PI * x)); }

// same stage schedule as lds_fft_dif / lds_fft_dit_inv in hx_sht.hip: fused radix-4 stages,
// plus one radix-2 stage when log2(M) is odd
template <class TW>
static void fft_dif(std::vector<double2> &b, int M, TW tw, int twN)
{
    int h = M / 2;
    if (ilog2(M) & 1) {
        for (int i = 0; i < M / 2; ++i) dif_butterfly(b.data(), i, h, tw, twN);
        h >>= 1;
    }
    for (h >>= 1; h >= 1; h >>= 2)
        for (int i = 0; i < M / 4; ++i) dif4_butterfly(b.data(), i, h, tw, twN);
}
template <class TW>
static void fft_dit_inv(std::vector<double2> &b, int M, TW tw, int twN)
{
    int h = 1;
    for (; 4 * h <= M; h <<= 2)
        for (int i = 0; i < M / 4; ++i) dit4_inv_butterfly(b.data(), i, h, tw, twN);
    if (2 * h <= M)
        for (int i = 0; i < M / 2; ++i) dit_inv_butterfly(b.data(), i, h, tw, twN);
}

// emulate one (ring, r) sub-DFT: input z[4n], output Y[k] = X[4k+r], k<n
static std::vector<double2> subdft(const std::vector<double2> &z, int n, int r,
                                   const std::vector<double2> &tw, int twN)
{
    bool blu = (n & (n - 1)) != 0;
    int M = fft_size_for(n), p = ilog2(M);
    // the ring kernel's twiddles: factored tables hi[a] = W^{64a}, lo[b] = W^b (TwFactored); the
    // filter-spectrum kernel reads the full table
    std::vector<double2> hi(twN >= 128 ? twN / 128 : 1), lo(64, mk(1, 0));
    for (size_t a = 0; a < hi.size(); ++a) hi[a] = tw[a * 64];
    for (int b = 0; b < 64 && b < twN / 2; ++b) lo[b] = tw[b];
    TwFactored twf;
    twf.hi = hi.data();
    twf.lo = lo.data();
    std::vector<double2> buf(M, mk(0, 0));
    for (int j = 0; j < n; ++j) {
        double2 t = dif4_combine(z[j], z[j + n], z[j + 2 * n], z[j + 3 * n], r);
        unsigned q = load_phase_num(j, r, n, blu);
        buf[j] = cmul(t, expipi(-(double)q / (2.0 * n)));
    }
    std::vector<double2> out(n);
    if (!blu) {
        fft_dif(buf, M, twf, twN);
        for (int k = 0; k < n; ++k) out[k] = buf[bitrev(k, p)];
        return out;
    }
    // Bluestein filter spectrum (bit-reversed order), as the init kernel builds it
    std::vector<double2> h(M, mk(0, 0));
    for (int j = 0; j < n; ++j) {
        double2 c = expipi((double)chirp_num(j, n) / n);
        h[j] = c;
        if (j) h[M - j] = c;
    }
    fft_dif(h, M, tw.data(), twN);
    fft_dif(buf, M, twf, twN);
    for (int i = 0; i < M; ++i) buf[i] = cmul(buf[i], h[i]);
    fft_dit_inv(buf, M, twf, twN);
    for (int k = 0; k < n; ++k)
        out[k] = cscale(cmul(buf[k], expipi(-(double)chirp_num(k, n) / n)), 1.0 / M);
    return out;
}

int main()
{
    const int twN = 8192;
    auto tw = make_tw(twN);
    double worst = 0;
    int sizes[] = {1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 33, 64, 100, 127, 128, 255, 256, 257, 512, 1000};
    for (int n : sizes) {
        int N = 4 * n;
        std::vector<double2> z(N);
        srand(n);
        for (auto &v : z) v = mk(rand() / (double)RAND_MAX - 0.5, rand() / (double)RAND_MAX - 0.5);
        std::vector<cld> X(N);
        for (int k = 0; k < N; ++k) {
            cld s = 0;
            for (int j = 0; j < N; ++j) {
                long double a = -2.0L * M_PIl * (long double)(((long long)j * k) % N) / N;
                s += cld(z[j].x, z[j].y) * cld(cosl(a), sinl(a));
            }
            X[k] = s;
        }
        double err = 0, nrm = 0;
        for (int r = 0; r < 4; ++r) {
            auto Y = subdft(z, n, r, tw, twN);
            for (int k = 0; k < n; ++k) {
                cld d = cld(Y[k].x, Y[k].y) - X[4 * k + r];
                err = fmax(err, (double)std::abs(d));
                nrm = fmax(nrm, (double)std::abs(X[4 * k + r]));
            }
        }
        printf("n_sub %5d  M %5d  max err %.3e (rel %.3e)\n", n, fft_size_for(n), err, err / nrm);
        worst = fmax(worst, err / nrm);
    }
    if (worst > 1e-12) { printf("FAIL\n"); return 1; }
    printf("OK\n");
    return 0;
}
