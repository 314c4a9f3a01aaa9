// Host emulation of the ring sub-DFT kernel's algorithm (hx_fft_core.h): radix-4 DIF split,
// in-place DIF FFT (fused radix-2^K passes on the padded buffer) / Bluestein with bit-reversed spectrum, DIT inverse.  Compares against a
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

// same pass schedule as lds_fft_dif / lds_fft_dit_inv in hx_sht_common.h: fused radix-2^K passes (fft_sched_k) on the padded buffer
template <class TW>
static void pass_dif(std::vector<double2> &b, int M, int K, int h, TW tw, int twN)
{
    for (int i = 0; i < (M >> K); ++i) {
        if (K == 4) dif_pass_butterfly<4>(b.data(), i, h, tw, twN);
        else if (K == 3) dif_pass_butterfly<3>(b.data(), i, h, tw, twN);
        else if (K == 2) dif_pass_butterfly<2>(b.data(), i, h, tw, twN);
        else dif_pass_butterfly<1>(b.data(), i, h, tw, twN);
    }
}
template <class TW>
static void pass_dit_inv(std::vector<double2> &b, int M, int K, int h, TW tw, int twN)
{
    for (int i = 0; i < (M >> K); ++i) {
        if (K == 4) dit_inv_pass_butterfly<4>(b.data(), i, h, tw, twN);
        else if (K == 3) dit_inv_pass_butterfly<3>(b.data(), i, h, tw, twN);
        else if (K == 2) dit_inv_pass_butterfly<2>(b.data(), i, h, tw, twN);
        else dit_inv_pass_butterfly<1>(b.data(), i, h, tw, twN);
    }
}
template <class TW>
static void fft_dif(std::vector<double2> &b, int M, TW tw, int twN, bool skip_last = false)
{
    const int p = ilog2(M), np = fft_sched_np(p);
    int h = M;
    for (int a = 0; a < np - (skip_last ? 1 : 0); ++a) {
        h >>= fft_sched_k(p, a);
        pass_dif(b, M, fft_sched_k(p, a), h, tw, twN);
    }
}
template <class TW>
static void fft_dit_inv(std::vector<double2> &b, int M, TW tw, int twN, bool skip_first = false)
{
    const int p = ilog2(M), np = fft_sched_np(p);
    int h = 1;
    for (int a = np - 1; a >= 0; --a) {
        if (!(skip_first && a == np - 1)) pass_dit_inv(b, M, fft_sched_k(p, a), h, tw, twN);
        h <<= fft_sched_k(p, a);
    }
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
    std::vector<double2> buf(lds_fft_slots(M), mk(0, 0));
    for (int j = 0; j < n; ++j) {
        double2 t = dif4_combine(z[j], z[j + n], z[j + 2 * n], z[j + 3 * n], r);
        unsigned q = load_phase_num(j, r, n, blu);
        buf[lds_slot(j)] = cmul(t, expipi(-(double)q / (2.0 * n)));
    }
    std::vector<double2> out(n);
    if (!blu) {
        fft_dif(buf, M, twf, twN);
        for (int k = 0; k < n; ++k) out[k] = buf[lds_slot(bitrev(k, p))];
        return out;
    }
    // Bluestein filter spectrum (bit-reversed order), stored transposed as the init kernel does: hT[j * (M/16) + i] = H[16 i + j]
    std::vector<double2> h(lds_fft_slots(M), mk(0, 0));
    for (int j = 0; j < n; ++j) {
        double2 c = expipi((double)chirp_num(j, n) / n);
        h[lds_slot(j)] = c;
        if (j) h[lds_slot(M - j)] = c;
    }
    fft_dif(h, M, tw.data(), twN);
    if (M >= 16) {
        std::vector<double2> hT(M);
        for (int e = 0; e < M; ++e) hT[(e & 15) * (M / 16) + (e >> 4)] = h[lds_slot(e)];
        // the kernel's fused middle: last forward pass (h = 1, 16 consecutive elements), filter, first inverse pass, in registers
        fft_dif(buf, M, twf, twN, true);
        for (int i = 0; i < M / 16; ++i) {
            double2 x[16];
            for (int j = 0; j < 16; ++j) x[j] = buf[lds_slot(16 * i + j)];
            dif_regs<4>(x);
            for (int j = 0; j < 16; ++j) x[j] = cmul(x[j], hT[j * (M / 16) + i]);
            dit_inv_regs<4>(x);
            for (int j = 0; j < 16; ++j) buf[lds_slot(16 * i + j)] = x[j];
        }
        fft_dit_inv(buf, M, twf, twN, true);
    } else {
        fft_dif(buf, M, twf, twN);
        for (int e = 0; e < M; ++e) buf[lds_slot(e)] = cmul(buf[lds_slot(e)], h[lds_slot(e)]);
        fft_dit_inv(buf, M, twf, twN);
    }
    for (int k = 0; k < n; ++k)
        out[k] = cscale(cmul(buf[lds_slot(k)], expipi(-(double)chirp_num(k, n) / n)), 1.0 / M);
    return out;
}

int main()
{
    const int twN = 8192;
    auto tw = make_tw(twN);
    double worst = 0;
    int sizes[] = {1, 2, 3, 4, 5, 7, 8, 12, 16, 31, 32, 33, 64, 100, 127, 128, 255, 256, 257, 512, 1000, 1024, 1500, 2048};
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
    // the kernel's phase numerators: mod_by_inv (double-precision reciprocal) against the 64-bit integer forms
    long bad = 0;
    for (unsigned n = 1; n <= 8192; n += (n < 300 ? 1 : 37)) {
        const double inv4n = 0.25 / (double)n;
        for (unsigned j = 0; j < n; j += (n < 300 ? 1 : 3))
            for (unsigned r = 0; r < 4; ++r) {
                if (mod_by_inv(j * r + 2u * j * j, 4u * n, inv4n) != load_phase_num(j, r, n, true)) ++bad;
                if (mod_by_inv(j * r, 4u * n, inv4n) != load_phase_num(j, r, n, false)) ++bad;
                if (mod_by_inv(j * j, 2u * n, 2.0 * inv4n) != chirp_num(j, n)) ++bad;
            }
    }
    printf("phase numerators: %ld mismatches\n", bad);
    if (bad) { printf("FAIL\n"); return 1; }
    if (worst > 1e-12) { printf("FAIL\n"); return 1; }
    printf("OK\n");
    return 0;
}
