// hx_fft_core.h -- index math and butterflies of the in-LDS power-of-two FFT (fused radix-2^K passes on a
// padded buffer) and of the radix-4 DIF split / Bluestein chirp used by the ring Fourier stage.  Pure functions,
// usable from device code and from the host-side emulation test (tests/csrc).
#pragma once
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define HX_HD __host__ __device__ __forceinline__
#else
#include <cmath>
#define HX_HD inline
struct double2 {
    double x, y;
};
#endif

namespace hxfft {

HX_HD double2 mk(double a, double b)
{
    double2 r;
    r.x = a;
    r.y = b;
    return r;
}
HX_HD double2 cadd(double2 a, double2 b) { return mk(a.x + b.x, a.y + b.y); }
HX_HD double2 csub(double2 a, double2 b) { return mk(a.x - b.x, a.y - b.y); }
HX_HD double2 cmul(double2 a, double2 b) { return mk(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
HX_HD double2 cmulc(double2 a, double2 b) /* a * conj(b) */
{
    return mk(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
HX_HD double2 cconj(double2 a) { return mk(a.x, -a.y); }
HX_HD double2 cscale(double2 a, double s) { return mk(a.x * s, a.y * s); }
HX_HD double2 mul_mi(double2 a) { return mk(a.y, -a.x); } /* a * (-i) */
HX_HD double2 mul_pi(double2 a) { return mk(-a.y, a.x); } /* a * (+i) */

HX_HD unsigned bitrev(unsigned v, int bits)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    v = (v >> 16) | (v << 16);
    return bits ? v >> (32 - bits) : 0u;
}

HX_HD int ilog2(unsigned v)
{
    int p = 0;
    while ((1u << p) < v) ++p;
    return p;
}

/* smallest power of two >= 2n-1 (Bluestein convolution length); n itself if n is 2^k */
HX_HD int fft_size_for(int n)
{
    if ((n & (n - 1)) == 0) return n;
    int m = 1;
    while (m < 2 * n - 1) m <<= 1;
    return m;
}

/* Twiddle provider of the LDS-resident FFT: W^k = exp(-2 pi i k / twN) = hi[k >> 6] * lo[k & 63] with
 * hi[a] = W^{64a}, lo[b] = W^b (twN/128 + 64 table entries instead of a twN/2-entry table in global
 * memory: a butterfly's twiddles then cost two LDS reads and a complex multiply, not an L2 round trip). */
struct TwFactored {
    const double2 *hi, *lo;
    HX_HD double2 operator[](int k) const
    {
        const double2 a = hi[k >> 6], b = lo[k & 63];
        double2 r;
        r.x = a.x * b.x - a.y * b.y;
        r.y = a.x * b.y + a.y * b.x;
        return r;
    }
};

/* ---- fused radix-2^K passes on the PADDED buffer (K <= 4) --------------------------------------------------------
 * Element e of a transform lives in slot e + (e >> 4) + (e >> 9) of the buffer (one empty 16-byte slot after every 16 and
 * one more after every 512): a thread that owns 16 consecutive elements (the h = 1 pass) then reads them with a lane stride
 * of 17 slots, 8 consecutive lanes of every other pass (h >= 16) read 8 consecutive slots, and the bit-reversed read-out of
 * a 4096-point transform (the equatorial rings: lane k reads element bitrev(k), i.e. 8 lanes differ in bits 9..11) hits 8
 * different slots mod 8 as well -- all free of bank conflicts for 128-bit accesses (8 lanes per 128-byte window; without
 * the second term that read-out is an 8-way conflict, the cost of eight extra passes).  K fused Gentleman-Sande stages = one radix-2^K butterfly on the elements p + j h,
 * j < 2^K: same in-place layout and bit-reversed final order as K radix-2 stages, one LDS round trip instead of K. */
HX_HD int lds_slot(int e) { return e + (e >> 4) + (e >> 9); }
HX_HD int lds_fft_slots(int M) { return M + (M >> 4) + (M >> 9) + 1; }

/* x * exp(-2 pi i k / 16), 0 <= k < 8 (k is a constant after unrolling) */
HX_HD double2 rot16(double2 x, int k)
{
    const double C = 0.92387953251128673848, S = 0.38268343236508977173, R = 0.70710678118654752440;
    switch (k & 7) {
    case 0: return x;
    case 1: return mk(C * x.x + S * x.y, C * x.y - S * x.x);
    case 2: return mk(R * (x.x + x.y), R * (x.y - x.x));
    case 3: return mk(S * x.x + C * x.y, S * x.y - C * x.x);
    case 4: return mk(x.y, -x.x);
    case 5: return mk(C * x.y - S * x.x, -(S * x.y + C * x.x));
    case 6: return mk(R * (x.y - x.x), -(R * (x.x + x.y)));
    default: return mk(S * x.y - C * x.x, -(C * x.y + S * x.x));
    }
}
/* x * exp(+2 pi i k / 16) */
HX_HD double2 rot16c(double2 x, int k) { return cconj(rot16(cconj(x), k)); }

/* x[bitrev_K(q)] *= w^q (CONJ: conj(w)^q) for 0 < q < 2^K.  The powers are formed on the fly as two chains that advance by
 * w^2 (odd and even q) instead of a table of 2^K - 1 of them: 12 registers instead of 60 for K = 4, which is what lets a
 * radix-16 butterfly sit beside the ring's pixels in the register file. */
template <int K, bool CONJ>
HX_HD void apply_tw_powers(double2 *x, double2 w)
{
    constexpr int R = 1 << K;
    const double2 w2 = cmul(w, w);
    double2 wo = w, we = w2;
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int q = 1; q < R; q += 2) {
        x[bitrev(q, K)] = CONJ ? cmulc(x[bitrev(q, K)], wo) : cmul(x[bitrev(q, K)], wo);
        if (q + 1 < R) x[bitrev(q + 1, K)] = CONJ ? cmulc(x[bitrev(q + 1, K)], we) : cmul(x[bitrev(q + 1, K)], we);
        if (q + 2 < R) {
            wo = cmul(wo, w2);
            we = cmul(we, w2);
        }
    }
}

/* the 2^K-point DFT on registers, constants only: x[j] <- y[bitrev_K(j)] */
template <int K>
HX_HD void dif_regs(double2 *x)
{
    constexpr int R = 1 << K;
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int s = 0; s < K; ++s) {
        const int half = R >> (s + 1);
#ifdef __HIPCC__
#pragma unroll
#endif
        for (int b = 0; b < R; b += 2 * half)
#ifdef __HIPCC__
#pragma unroll
#endif
            for (int jj = 0; jj < half; ++jj) {
                const double2 u = x[b + jj], v = x[b + jj + half];
                x[b + jj] = cadd(u, v);
                x[b + jj + half] = rot16(csub(u, v), jj * (8 / half));
            }
    }
}
/* its inverse (unnormalised): x[j] = y[bitrev_K(j)] -> natural order */
template <int K>
HX_HD void dit_inv_regs(double2 *x)
{
    constexpr int R = 1 << K;
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int s = K - 1; s >= 0; --s) {
        const int half = R >> (s + 1);
#ifdef __HIPCC__
#pragma unroll
#endif
        for (int b = 0; b < R; b += 2 * half)
#ifdef __HIPCC__
#pragma unroll
#endif
            for (int jj = 0; jj < half; ++jj) {
                const double2 u = x[b + jj], v = rot16c(x[b + jj + half], jj * (8 / half));
                x[b + jj] = cadd(u, v);
                x[b + jj + half] = csub(u, v);
            }
    }
}

/* butterfly number i < M >> K of the forward pass whose elements are h apart (half-sizes 2^(K-1) h ... h) */
template <int K, class TW>
HX_HD void dif_pass_butterfly(double2 *buf, int i, int h, TW tw, int twN)
{
    constexpr int R = 1 << K;
    const int t = i & (h - 1), p = ((i - t) << K) + t;
    double2 x[R];
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int j = 0; j < R; ++j) x[j] = buf[lds_slot(p + j * h)];
    dif_regs<K>(x);
    if (h > 1 && t) apply_tw_powers<K, false>(x, tw[t * (twN / (R * h))]);
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int j = 0; j < R; ++j) buf[lds_slot(p + j * h)] = x[j];
}
/* inverse pass (conjugate twiddles), bit-reversed input -> natural output after the passes h = 1, 2^K1, ... */
template <int K, class TW>
HX_HD void dit_inv_pass_butterfly(double2 *buf, int i, int h, TW tw, int twN)
{
    constexpr int R = 1 << K;
    const int t = i & (h - 1), p = ((i - t) << K) + t;
    double2 x[R];
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int j = 0; j < R; ++j) x[j] = buf[lds_slot(p + j * h)];
    if (h > 1 && t) apply_tw_powers<K, true>(x, tw[t * (twN / (R * h))]);
    dit_inv_regs<K>(x);
#ifdef __HIPCC__
#pragma unroll
#endif
    for (int j = 0; j < R; ++j) buf[lds_slot(p + j * h)] = x[j];
}

/* Pass schedule of a 2^p-point transform: the last forward pass (h = 1) takes min(p, 4) stages -- 16 consecutive elements per
 * thread, which is what the padding is made for -- and the other stages are spread evenly over as few passes of <= 4 stages
 * as possible.  Forward order: pass 0 has its elements 2^(p - K_0) apart; the inverse runs the schedule backwards. */
HX_HD int fft_sched_np(int p) { return p <= 0 ? 0 : (p - (p < 4 ? p : 4) + 3) / 4 + 1; }
HX_HD int fft_sched_k(int p, int a)  /* stages of forward pass a < np */
{
    const int last = p < 4 ? p : 4, rest = p - last, nfront = (rest + 3) / 4;
    return a < nfront ? rest / nfront + (a < rest % nfront ? 1 : 0) : last;
}

/* radix-4 DIF pre-step: t_r = sum_q z_q (-i)^{q r}.  X[4k+r] = DFT_n( t_r[j] W_{4n}^{j r} ) */
HX_HD double2 dif4_combine(double2 z0, double2 z1, double2 z2, double2 z3, int r)
{
    switch (r & 3) {
    case 0: return cadd(cadd(z0, z1), cadd(z2, z3));
    case 1: return cadd(csub(z0, z2), mul_mi(csub(z1, z3)));
    case 2: return cadd(csub(z0, z1), csub(z2, z3));
    default: return cadd(csub(z0, z2), mul_pi(csub(z1, z3)));
    }
}

/* x mod d for x < 2^31, d < 2^31, with inv = 1.0 / d in double precision: floor(x inv) is off by at most one, corrected by
 * two compares -- a handful of instructions where the 64-bit integer division behind `%` is a subroutine of hundreds (the ring
 * kernel reduces one phase numerator per pixel and sub-DFT). */
HX_HD unsigned mod_by_inv(unsigned x, unsigned d, double inv)
{
    const int q = (int)((double)x * inv);
    int r = (int)x - q * (int)d;
    if (r < 0) r += (int)d;
    if (r >= (int)d) r -= (int)d;
    return (unsigned)r;
}

/* numerator q of the load-phase phase exp(-i pi q / (2n)):  q = j r (+ 2 j^2 if Bluestein),
 * reduced mod 4n */
HX_HD unsigned load_phase_num(unsigned j, unsigned r, unsigned n, bool bluestein)
{
    unsigned long long q = (unsigned long long)j * r;
    if (bluestein) q += 2ull * j * j;
    return (unsigned)(q % (4ull * n));
}

/* numerator of the chirp exp(+- i pi q / n): q = k^2 mod 2n */
HX_HD unsigned chirp_num(unsigned k, unsigned n)
{
    return (unsigned)(((unsigned long long)k * k) % (2ull * n));
}

}  // namespace hxfft
