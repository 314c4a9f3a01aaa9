// hx_fft_core.h -- index math and butterflies of the in-LDS power-of-two FFT and of the
// radix-4 DIF split / Bluestein chirp used by the ring Fourier stage.  Pure functions,
// usable from device code and from the host-side emulation test (tests/csrc).
#pragma once
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#define HX_HD __host__ __device__ inline
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

/* Gentleman-Sande (DIF) butterfly number i of the stage with half-size h.
 * tw[k] = exp(-2 pi i k / twN), k < twN/2 (a pointer to the full table or a TwFactored).  Natural-order input -> bit-reversed output
 * after stages h = M/2, M/4, ..., 1. */
template <class TW>
HX_HD void dif_butterfly(double2 *buf, int i, int h, TW tw, int twN)
{
    int t = i & (h - 1);
    int p0 = ((i - t) << 1) + t, p1 = p0 + h;
    double2 u = buf[p0], v = buf[p1];
    buf[p0] = cadd(u, v);
    double2 d = csub(u, v);
    buf[p1] = t ? cmul(d, tw[t * (twN / (2 * h))]) : d;
}

/* Cooley-Tukey (DIT) inverse butterfly, conj twiddles: bit-reversed input -> natural
 * output after stages h = 1, 2, ..., M/2 (unnormalised inverse DFT). */
template <class TW>
HX_HD void dit_inv_butterfly(double2 *buf, int i, int h, TW tw, int twN)
{
    int t = i & (h - 1);
    int p0 = ((i - t) << 1) + t, p1 = p0 + h;
    double2 u = buf[p0], v = buf[p1];
    if (t) v = cmulc(v, tw[t * (twN / (2 * h))]);
    buf[p0] = cadd(u, v);
    buf[p1] = csub(u, v);
}

/* Two fused Gentleman-Sande stages (half-sizes 2h and h) = one radix-4 DIF butterfly on
 * (p, p+h, p+2h, p+3h); same in-place layout and final (bit-reversed) order as the two
 * radix-2 stages.  i in [0, M/4), W = exp(-2 pi i / 4h). */
template <class TW>
HX_HD void dif4_butterfly(double2 *buf, int i, int h, TW tw, int twN)
{
    int t = i & (h - 1);
    int p = ((i - t) << 2) + t;
    double2 x0 = buf[p], x1 = buf[p + h], x2 = buf[p + 2 * h], x3 = buf[p + 3 * h];
    double2 a = cadd(x0, x2), b = cadd(x1, x3), c = csub(x0, x2), d = mul_mi(csub(x1, x3));
    double2 y0 = cadd(a, b), y1 = csub(a, b), y2 = cadd(c, d), y3 = csub(c, d);
    if (t) {
        double2 w1 = tw[t * (twN / (4 * h))], w2 = tw[2 * t * (twN / (4 * h))];
        y1 = cmul(y1, w2);
        y2 = cmul(y2, w1);
        y3 = cmul(y3, cmul(w1, w2));
    }
    buf[p] = y0; buf[p + h] = y1; buf[p + 2 * h] = y2; buf[p + 3 * h] = y3;
}

/* Two fused inverse Cooley-Tukey stages (half-sizes h then 2h), conj twiddles. */
template <class TW>
HX_HD void dit4_inv_butterfly(double2 *buf, int i, int h, TW tw, int twN)
{
    int t = i & (h - 1);
    int p = ((i - t) << 2) + t;
    double2 x0 = buf[p], x1 = buf[p + h], x2 = buf[p + 2 * h], x3 = buf[p + 3 * h];
    if (t) {
        double2 w2 = tw[2 * t * (twN / (4 * h))];  /* W_{2h}^t */
        x1 = cmulc(x1, w2);
        x3 = cmulc(x3, w2);
    }
    double2 a0 = cadd(x0, x1), a1 = csub(x0, x1), a2 = cadd(x2, x3), a3 = csub(x2, x3);
    if (t) {
        double2 w1 = tw[t * (twN / (4 * h))];      /* W_{4h}^t */
        a2 = cmulc(a2, w1);
        a3 = cmulc(a3, w1);
    }
    a3 = mul_pi(a3);                               /* conj(W_{4h}^{t+h}) = conj(W^t) * (+i) */
    buf[p] = cadd(a0, a2); buf[p + 2 * h] = csub(a0, a2);
    buf[p + h] = cadd(a1, a3); buf[p + 3 * h] = csub(a1, a3);
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
