/* hx_cpu_fast.c -- a VECTORISED CPU restatement of HEALPix map2alm (spin 0 and 2), for bench.py's `cpu_baseline` leg ONLY.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE.  Nothing under heracles_amd/ may call into this file.  It is not the checker either: the
 * scalar oracle (hx_oracle.c) stays the checker and checks THIS file at 1e-11 (tests/test_oracle_fast.py).  It exists because the
 * engines the reference runs on the CPU (healpy / ducc0 behind heracles/healpy.py:183-189) are absent from this image, and a
 * scalar port says nothing about what the host cores can do: SURVEY section 8d (3) asks for "the build's own C++/OpenMP CPU
 * restatement on all cores".  kind = "port-vectorised" -- never "ducc".
 *
 * Shape (the textbook organisation of a CPU SHT, written from the definitions):
 *   ring stage     one complex FFT per ring PAIR (z = f_N + i f_S), radix-2 DIF/DIT without bit reversal, Bluestein for the cap
 *                  rings; output planes Fe = w (F_N + F_S), Fo = w (F_N - F_S) stored [m][ring pair] so that the Legendre stage
 *                  reads eight ring pairs of one m with one vector load;
 *   Legendre stage threads over m (heaviest first); ring pairs across the 8 AVX-512 lanes; scaled three-term recursion
 *                  (value = v * 2^(300 e), per lane) -- chains below 2^-600 of their value only run the recursion -- libsharp's
 *                  published mlim rule for the polar rings; sums over rings kept as lane vectors per l in an L1-resident block
 *                  of multipoles and reduced across lanes once per (l, m).
 * AVX-512 (F + DQ) is required: the file is compiled with -mavx512f -mavx512dq and hxf_map2alm returns -3 on a CPU without it.
 */
#include <complex.h>
#include <immintrin.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef double _Complex cplx;
typedef __m512d v8;

#define SCB 300
static const double BIG = 0x1p+300, SMALL = 0x1p-300;

static double wall(void)
{
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

int hxf_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int hxf_supported(void) { return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq"); }

/* ---- scaled scalars: value = v * 2^(SCB e), |v| kept inside [2^-300, 2^300] ---------------------------------------- */
typedef struct { double v; int e; } sval;
static inline sval snorm(sval s)
{
    double a = fabs(s.v);
    if (a == 0.0) return s;
    while (a < SMALL) { s.v *= BIG; s.e -= 1; a *= BIG; }
    while (a > BIG) { s.v *= SMALL; s.e += 1; a *= SMALL; }
    return s;
}
static inline sval smul(sval a, sval b) { sval r = { a.v * b.v, a.e + b.e }; return snorm(r); }
static inline sval smuld(sval a, double b) { sval r = { a.v * b, a.e }; return snorm(r); }
/* a^n by squaring (n >= 0) */
static sval spow(double a, int n)
{
    sval r = { 1.0, 0 }, p = { a, 0 };
    p = snorm(p);
    while (n) {
        if (n & 1) r = smul(r, p);
        n >>= 1;
        if (n) p = smul(p, p);
    }
    return r;
}

/* ---- geometry of the north ring pairs (pair i = rings i + 1 and 4 nside - 1 - i; the equator has no partner) ------- */
typedef struct {
    int nside, nrp, nrp_pad;
    double *z, *omz, *sth;
    int *nphi, *shifted;
    int64_t *startn, *starts; /* starts = -1: no southern partner */
} geom;

static geom make_geom(int nside)
{
    geom g;
    int64_t ns = nside, npix = 12 * ns * ns, ncap = 2 * ns * (ns - 1);
    g.nside = nside;
    g.nrp = 2 * nside;
    g.nrp_pad = (g.nrp + 7) / 8 * 8;
    g.z = calloc(g.nrp_pad, sizeof(double));
    g.omz = calloc(g.nrp_pad, sizeof(double));
    g.sth = calloc(g.nrp_pad, sizeof(double));
    g.nphi = calloc(g.nrp_pad, sizeof(int));
    g.shifted = calloc(g.nrp_pad, sizeof(int));
    g.startn = calloc(g.nrp_pad, sizeof(int64_t));
    g.starts = calloc(g.nrp_pad, sizeof(int64_t));
    double fact2 = 4.0 / (double)npix, fact1 = (double)(2 * ns) * fact2;
    for (int i = 0; i < g.nrp; ++i) {
        int nr = i + 1;
        if (nr < nside) {
            double tmp = (double)nr * (double)nr * fact2;
            g.z[i] = 1.0 - tmp;
            g.omz[i] = tmp;
            g.sth[i] = sqrt(tmp * (2.0 - tmp));
            g.nphi[i] = 4 * nr;
            g.startn[i] = 2 * (int64_t)nr * (nr - 1);
            g.shifted[i] = 1;
        } else {
            g.z[i] = (double)(2 * nside - nr) * fact1;
            g.omz[i] = 1.0 - g.z[i];
            g.sth[i] = sqrt((1.0 - g.z[i]) * (1.0 + g.z[i]));
            g.nphi[i] = 4 * nside;
            g.startn[i] = ncap + (int64_t)(nr - nside) * 4 * ns;
            g.shifted[i] = ((nr - nside) & 1) == 0;
        }
        g.starts[i] = nr == 2 * nside ? -1 : npix - g.startn[i] - g.nphi[i];
    }
    return g;
}
static void free_geom(geom *g)
{
    free(g->z); free(g->omz); free(g->sth); free(g->nphi); free(g->shifted); free(g->startn); free(g->starts);
}

/* libsharp's published rule for the largest order m that contributes on a ring (sharp_get_mlim) */
static int ring_mlim(int lmax, int spin, double sth, double cth)
{
    double ofs = lmax * 0.01;
    if (ofs < 100.) ofs = 100.;
    double b = -2 * spin * fabs(cth);
    double t1 = lmax * sth + ofs;
    double c = (double)spin * spin - t1 * t1;
    double discr = b * b - 4 * c;
    if (discr <= 0) return lmax;
    double res = (-b + sqrt(discr)) / 2.;
    if (res > lmax) res = lmax;
    return (int)(res + 0.5);
}

/* ---- FFT: radix 2, split re / im, twiddles per stage; forward = DIF (natural in, bit-reversed out), inverse = DIT
 *      (bit-reversed in, natural out, unnormalised): a convolution needs no bit reversal at all ------------------------ */
typedef struct { int nmax; double *wr, *wi; } twid; /* stage of length len: entries [len/2 .. len), w = exp(-2 pi i k / len) */

static twid make_twid(int nmax)
{
    twid t;
    t.nmax = nmax;
    t.wr = malloc(sizeof(double) * (nmax > 1 ? nmax : 2));
    t.wi = malloc(sizeof(double) * (nmax > 1 ? nmax : 2));
    for (int len = 2; len <= nmax; len <<= 1)
        for (int k = 0; k < len / 2; ++k) {
            double a = -2.0 * M_PI * k / len;
            t.wr[len / 2 + k] = cos(a);
            t.wi[len / 2 + k] = sin(a);
        }
    return t;
}

static void fft_dif(double *restrict re, double *restrict im, int n, const twid *t)
{
    for (int len = n; len >= 2; len >>= 1) {
        const int half = len >> 1;
        const double *restrict wr = t->wr + half, *restrict wi = t->wi + half;
        for (int i0 = 0; i0 < n; i0 += len) {
            double *restrict ar = re + i0, *restrict ai = im + i0, *restrict br = re + i0 + half, *restrict bi = im + i0 + half;
            for (int k = 0; k < half; ++k) {
                const double ur = ar[k], ui = ai[k], vr = br[k], vi = bi[k];
                const double dr = ur - vr, di = ui - vi;
                ar[k] = ur + vr;
                ai[k] = ui + vi;
                br[k] = dr * wr[k] - di * wi[k];
                bi[k] = dr * wi[k] + di * wr[k];
            }
        }
    }
}
static void ifft_dit(double *restrict re, double *restrict im, int n, const twid *t)
{
    for (int len = 2; len <= n; len <<= 1) {
        const int half = len >> 1;
        const double *restrict wr = t->wr + half, *restrict wi = t->wi + half;
        for (int i0 = 0; i0 < n; i0 += len) {
            double *restrict ar = re + i0, *restrict ai = im + i0, *restrict br = re + i0 + half, *restrict bi = im + i0 + half;
            for (int k = 0; k < half; ++k) {
                const double vr = br[k] * wr[k] + bi[k] * wi[k], vi = bi[k] * wr[k] - br[k] * wi[k]; /* times conj(w) */
                const double ur = ar[k], ui = ai[k];
                ar[k] = ur + vr;
                ai[k] = ui + vi;
                br[k] = ur - vr;
                bi[k] = ui - vi;
            }
        }
    }
}
static inline int brev(int x, int bits)
{
    unsigned v = (unsigned)x;
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    v = (v >> 16) | (v << 16);
    return (int)(v >> (32 - bits));
}
static inline int ilog2(int n) { int b = 0; while ((1 << b) < n) ++b; return b; }

/* X[k] = sum_j x[j] exp(-2 pi i j k / n), k < n, any n; x in (xr, xi) length n; result in (Xr, Xi) natural order.
 * scratch: 4 arrays of M doubles (M = n if a power of two, else the power of two >= 2n - 1) + 2 of n */
typedef struct { double *ar, *ai, *br, *bi, *cr, *ci; } fftws;
static void dft_forward(const double *xr, const double *xi, int n, double *Xr, double *Xi, const twid *t, fftws *w)
{
    if ((n & (n - 1)) == 0) {
        memcpy(w->ar, xr, sizeof(double) * n);
        memcpy(w->ai, xi, sizeof(double) * n);
        fft_dif(w->ar, w->ai, n, t);
        const int bits = ilog2(n);
        for (int k = 0; k < n; ++k) {
            const int q = bits ? brev(k, bits) : 0;
            Xr[k] = w->ar[q];
            Xi[k] = w->ai[q];
        }
        return;
    }
    int M = 1;
    while (M < 2 * n - 1) M <<= 1;
    /* chirp c[j] = exp(-i pi j^2 / n) */
    for (int j = 0; j < n; ++j) {
        const int64_t q = ((int64_t)j * j) % (2 * (int64_t)n);
        const double a = -M_PI * (double)q / n;
        w->cr[j] = cos(a);
        w->ci[j] = sin(a);
    }
    memset(w->ar, 0, sizeof(double) * M);
    memset(w->ai, 0, sizeof(double) * M);
    memset(w->br, 0, sizeof(double) * M);
    memset(w->bi, 0, sizeof(double) * M);
    for (int j = 0; j < n; ++j) {
        w->ar[j] = xr[j] * w->cr[j] - xi[j] * w->ci[j];
        w->ai[j] = xr[j] * w->ci[j] + xi[j] * w->cr[j];
    }
    w->br[0] = w->cr[0];
    w->bi[0] = -w->ci[0];
    for (int j = 1; j < n; ++j) {
        w->br[j] = w->br[M - j] = w->cr[j];
        w->bi[j] = w->bi[M - j] = -w->ci[j];
    }
    fft_dif(w->ar, w->ai, M, t);
    fft_dif(w->br, w->bi, M, t);
    for (int j = 0; j < M; ++j) {
        const double pr = w->ar[j] * w->br[j] - w->ai[j] * w->bi[j], pi_ = w->ar[j] * w->bi[j] + w->ai[j] * w->br[j];
        w->ar[j] = pr;
        w->ai[j] = pi_;
    }
    ifft_dit(w->ar, w->ai, M, t);
    const double inv = 1.0 / M;
    for (int k = 0; k < n; ++k) {
        Xr[k] = (w->ar[k] * w->cr[k] - w->ai[k] * w->ci[k]) * inv;
        Xi[k] = (w->ar[k] * w->ci[k] + w->ai[k] * w->cr[k]) * inv;
    }
}

/* ---- ring stage: planes[4][(mmax + 1)][nrp_pad] = Fe.re, Fe.im, Fo.re, Fo.im of one component --------------------- */
static void ring_stage(const geom *g, int lmax, const double *map, const double *pw, double *planes, const twid *t)
{
    const int mmax = lmax, nside = g->nside;
    const size_t plane = (size_t)(mmax + 1) * g->nrp_pad;
    const double wpix = 4.0 * M_PI / (12.0 * (double)nside * nside);
    const int nmax = 4 * nside;
    int Mmax = 1;
    while (Mmax < 2 * nmax - 1) Mmax <<= 1;
#pragma omp parallel
    {
        fftws w;
        w.ar = malloc(sizeof(double) * Mmax); w.ai = malloc(sizeof(double) * Mmax);
        w.br = malloc(sizeof(double) * Mmax); w.bi = malloc(sizeof(double) * Mmax);
        w.cr = malloc(sizeof(double) * nmax); w.ci = malloc(sizeof(double) * nmax);
        double *xr = malloc(sizeof(double) * nmax), *xi = malloc(sizeof(double) * nmax);
        double *Xr = malloc(sizeof(double) * nmax), *Xi = malloc(sizeof(double) * nmax);
        double *blk = malloc(sizeof(double) * 4 * 8 * (size_t)(mmax + 1)); /* [plane][m][lane] of the task's 8 ring pairs */
        double *phr = malloc(sizeof(double) * (mmax + 1)), *phi_ = malloc(sizeof(double) * (mmax + 1));
#pragma omp for schedule(dynamic, 1)
        for (int rv = g->nrp_pad / 8 - 1; rv >= 0; --rv) { /* belt rings (the long ones) first */
            memset(blk, 0, sizeof(double) * 4 * 8 * (size_t)(mmax + 1));
            for (int lane = 0; lane < 8; ++lane) {
                const int i = rv * 8 + lane;
                if (i >= g->nrp) continue;
                const int n = g->nphi[i];
                const double *pn = map + g->startn[i], *ps = g->starts[i] >= 0 ? map + g->starts[i] : NULL;
                const double *wn = pw ? pw + g->startn[i] : NULL, *ws = (pw && ps) ? pw + g->starts[i] : NULL;
                for (int j = 0; j < n; ++j) {
                    xr[j] = wn ? pn[j] * wn[j] : pn[j];
                    xi[j] = ps ? (ws ? ps[j] * ws[j] : ps[j]) : 0.0;
                }
                dft_forward(xr, xi, n, Xr, Xi, t, &w);
                /* phase exp(-i m phi0), phi0 = pi / n on shifted rings: periodic in m with period 2n */
                const int nph = mmax + 1 < 2 * n ? mmax + 1 : 2 * n;
                if (g->shifted[i])
                    for (int q = 0; q < nph; ++q) {
                        const double a = -M_PI * (double)q / n;
                        phr[q] = cos(a);
                        phi_[q] = sin(a);
                    }
                for (int m = 0; m <= mmax; ++m) {
                    const int k = m % n, k2 = (n - k) % n;
                    /* F_N = (X[k] + conj X[n-k]) / 2, F_S = (X[k] - conj X[n-k]) / (2 i) */
                    double fnr = 0.5 * (Xr[k] + Xr[k2]), fni = 0.5 * (Xi[k] - Xi[k2]);
                    double fsr = 0.5 * (Xi[k] + Xi[k2]), fsi = -0.5 * (Xr[k] - Xr[k2]);
                    if (g->shifted[i]) {
                        const int q = m % (2 * n);
                        const double cr = phr[q], ci = phi_[q];
                        double tr = fnr * cr - fni * ci, ti = fnr * ci + fni * cr;
                        fnr = tr; fni = ti;
                        tr = fsr * cr - fsi * ci; ti = fsr * ci + fsi * cr;
                        fsr = tr; fsi = ti;
                    }
                    double *b = blk + (size_t)m * 8 + lane;
                    const size_t ps_ = (size_t)8 * (mmax + 1);
                    b[0] = wpix * (fnr + fsr);
                    b[ps_] = wpix * (fni + fsi);
                    b[2 * ps_] = wpix * (fnr - fsr);
                    b[3 * ps_] = wpix * (fni - fsi);
                }
            }
            for (int p = 0; p < 4; ++p)
                for (int m = 0; m <= mmax; ++m)
                    memcpy(planes + p * plane + (size_t)m * g->nrp_pad + rv * 8, blk + ((size_t)p * (mmax + 1) + m) * 8, sizeof(double) * 8);
        }
        free(w.ar); free(w.ai); free(w.br); free(w.bi); free(w.cr); free(w.ci);
        free(xr); free(xi); free(Xr); free(Xi); free(blk); free(phr); free(phi_);
    }
}

/* ---- Legendre stage ------------------------------------------------------------------------------------------------ */
static inline double hsum(v8 a) { return _mm512_reduce_add_pd(a); }
static inline __mmask8 too_big(v8 a)
{
    return _mm512_cmp_pd_mask(_mm512_abs_pd(a), _mm512_set1_pd(BIG), _CMP_GT_OQ);
}
static inline double fac_of(int e) { return e == 0 ? 1.0 : (e == -1 ? SMALL : 0.0); }

/* per lane: (vc, vp) *= 2^-300, e += 1 where |vc| > 2^300 and e < 0; returns whether any lane of the vector now counts (e >= -1) */
static int rescale(v8 *vc, v8 *vp, int *e, v8 *fac, __mmask8 mk)
{
    double c[8] __attribute__((aligned(64))), p[8] __attribute__((aligned(64))), f[8] __attribute__((aligned(64)));
    _mm512_store_pd(c, *vc);
    _mm512_store_pd(p, *vp);
    int wet = 0;
    for (int k = 0; k < 8; ++k) {
        if (((mk >> k) & 1) && e[k] < 0) {
            c[k] *= SMALL;
            p[k] *= SMALL;
            e[k] += 1;
        }
        f[k] = fac_of(e[k]);
        wet |= e[k] >= -1;
    }
    *vc = _mm512_load_pd(c);
    *vp = _mm512_load_pd(p);
    *fac = _mm512_load_pd(f);
    return wet;
}

#define LB0 128 /* multipoles per block, spin 0: 2 x 128 x 64 B = 16 KiB of sums in L1 */
#define LB2 64  /* spin 2: 4 x 64 x 64 B */

static void legendre0(const geom *g, int lmax, const double *planes, cplx *alm)
{
    const int mmax = lmax, nrp = g->nrp, nrv = g->nrp_pad / 8;
    const size_t plane = (size_t)(mmax + 1) * g->nrp_pad;
    int *mlim = malloc(sizeof(int) * g->nrp_pad);
    for (int i = 0; i < g->nrp_pad; ++i) mlim[i] = i < nrp ? ring_mlim(lmax, 0, g->sth[i], g->z[i]) : -1;
    /* C_m = (-1)^m sqrt((2m+1)/(4 pi)) sqrt(prod_{k<=m} (2k-1)/(2k)), scaled */
    sval *cm = malloc(sizeof(sval) * (mmax + 1));
    {
        sval s = { sqrt(1.0 / (4.0 * M_PI)), 0 };
        cm[0] = s;
        for (int m = 1; m <= mmax; ++m) {
            s = smuld(s, -sqrt((2.0 * m + 1.0) / (2.0 * m)));
            cm[m] = s;
        }
    }
#pragma omp parallel
    {
        v8 *svp = aligned_alloc(64, sizeof(v8) * nrv), *svc = aligned_alloc(64, sizeof(v8) * nrv), *sfac = aligned_alloc(64, sizeof(v8) * nrv);
        int *sex = malloc(sizeof(int) * 8 * nrv), *swet = malloc(sizeof(int) * nrv);
        v8 *accr = aligned_alloc(64, sizeof(v8) * LB0), *acci = aligned_alloc(64, sizeof(v8) * LB0);
        double *ca = malloc(sizeof(double) * (lmax + 3)), *ia = malloc(sizeof(double) * (lmax + 3));
#pragma omp for schedule(dynamic, 1)
        for (int m = 0; m <= mmax; ++m) {
            /* lambda_l = ca[l] (x lambda_{l-1} - ia[l] lambda_{l-2}), ia[l] = 1 / ca[l-1] (0 for l = m + 1) */
            for (int l = m + 1; l <= lmax; ++l) ca[l] = sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
            ia[m + 1] = 0.0;
            for (int l = m + 2; l <= lmax; ++l) ia[l] = 1.0 / ca[l - 1];
            int first = 0;
            while (first < nrp && mlim[first] < m) ++first;
            if (first >= nrp) first = nrp - 1;
            const int rv0 = first / 8;
            for (int rv = rv0; rv < nrv; ++rv) {
                double c[8] __attribute__((aligned(64))), f[8] __attribute__((aligned(64)));
                int wet = 0;
                for (int k = 0; k < 8; ++k) {
                    const int i = rv * 8 + k;
                    sval s = { 0.0, 0 };
                    if (i < nrp) s = smul(cm[m], spow(g->sth[i], m));
                    if (s.e > 0) { s.v = ldexp(s.v, SCB * s.e); s.e = 0; } /* (values are O(1): never) */
                    c[k] = s.v;
                    sex[rv * 8 + k] = s.v == 0.0 ? -1000000 : s.e;
                    f[k] = fac_of(sex[rv * 8 + k]);
                    wet |= sex[rv * 8 + k] >= -1;
                }
                svc[rv] = _mm512_load_pd(c);
                svp[rv] = _mm512_setzero_pd();
                sfac[rv] = _mm512_load_pd(f);
                swet[rv] = wet;
            }
            cplx *out = alm + (int64_t)m * (2 * lmax + 1 - m) / 2;
            for (int lb = m; lb <= lmax; lb += LB0) {
                const int le = lb + LB0 - 1 < lmax ? lb + LB0 - 1 : lmax;
                for (int q = 0; q <= le - lb; ++q) accr[q] = acci[q] = _mm512_setzero_pd();
                /* TWO ring vectors (16 ring pairs) per pass over the block: they share the coefficient broadcasts and -- what matters -- one
                 * load / store of the sums per step (spin 0 has 2 multiply-adds per step and vector to set against them; one vector per pass:
                 * 0.60 TFLOP/s algorithmic on 16 cores, two: see profiles/r06_cpu_baseline_threads.txt).  A vector that does not count yet has
                 * fac = 0 in every lane, so it may ride along. */
                for (int rv = rv0; rv < nrv; rv += 2) {
                    const int rb = rv + 1 < nrv ? rv + 1 : rv; /* (an odd tail: the second vector is all zeros) */
                    const int two = rb != rv;
                    const v8 xa = _mm512_loadu_pd(g->z + rv * 8), xb = _mm512_loadu_pd(g->z + rb * 8);
                    const double *pa = planes + (size_t)m * g->nrp_pad + rv * 8, *pb = planes + (size_t)m * g->nrp_pad + rb * 8;
                    const v8 zero = _mm512_setzero_pd();
                    const v8 aer = _mm512_loadu_pd(pa), aei = _mm512_loadu_pd(pa + plane), aor = _mm512_loadu_pd(pa + 2 * plane), aoi = _mm512_loadu_pd(pa + 3 * plane);
                    const v8 ber = two ? _mm512_loadu_pd(pb) : zero, bei = two ? _mm512_loadu_pd(pb + plane) : zero;
                    const v8 bor = two ? _mm512_loadu_pd(pb + 2 * plane) : zero, boi = two ? _mm512_loadu_pd(pb + 3 * plane) : zero;
                    v8 vpa = svp[rv], vca = svc[rv], fa = sfac[rv], vpb = two ? svp[rb] : zero, vcb = two ? svc[rb] : zero, fb = two ? sfac[rb] : zero;
                    int weta = swet[rv], wetb = two ? swet[rb] : 0;
                    int l = lb;
                    if (lb == m) { /* the seed itself is lambda_mm (even parity) */
                        if (weta | wetb) {
                            const v8 la = _mm512_mul_pd(vca, fa), lbv = _mm512_mul_pd(vcb, fb);
                            accr[0] = _mm512_fmadd_pd(lbv, ber, _mm512_fmadd_pd(la, aer, accr[0]));
                            acci[0] = _mm512_fmadd_pd(lbv, bei, _mm512_fmadd_pd(la, aei, acci[0]));
                        }
                        l = m + 1;
                    }
                    while (l <= le) {
                        const int stop = l + 3 < le ? l + 3 : le; /* four steps, then look at the magnitudes */
                        const int wet = weta | wetb;
                        for (; l <= stop; ++l) {
                            const v8 c = _mm512_set1_pd(ca[l]), ic = _mm512_set1_pd(ia[l]);
                            const v8 na = _mm512_mul_pd(c, _mm512_fnmadd_pd(vpa, ic, _mm512_mul_pd(xa, vca)));
                            const v8 nb = _mm512_mul_pd(c, _mm512_fnmadd_pd(vpb, ic, _mm512_mul_pd(xb, vcb)));
                            vpa = vca; vca = na; vpb = vcb; vcb = nb;
                            if (wet) {
                                const v8 la = _mm512_mul_pd(vca, fa), lbv = _mm512_mul_pd(vcb, fb);
                                const int q = l - lb;
                                if ((l - m) & 1) {
                                    accr[q] = _mm512_fmadd_pd(lbv, bor, _mm512_fmadd_pd(la, aor, accr[q]));
                                    acci[q] = _mm512_fmadd_pd(lbv, boi, _mm512_fmadd_pd(la, aoi, acci[q]));
                                } else {
                                    accr[q] = _mm512_fmadd_pd(lbv, ber, _mm512_fmadd_pd(la, aer, accr[q]));
                                    acci[q] = _mm512_fmadd_pd(lbv, bei, _mm512_fmadd_pd(la, aei, acci[q]));
                                }
                            }
                        }
                        const __mmask8 ka = too_big(vca), kb = too_big(vcb);
                        if (ka) weta = rescale(&vca, &vpa, sex + rv * 8, &fa, ka);
                        if (kb && two) wetb = rescale(&vcb, &vpb, sex + rb * 8, &fb, kb);
                    }
                    svp[rv] = vpa; svc[rv] = vca; sfac[rv] = fa; swet[rv] = weta;
                    if (two) { svp[rb] = vpb; svc[rb] = vcb; sfac[rb] = fb; swet[rb] = wetb; }
                }
                for (int l = lb; l <= le; ++l) out[l] = hsum(accr[l - lb]) + I * hsum(acci[l - lb]);
            }
        }
        free(svp); free(svc); free(sfac); free(sex); free(swet); free(accr); free(acci); free(ca); free(ia);
    }
    free(mlim);
    free(cm);
}

/* spin 2: chains P = N_l d^l_{m,-2}, M = N_l d^l_{m,+2}; f1 = (P + M) / 2, f2 = (P - M) / 2;
 *   even (l + m):  E += f1 (-Qe) + f2 (-i Uo),  B += f1 (-Ue) + f2 (i Qo);   odd: e <-> o
 * planesQ / planesU as ring_stage writes them for the Q and the U map. */
static void legendre2(const geom *g, int lmax, const double *pq, const double *pu, cplx *almE, cplx *almB)
{
    const int mmax = lmax, nrp = g->nrp, nrv = g->nrp_pad / 8;
    const size_t plane = (size_t)(mmax + 1) * g->nrp_pad;
    int *mlim = malloc(sizeof(int) * g->nrp_pad);
    for (int i = 0; i < g->nrp_pad; ++i) mlim[i] = i < nrp ? ring_mlim(lmax, 2, g->sth[i], g->z[i]) : -1;
    /* K_m = sqrt((2m)! / ((m-2)! (m+2)!)) = prod_{k=3..m} sqrt(2k (2k-1) / ((k-2)(k+2))), scaled */
    sval *km = malloc(sizeof(sval) * (mmax + 3));
    {
        sval s = { 1.0, 0 };
        km[0] = km[1] = km[2] = s;
        for (int k = 3; k <= mmax; ++k) {
            s = smuld(s, sqrt((2.0 * k) * (2.0 * k - 1.0) / ((k - 2.0) * (k + 2.0))));
            km[k] = s;
        }
    }
#pragma omp parallel
    {
        v8 *spp = aligned_alloc(64, sizeof(v8) * nrv), *spc = aligned_alloc(64, sizeof(v8) * nrv), *spf = aligned_alloc(64, sizeof(v8) * nrv);
        v8 *smp = aligned_alloc(64, sizeof(v8) * nrv), *smc = aligned_alloc(64, sizeof(v8) * nrv), *smf = aligned_alloc(64, sizeof(v8) * nrv);
        int *pex = malloc(sizeof(int) * 8 * nrv), *mex = malloc(sizeof(int) * 8 * nrv), *swet = malloc(sizeof(int) * nrv);
        v8 *aer = aligned_alloc(64, sizeof(v8) * LB2), *aei = aligned_alloc(64, sizeof(v8) * LB2);
        v8 *abr = aligned_alloc(64, sizeof(v8) * LB2), *abi = aligned_alloc(64, sizeof(v8) * LB2);
        double *k0 = malloc(sizeof(double) * (lmax + 3)), *k1 = malloc(sizeof(double) * (lmax + 3)), *k2 = malloc(sizeof(double) * (lmax + 3));
#pragma omp for schedule(dynamic, 1)
        for (int m = 0; m <= mmax; ++m) {
            const int l0 = m > 2 ? m : 2;
            if (l0 > lmax) continue;
            /* g_{l+1} = (k0[l] x + s k1[l]) g_l - k2[l] g_{l-1}: s = +1 for the (m, -2) chain, -1 for (m, +2) */
            for (int l = l0; l < lmax; ++l) {
                const double dl = l, lp = l + 1.0, dm = m;
                const double den = dl * sqrt((lp * lp - dm * dm) * (lp * lp - 4.0));
                const double r1 = sqrt((2.0 * dl + 3.0) / (2.0 * dl + 1.0));
                k0[l] = r1 * (2.0 * dl + 1.0) * dl * lp / den;
                k1[l] = r1 * (2.0 * dl + 1.0) * dm * 2.0 / den; /* -r1 (2l+1) m n / den with n = -2 */
                k2[l] = l > l0 ? sqrt((2.0 * dl + 3.0) / (2.0 * dl - 1.0)) * lp * sqrt((dl * dl - dm * dm) * (dl * dl - 4.0)) / den : 0.0;
            }
            int first = 0;
            while (first < nrp && mlim[first] < m) ++first;
            if (first >= nrp) first = nrp - 1;
            const int rv0 = first / 8;
            const double nrm = sqrt((2.0 * l0 + 1.0) / (4.0 * M_PI));
            for (int rv = rv0; rv < nrv; ++rv) {
                double cp[8] __attribute__((aligned(64))), cm_[8] __attribute__((aligned(64)));
                double fp[8] __attribute__((aligned(64))), fm[8] __attribute__((aligned(64)));
                int wet = 0;
                for (int k = 0; k < 8; ++k) {
                    const int i = rv * 8 + k;
                    sval sp = { 0.0, 0 }, sm = { 0.0, 0 };
                    if (i < nrp) {
                        const double x = g->z[i], omx = g->omz[i], opx = 2.0 - g->omz[i], sth = g->sth[i]; /* x >= 0 on north rings */
                        (void)x;
                        if (m == 0) {
                            sp.v = sm.v = nrm * sqrt(6.0) / 4.0 * sth * sth;
                        } else if (m == 1) {
                            sp.v = nrm * (-0.5 * omx * sth);
                            sm.v = nrm * (0.5 * opx * sth);
                        } else {
                            sval b = smul(km[m], spow(0.5 * sth, m - 2));
                            const double sg = (m & 1) ? -nrm : nrm;
                            sp = smuld(b, sg * 0.25 * omx * omx);
                            sm = smuld(b, sg * 0.25 * opx * opx);
                        }
                        sp = snorm(sp);
                        sm = snorm(sm);
                    }
                    cp[k] = sp.v; cm_[k] = sm.v;
                    pex[rv * 8 + k] = sp.v == 0.0 ? -1000000 : sp.e;
                    mex[rv * 8 + k] = sm.v == 0.0 ? -1000000 : sm.e;
                    fp[k] = fac_of(pex[rv * 8 + k]);
                    fm[k] = fac_of(mex[rv * 8 + k]);
                    wet |= pex[rv * 8 + k] >= -1 || mex[rv * 8 + k] >= -1;
                }
                spc[rv] = _mm512_load_pd(cp); spp[rv] = _mm512_setzero_pd(); spf[rv] = _mm512_load_pd(fp);
                smc[rv] = _mm512_load_pd(cm_); smp[rv] = _mm512_setzero_pd(); smf[rv] = _mm512_load_pd(fm);
                swet[rv] = wet;
            }
            cplx *oe = almE + (int64_t)m * (2 * lmax + 1 - m) / 2, *ob = almB + (int64_t)m * (2 * lmax + 1 - m) / 2;
            for (int lb = l0; lb <= lmax; lb += LB2) {
                const int le = lb + LB2 - 1 < lmax ? lb + LB2 - 1 : lmax;
                for (int q = 0; q <= le - lb; ++q) aer[q] = aei[q] = abr[q] = abi[q] = _mm512_setzero_pd();
                for (int rv = rv0; rv < nrv; ++rv) {
                    const v8 x = _mm512_loadu_pd(g->z + rv * 8);
                    const double *q_ = pq + (size_t)m * g->nrp_pad + rv * 8, *u_ = pu + (size_t)m * g->nrp_pad + rv * 8;
                    const v8 qer = _mm512_loadu_pd(q_), qei = _mm512_loadu_pd(q_ + plane), qor = _mm512_loadu_pd(q_ + 2 * plane), qoi = _mm512_loadu_pd(q_ + 3 * plane);
                    const v8 uer = _mm512_loadu_pd(u_), uei = _mm512_loadu_pd(u_ + plane), uor = _mm512_loadu_pd(u_ + 2 * plane), uoi = _mm512_loadu_pd(u_ + 3 * plane);
                    v8 pp = spp[rv], pc = spc[rv], pf = spf[rv], mp = smp[rv], mc = smc[rv], mf = smf[rv];
                    int wet = swet[rv];
                    const v8 half = _mm512_set1_pd(0.5);
#define HXF_ACC2(q, ODD)                                                                                                   \
    do {                                                                                                                   \
        const v8 lp_ = _mm512_mul_pd(pc, pf), lm_ = _mm512_mul_pd(mc, mf);                                                  \
        const v8 f1 = _mm512_mul_pd(half, _mm512_add_pd(lp_, lm_)), f2 = _mm512_mul_pd(half, _mm512_sub_pd(lp_, lm_));       \
        if (ODD) { /* E += f1 (-Qo) + f2 (-i Ue), B += f1 (-Uo) + f2 (i Qe) */                                              \
            aer[q] = _mm512_fmadd_pd(f2, uei, _mm512_fnmadd_pd(f1, qor, aer[q]));                                           \
            aei[q] = _mm512_fnmadd_pd(f2, uer, _mm512_fnmadd_pd(f1, qoi, aei[q]));                                          \
            abr[q] = _mm512_fnmadd_pd(f2, qei, _mm512_fnmadd_pd(f1, uor, abr[q]));                                          \
            abi[q] = _mm512_fmadd_pd(f2, qer, _mm512_fnmadd_pd(f1, uoi, abi[q]));                                           \
        } else { /* E += f1 (-Qe) + f2 (-i Uo), B += f1 (-Ue) + f2 (i Qo) */                                                \
            aer[q] = _mm512_fmadd_pd(f2, uoi, _mm512_fnmadd_pd(f1, qer, aer[q]));                                           \
            aei[q] = _mm512_fnmadd_pd(f2, uor, _mm512_fnmadd_pd(f1, qei, aei[q]));                                          \
            abr[q] = _mm512_fnmadd_pd(f2, qoi, _mm512_fnmadd_pd(f1, uer, abr[q]));                                          \
            abi[q] = _mm512_fmadd_pd(f2, qor, _mm512_fnmadd_pd(f1, uei, abi[q]));                                           \
        }                                                                                                                  \
    } while (0)
                    int l = lb;
                    if (lb == l0) {
                        if (wet) {
                            if ((l0 + m) & 1) HXF_ACC2(0, 1); else HXF_ACC2(0, 0);
                        }
                        l = l0 + 1;
                    }
                    while (l <= le) {
                        const int stop = l + 3 < le ? l + 3 : le;
                        for (; l <= stop; ++l) {
                            const v8 a0 = _mm512_mul_pd(_mm512_set1_pd(k0[l - 1]), x), a1 = _mm512_set1_pd(k1[l - 1]), a2 = _mm512_set1_pd(k2[l - 1]);
                            const v8 pn = _mm512_fnmadd_pd(a2, pp, _mm512_mul_pd(_mm512_add_pd(a0, a1), pc));
                            const v8 mn = _mm512_fnmadd_pd(a2, mp, _mm512_mul_pd(_mm512_sub_pd(a0, a1), mc));
                            pp = pc; pc = pn; mp = mc; mc = mn;
                            if (wet) {
                                const int q = l - lb;
                                if ((l + m) & 1) HXF_ACC2(q, 1); else HXF_ACC2(q, 0);
                            }
                        }
                        const __mmask8 kp = too_big(pc), kmm = too_big(mc);
                        if (kp | kmm) {
                            int w1 = rescale(&pc, &pp, pex + rv * 8, &pf, kp);
                            int w2 = rescale(&mc, &mp, mex + rv * 8, &mf, kmm);
                            wet = w1 | w2;
                        }
                    }
#undef HXF_ACC2
                    spp[rv] = pp; spc[rv] = pc; spf[rv] = pf; smp[rv] = mp; smc[rv] = mc; smf[rv] = mf; swet[rv] = wet;
                }
                for (int l = lb; l <= le; ++l) {
                    oe[l] = hsum(aer[l - lb]) + I * hsum(aei[l - lb]);
                    ob[l] = hsum(abr[l - lb]) + I * hsum(abi[l - lb]);
                }
            }
        }
        free(spp); free(spc); free(spf); free(smp); free(smc); free(smf); free(pex); free(mex); free(swet);
        free(aer); free(aei); free(abr); free(abi); free(k0); free(k1); free(k2);
    }
    free(mlim);
    free(km);
}

/* maps [ncomp][npix] RING -> alms [ncomp][nlm] (m-major); spin 0: every component on its own; spin 2: ncomp even, rows (Q, U) ->
 * (E, B).  pix_weights: full-sky array or NULL.  timings (nullable, 2 doubles): seconds of the ring stage and of the Legendre stage.
 * Returns 0, -1 bad arguments, -2 out of memory, -3 no AVX-512 on this CPU. */
int hxf_map2alm(int nside, int lmax, int spin, int ncomp, const double *maps, const double *pix_weights, cplx *alms, double *timings)
{
    if ((spin != 0 && spin != 2) || (spin == 2 && (ncomp & 1)) || nside < 1 || lmax < 0 || ncomp < 1) return -1;
    if (!hxf_supported()) return -3;
    geom g = make_geom(nside);
    const int64_t npix = 12 * (int64_t)nside * nside, nlm = (int64_t)(lmax + 1) * (lmax + 2) / 2;
    const size_t plane = (size_t)(lmax + 1) * g.nrp_pad;
    const int per = spin == 0 ? 1 : 2;
    double *planes = malloc(sizeof(double) * 4 * plane * per);
    if (!planes) { free_geom(&g); return -2; }
    /* first touch in contiguous per-thread chunks: the ring tasks write one cache line per (m, plane) each, so without this every page
     * is faulted in by whichever of ~64 tasks gets there first while the others wait on it (16 threads: 2x SLOWER than 8) */
    {
        const size_t nbytes = sizeof(double) * 4 * plane * per, chunk = (size_t)2 << 20;
        const int64_t nchunks = (int64_t)((nbytes + chunk - 1) / chunk);
#pragma omp parallel for schedule(static)
        for (int64_t c = 0; c < nchunks; ++c) {
            const size_t o = (size_t)c * chunk;
            memset((char *)planes + o, 0, o + chunk <= nbytes ? chunk : nbytes - o);
        }
    }
    int Mmax = 1;
    while (Mmax < 8 * nside - 1) Mmax <<= 1;
    twid t = make_twid(Mmax);
    double tf = 0.0, tl = 0.0;
    memset(alms, 0, sizeof(cplx) * nlm * ncomp);
    for (int c = 0; c < ncomp; c += per) {
        double t0 = wall();
        for (int k = 0; k < per; ++k) ring_stage(&g, lmax, maps + (int64_t)(c + k) * npix, pix_weights, planes + (size_t)k * 4 * plane, &t);
        double t1 = wall();
        if (spin == 0) legendre0(&g, lmax, planes, alms + (int64_t)c * nlm);
        else legendre2(&g, lmax, planes, planes + 4 * plane, alms + (int64_t)c * nlm, alms + (int64_t)(c + 1) * nlm);
        tf += t1 - t0;
        tl += wall() - t1;
    }
    if (timings) { timings[0] = tf; timings[1] = tl; }
    free(t.wr); free(t.wi);
    free(planes);
    free_geom(&g);
    return 0;
}

/* cl[l] = (a_l0 b_l0 + 2 sum_{m=1..l} Re(a_lm conj b_lm)) / (2 l + 1) for two alm sets of the same lmax (heracles/twopoint.py:63-101:
 * the imaginary part of m = 0 is ignored); threads over ranges of m, private sums added in thread order. */
void hxf_alm2cl(const cplx *a, const cplx *b, int lmax, double *cl)
{
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    double *part = calloc((size_t)nt * (lmax + 1), sizeof(double));
#pragma omp parallel
    {
        int tid = 0;
#ifdef _OPENMP
        tid = omp_get_thread_num();
#endif
        double *p = part + (size_t)tid * (lmax + 1);
#pragma omp for schedule(dynamic, 16)
        for (int m = 0; m <= lmax; ++m) {
            const cplx *pa = a + (int64_t)m * (2 * lmax + 1 - m) / 2, *pb = b + (int64_t)m * (2 * lmax + 1 - m) / 2;
            const double w = m == 0 ? 1.0 : 2.0;
            if (m == 0)
                for (int l = 0; l <= lmax; ++l) p[l] += creal(pa[l]) * creal(pb[l]);
            else
                for (int l = m; l <= lmax; ++l) p[l] += w * (creal(pa[l]) * creal(pb[l]) + cimag(pa[l]) * cimag(pb[l]));
        }
    }
    for (int l = 0; l <= lmax; ++l) {
        double s = 0.0;
        for (int t = 0; t < nt; ++t) s += part[(size_t)t * (lmax + 1) + l];
        cl[l] = s / (2.0 * l + 1.0);
    }
    free(part);
}

/* threads of the OpenMP regions from now on (this library and any other in the process that shares the OpenMP runtime) */
void hxf_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}
