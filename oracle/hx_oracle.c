/* hx_oracle.c -- CPU ORACLE for the heracles_amd hot path.  TEST INFRASTRUCTURE ONLY.
 * See hx_oracle.h for the parity status of each function.  Nothing under heracles_amd/
 * may call into this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg do.
 *
 * Algorithms are deliberately the "textbook" forms (one-step three-term recursions,
 * incremental seeds, Bluestein FFT), so that agreement with the HIP path -- which uses
 * different recursions (two-step x^2 form, power-by-squaring seeds, LDS FFTs) -- is a
 * meaningful cross-check.
 */
#include "hx_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

typedef double _Complex cplx;

/* cpu_baseline sampling: only every g_mstride-th m is processed in the Legendre stage
 * (outputs of the skipped m stay zero).  1 = full transform (the default, used by tests). */
static int g_mstride = 1;
void hxo_set_mstride(int s) { g_mstride = s > 0 ? s : 1; }
/* wall-clock seconds of the last hxo_map2alm call, split by stage (bench sampling) */
static double g_t_fourier = 0.0, g_t_legendre = 0.0;
void hxo_last_timings(double *t_fourier, double *t_legendre)
{
    *t_fourier = g_t_fourier;
    *t_legendre = g_t_legendre;
}
static double wall(void)
{
#ifdef _OPENMP
    return omp_get_wtime();
#else
    return 0.0;
#endif
}

int hxo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* ------------------------------------------------------------------------------------
 * HEALPix RING geometry (the pixelisation healpy.map2alm works on; reference contract:
 * heracles/healpy.py:117-142,183-189 -- RING-ordered maps of 12*nside^2 pixels).
 * ---------------------------------------------------------------------------------- */
void hxo_ring_info(int nside, int ring, int64_t *startpix, int *nphi, double *z,
                   double *sth, double *phi0)
{
    int64_t ns = nside, npix = 12 * ns * ns, ncap = 2 * ns * (ns - 1);
    int nr = ring > 2 * nside ? 4 * nside - ring : ring; /* mirrored "north" ring */
    double zz, s, fact2 = 4.0 / (double)npix, fact1 = (double)(2 * ns) * fact2;
    int np;
    int64_t sp;
    int shifted;
    if (nr < nside) {
        double tmp = (double)nr * (double)nr * fact2;
        zz = 1.0 - tmp;
        s = sqrt(tmp * (2.0 - tmp));
        np = 4 * nr;
        sp = 2 * (int64_t)nr * (nr - 1);
        shifted = 1;
    } else {
        zz = (double)(2 * nside - nr) * fact1;
        s = sqrt((1.0 - zz) * (1.0 + zz));
        np = 4 * nside;
        sp = ncap + (int64_t)(nr - nside) * 4 * ns;
        shifted = ((nr - nside) & 1) == 0;
    }
    if (nr != ring) {
        zz = -zz;
        sp = npix - sp - np;
    }
    if (startpix) *startpix = sp;
    if (nphi) *nphi = np;
    if (z) *z = zz;
    if (sth) *sth = s;
    if (phi0) *phi0 = shifted ? M_PI / np : 0.0;
}

int64_t hxo_nlm(int lmax) { return (int64_t)(lmax + 1) * (lmax + 2) / 2; }

static inline int64_t almidx(int lmax, int l, int m)
{
    return (int64_t)m * (2 * lmax + 1 - m) / 2 + l;
}

/* ------------------------------------------------------------------------------------
 * FFT: radix-2 for powers of two, Bluestein otherwise.  sign=-1 forward.
 * ---------------------------------------------------------------------------------- */
static void fft_pow2(cplx *a, int n, int sign)
{
    for (int i = 1, j = 0; i < n; ++i) {
        int bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) { cplx t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    for (int len = 2; len <= n; len <<= 1) {
        int half = len >> 1;
        for (int k = 0; k < half; ++k) {
            double ang = sign * 2.0 * M_PI * k / len;
            cplx w = cos(ang) + I * sin(ang);
            for (int i = k; i < n; i += len) {
                cplx u = a[i], v = a[i + half] * w;
                a[i] = u + v;
                a[i + half] = u - v;
            }
        }
    }
}

/* out[k] = sum_j in[j] exp(sign * 2 pi i j k / n), any n >= 1 */
static void dft_any(const cplx *in, cplx *out, int n, int sign)
{
    if ((n & (n - 1)) == 0) {
        memcpy(out, in, sizeof(cplx) * n);
        fft_pow2(out, n, sign);
        return;
    }
    int m = 1;
    while (m < 2 * n - 1) m <<= 1;
    cplx *a = calloc(m, sizeof(cplx)), *b = calloc(m, sizeof(cplx));
    cplx *chirp = malloc(sizeof(cplx) * n);
    for (int j = 0; j < n; ++j) {
        int64_t q = ((int64_t)j * j) % (2 * (int64_t)n);
        double ang = sign * M_PI * (double)q / n;
        chirp[j] = cos(ang) + I * sin(ang); /* exp(sign i pi j^2/n) */
    }
    for (int j = 0; j < n; ++j) a[j] = in[j] * chirp[j];
    b[0] = conj(chirp[0]);
    for (int j = 1; j < n; ++j) b[j] = b[m - j] = conj(chirp[j]);
    fft_pow2(a, m, -1);
    fft_pow2(b, m, -1);
    for (int j = 0; j < m; ++j) a[j] *= b[j];
    fft_pow2(a, m, +1);
    for (int k = 0; k < n; ++k) out[k] = a[k] * chirp[k] / m;
    free(a); free(b); free(chirp);
}

static void dft_direct(const cplx *in, cplx *out, int n, int sign)
{
    for (int k = 0; k < n; ++k) {
        cplx s = 0;
        for (int j = 0; j < n; ++j) {
            int64_t q = ((int64_t)j * k) % n;
            double ang = sign * 2.0 * M_PI * (double)q / n;
            s += in[j] * (cos(ang) + I * sin(ang));
        }
        out[k] = s;
    }
}

/* exp(sign * i * m * phi0) with phi0 = pi/nphi (shifted rings) or 0 */
static inline cplx ring_phase(int m, int nphi, int shifted, int sign)
{
    if (!shifted) return 1.0;
    int64_t r = (int64_t)m % (2 * (int64_t)nphi);
    double ang = sign * M_PI * (double)r / nphi;
    return cos(ang) + I * sin(ang);
}

/* ------------------------------------------------------------------------------------
 * Scaled Legendre / Wigner-d generators.  Values are carried as v * 2^e with e a
 * multiple of SC (<= 0), so that (sin theta)^m for m ~ 10^4 never underflows.
 * ---------------------------------------------------------------------------------- */
#define SC 300
static const double TWO_P = 0x1p+300, TWO_M = 0x1p-300;

typedef struct { double v; int e; } sval;

static inline void snorm(sval *s)
{
    double a = fabs(s->v);
    if (a == 0.0) return;
    while (a < TWO_M) { s->v *= TWO_P; s->e -= SC; a *= TWO_P; }
    while (a > TWO_P) { s->v *= TWO_M; s->e += SC; a *= TWO_M; }
}

static inline double sget(double v, int e) /* true value of v*2^e, 0 if far below range */
{
    if (e == 0) return v;
    if (e < -1200) return 0.0;
    return ldexp(v, e);
}

typedef struct {
    int nside, nrings;
    double *z, *omz /* 1-|z| */, *sth;
    int *nphi, *shifted;
    int64_t *start;
} geom;

static geom make_geom(int nside)
{
    geom g;
    g.nside = nside;
    g.nrings = 4 * nside - 1;
    g.z = malloc(sizeof(double) * g.nrings);
    g.omz = malloc(sizeof(double) * g.nrings);
    g.sth = malloc(sizeof(double) * g.nrings);
    g.nphi = malloc(sizeof(int) * g.nrings);
    g.shifted = malloc(sizeof(int) * g.nrings);
    g.start = malloc(sizeof(int64_t) * g.nrings);
    for (int r = 0; r < g.nrings; ++r) {
        double p0;
        hxo_ring_info(nside, r + 1, &g.start[r], &g.nphi[r], &g.z[r], &g.sth[r], &p0);
        g.shifted[r] = p0 != 0.0;
        int nr = (r + 1) > 2 * nside ? 4 * nside - (r + 1) : (r + 1);
        if (nr < nside)
            g.omz[r] = (double)nr * (double)nr * (4.0 / (12.0 * (double)nside * nside));
        else
            g.omz[r] = 1.0 - fabs(g.z[r]);
    }
    return g;
}

static void free_geom(geom *g)
{
    free(g->z); free(g->omz); free(g->sth); free(g->nphi); free(g->shifted); free(g->start);
}

/* lambda_lm(x) for l = m..lmax at one ring; calls back per l with the true value.
 * lambda_mm = (-1)^m sqrt((2m+1)/(4 pi)) sqrt(prod_{k=1..m} (2k-1)/(2k)) sin^m(theta)     */
typedef struct { double *a; double *ia; } rec0;

static void lam0_seed(int m, double sth, sval *out)
{
    sval s = { sqrt(1.0 / (4.0 * M_PI)), 0 };
    for (int k = 1; k <= m; ++k) {
        s.v *= -sth * sqrt((2.0 * k + 1.0) / (2.0 * k));
        snorm(&s);
    }
    *out = s;
}

/* spin-2 seeds at l0 = max(m,2): values of N_l0 d^{l0}_{m,-2} (-> +2 lambda) and N_l0 d^{l0}_{m,+2} */
static void lam2_seed(int m, double x, double omz, double sth, sval *sp, sval *sm)
{
    /* 1-x and 1+x without cancellation: omz = 1-|x| */
    double omx = x >= 0 ? omz : 2.0 - omz, opx = x >= 0 ? 2.0 - omz : omz;
    int l0 = m > 2 ? m : 2;
    double nrm = sqrt((2.0 * l0 + 1.0) / (4.0 * M_PI));
    if (m == 0) {
        double d = sqrt(6.0) / 4.0 * sth * sth;
        sp->v = nrm * d; sp->e = 0; sm->v = nrm * d; sm->e = 0;
    } else if (m == 1) {
        sp->v = nrm * (-0.5 * omx * sth); sp->e = 0; /* d^2_{1,-2} = d^2_{2,-1} */
        sm->v = nrm * (0.5 * opx * sth);  sm->e = 0; /* d^2_{1, 2} = -d^2_{2,1} */
    } else {
        /* base_m = K_m (cos(t/2) sin(t/2))^{m-2}, K_m = sqrt((2m)!/((m-2)!(m+2)!)) */
        sval b = { 1.0, 0 };
        double cs = 0.5 * sth;
        for (int k = 3; k <= m; ++k) {
            b.v *= cs * sqrt((2.0 * k) * (2.0 * k - 1.0) / ((k - 2.0) * (k + 2.0)));
            snorm(&b);
        }
        double sgn = (m & 1) ? -1.0 : 1.0;
        double s4 = 0.25 * omx * omx, c4 = 0.25 * opx * opx; /* sin^4(t/2), cos^4(t/2) */
        sp->v = sgn * nrm * b.v * s4; sp->e = b.e; /* d^m_{m,-2} */
        sm->v = sgn * nrm * b.v * c4; sm->e = b.e; /* d^m_{m,+2} */
    }
    snorm(sp); snorm(sm);
}

/* one step of the normalised Wigner-d recursion g_{l+1} from g_l, g_{l-1}; n = m' */
static inline void wd_coef(int l, int m, int n, double *c1x, double *c1c, double *c2)
{
    double dl = l, lp = l + 1.0;
    double den = dl * sqrt((lp * lp - (double)m * m) * (lp * lp - (double)n * n));
    double r1 = sqrt((2.0 * dl + 3.0) / (2.0 * dl + 1.0));
    *c1x = r1 * (2.0 * dl + 1.0) * dl * lp / den;
    *c1c = -r1 * (2.0 * dl + 1.0) * (double)m * n / den;
    if (l >= 1) {
        double r2 = sqrt((2.0 * dl + 3.0) / (2.0 * dl - 1.0));
        *c2 = r2 * lp * sqrt((dl * dl - (double)m * m) * (dl * dl - (double)n * n)) / den;
    } else
        *c2 = 0.0;
}

/* ------------------------------------------------------------------------------------
 * Ring Fourier stage
 * ---------------------------------------------------------------------------------- */
/* F[m] = sum_j f_j exp(-i m phi_j) for m=0..mmax on one ring */
static void ring_analyse(const double *pix, const double *pw, int nphi, int shifted,
                         int mmax, int use_fft, cplx *F)
{
    cplx *in = malloc(sizeof(cplx) * nphi), *X = malloc(sizeof(cplx) * nphi);
    for (int j = 0; j < nphi; ++j) in[j] = pw ? pix[j] * pw[j] : pix[j];
    if (use_fft) dft_any(in, X, nphi, -1); else dft_direct(in, X, nphi, -1);
    for (int m = 0; m <= mmax; ++m) F[m] = X[m % nphi] * ring_phase(m, nphi, shifted, -1);
    free(in); free(X);
}

/* f_j = sum_{m>=0} (2-delta_m0) Re(F_m exp(i m phi_j)) */
static void ring_synthesise(const cplx *F, int nphi, int shifted, int mmax, int use_fft,
                            double *pix)
{
    cplx *G = calloc(nphi, sizeof(cplx)), *X = malloc(sizeof(cplx) * nphi);
    for (int m = 0; m <= mmax; ++m)
        G[m % nphi] += (m == 0 ? 1.0 : 2.0) * F[m] * ring_phase(m, nphi, shifted, +1);
    if (use_fft) dft_any(G, X, nphi, +1); else dft_direct(G, X, nphi, +1);
    for (int j = 0; j < nphi; ++j) pix[j] = creal(X[j]);
    free(G); free(X);
}

/* ------------------------------------------------------------------------------------
 * Legendre stage.  F layout: F[(c*nrings + r)*(mmax+1) + m]
 * ---------------------------------------------------------------------------------- */
/* lambda_mm seeds for all (m, north ring), built incrementally in m: O(1) per entry */
typedef struct { double *v; int *e; } seedtab;

static seedtab make_seeds0(const geom *g, int mmax)
{
    int nrp = 2 * g->nside;
    seedtab t;
    t.v = malloc(sizeof(double) * (size_t)(mmax + 1) * nrp);
    t.e = malloc(sizeof(int) * (size_t)(mmax + 1) * nrp);
#pragma omp parallel for schedule(static)
    for (int r = 0; r < nrp; ++r) {
        sval s = { sqrt(1.0 / (4.0 * M_PI)), 0 };
        t.v[r] = s.v; t.e[r] = s.e;
        for (int m = 1; m <= mmax; ++m) {
            s.v *= -g->sth[r] * sqrt((2.0 * m + 1.0) / (2.0 * m));
            snorm(&s);
            t.v[(size_t)m * nrp + r] = s.v; t.e[(size_t)m * nrp + r] = s.e;
        }
    }
    return t;
}
static void free_seeds(seedtab *t) { free(t->v); free(t->e); }

static void legendre_analysis(const geom *g, int lmax, int spin, int ncomp,
                              const cplx *F, const double *rw, cplx *alm, int add)
{
    int nside = g->nside, nr = g->nrings, mmax = lmax;
    int64_t nlm = hxo_nlm(lmax);
    double wpix = 4.0 * M_PI / (12.0 * (double)nside * nside);
    if (!add) memset(alm, 0, sizeof(cplx) * nlm * ncomp);
    seedtab sd = {0, 0};
    if (spin == 0) sd = make_seeds0(g, mmax);
#pragma omp parallel
    {
        cplx *acc = malloc(sizeof(cplx) * (lmax + 1) * ncomp);
        double *ca = malloc(sizeof(double) * (lmax + 2) * 6);
        cplx *fe = malloc(sizeof(cplx) * ncomp * 2), *fo = malloc(sizeof(cplx) * ncomp * 2);
#pragma omp for schedule(dynamic, 1)
        for (int mi = 0; mi <= mmax / g_mstride; ++mi) {
            int m = mi * g_mstride;
            int l0 = spin == 0 ? m : (m > 2 ? m : 2);
            memset(acc, 0, sizeof(cplx) * (lmax + 1) * ncomp);
            if (l0 > lmax) continue;
            /* recursion coefficients for this m */
            if (spin == 0) {
                for (int l = m + 1; l <= lmax; ++l)
                    ca[l] = sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
            } else {
                for (int l = l0; l < lmax; ++l) {
                    wd_coef(l, m, -2, &ca[6 * l], &ca[6 * l + 1], &ca[6 * l + 2]);
                    wd_coef(l, m, +2, &ca[6 * l + 3], &ca[6 * l + 4], &ca[6 * l + 5]);
                }
            }
            for (int r = 0; r < 2 * nside; ++r) { /* north rings incl. equator */
                int rs = nr - 1 - r;              /* southern partner */
                int has_s = rs != r;
                double x = g->z[r], w = wpix * (rw ? rw[r] : 1.0);
                if (spin == 0) {
                    double vp = 0.0, vc = sd.v[(size_t)m * 2 * nside + r];
                    int e = sd.e[(size_t)m * 2 * nside + r];
                    for (int c = 0; c < ncomp; ++c) {
                        cplx fn = F[((int64_t)c * nr + r) * (mmax + 1) + m];
                        cplx fs = has_s ? F[((int64_t)c * nr + rs) * (mmax + 1) + m] : 0.0;
                        fe[c] = w * (fn + fs); fo[c] = w * (fn - fs);
                    }
                    for (int l = m; l <= lmax; ++l) {
                        if (l > m) {
                            double vn = ca[l] * (x * vc - (l - 1 > m ? vp / ca[l - 1] : 0.0));
                            vp = vc; vc = vn;
                            if (e != 0 && fabs(vc) > TWO_P) { vc *= TWO_M; vp *= TWO_M; e += SC; }
                        }
                        double lam = e == 0 ? vc : sget(vc, e);
                        if (lam == 0.0) continue;
                        const cplx *ff = ((l + m) & 1) ? fo : fe;
                        for (int c = 0; c < ncomp; ++c) acc[c * (lmax + 1) + l] += lam * ff[c];
                    }
                } else {
                    sval sp, sm;
                    lam2_seed(m, x, g->omz[r], g->sth[r], &sp, &sm);
                    /* bring both to a common exponent */
                    double pp = 0.0, pc = sp.v, mp = 0.0, mc = sm.v;
                    int ep = sp.e, em = sm.e;
                    /* F1_S = par F1_N, F2_S = -par F2_N: per parity, the operands multiplying
                     * F1 and F2 for E ([c]) and B ([c+1]); fe = even (par=+1), fo = odd */
                    for (int c = 0; c < ncomp; c += 2) {
                        cplx qn = F[((int64_t)c * nr + r) * (mmax + 1) + m];
                        cplx un = F[((int64_t)(c + 1) * nr + r) * (mmax + 1) + m];
                        cplx qs = 0.0, us = 0.0;
                        if (has_s) {
                            qs = F[((int64_t)c * nr + rs) * (mmax + 1) + m];
                            us = F[((int64_t)(c + 1) * nr + rs) * (mmax + 1) + m];
                        }
                        /* E += -(f1 q1 + i f2 u2), B += -(f1 u1 - i f2 q2) */
                        fe[c] = -w * (qn + qs);           fe[ncomp + c] = -w * I * (un - us);
                        fe[c + 1] = -w * (un + us);       fe[ncomp + c + 1] = w * I * (qn - qs);
                        fo[c] = -w * (qn - qs);           fo[ncomp + c] = -w * I * (un + us);
                        fo[c + 1] = -w * (un - us);       fo[ncomp + c + 1] = w * I * (qn + qs);
                    }
                    for (int l = l0; l <= lmax; ++l) {
                        if (l > l0) {
                            const double *k = &ca[6 * (l - 1)];
                            double pn = (k[0] * x + k[1]) * pc - k[2] * pp;
                            double mn = (k[3] * x + k[4]) * mc - k[5] * mp;
                            pp = pc; pc = pn; mp = mc; mc = mn;
                            if (ep != 0 && fabs(pc) > TWO_P) { pc *= TWO_M; pp *= TWO_M; ep += SC; }
                            if (em != 0 && fabs(mc) > TWO_P) { mc *= TWO_M; mp *= TWO_M; em += SC; }
                        }
                        double lp2 = sget(pc, ep), lm2 = sget(mc, em);
                        if (lp2 == 0.0 && lm2 == 0.0) continue;
                        double f1 = 0.5 * (lp2 + lm2), f2 = 0.5 * (lp2 - lm2);
                        const cplx *ff = ((l + m) & 1) ? fo : fe;
                        for (int c = 0; c < ncomp; ++c)
                            acc[c * (lmax + 1) + l] += f1 * ff[c] + f2 * ff[ncomp + c];
                    }
                }
            }
            for (int c = 0; c < ncomp; ++c)
                for (int l = l0; l <= lmax; ++l)
                    alm[c * nlm + almidx(lmax, l, m)] += acc[c * (lmax + 1) + l];
        }
        free(acc); free(ca); free(fe); free(fo);
    }
    if (spin == 0) free_seeds(&sd);
}

static void legendre_synthesis(const geom *g, int lmax, int spin, int ncomp,
                               const cplx *alm, cplx *F)
{
    int nside = g->nside, nr = g->nrings, mmax = lmax;
    int64_t nlm = hxo_nlm(lmax);
#pragma omp parallel
    {
        double *ca = malloc(sizeof(double) * (lmax + 2) * 6);
        cplx *ev = malloc(sizeof(cplx) * ncomp * 2), *od = malloc(sizeof(cplx) * ncomp * 2);
#pragma omp for schedule(dynamic, 1)
        for (int m = 0; m <= mmax; ++m) {
            int l0 = spin == 0 ? m : (m > 2 ? m : 2);
            if (spin == 0) {
                for (int l = m + 1; l <= lmax; ++l)
                    ca[l] = sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
            } else {
                for (int l = l0; l < lmax; ++l) {
                    wd_coef(l, m, -2, &ca[6 * l], &ca[6 * l + 1], &ca[6 * l + 2]);
                    wd_coef(l, m, +2, &ca[6 * l + 3], &ca[6 * l + 4], &ca[6 * l + 5]);
                }
            }
            for (int r = 0; r < 2 * nside; ++r) {
                int rs = nr - 1 - r, has_s = rs != r;
                double x = g->z[r];
                for (int c = 0; c < 2 * ncomp; ++c) ev[c] = od[c] = 0.0;
                if (l0 <= lmax && spin == 0) {
                    sval s;
                    lam0_seed(m, g->sth[r], &s);
                    double vp = 0.0, vc = s.v;
                    int e = s.e;
                    for (int l = m; l <= lmax; ++l) {
                        if (l > m) {
                            double vn = ca[l] * (x * vc - (l - 1 > m ? vp / ca[l - 1] : 0.0));
                            vp = vc; vc = vn;
                            if (fabs(vc) > TWO_P) { vc *= TWO_M; vp *= TWO_M; e += SC; }
                        }
                        double lam = sget(vc, e);
                        if (lam == 0.0) continue;
                        cplx *dst = ((l + m) & 1) ? od : ev;
                        for (int c = 0; c < ncomp; ++c)
                            dst[c] += lam * alm[c * nlm + almidx(lmax, l, m)];
                    }
                    for (int c = 0; c < ncomp; ++c) {
                        F[((int64_t)c * nr + r) * (mmax + 1) + m] = ev[c] + od[c];
                        if (has_s) F[((int64_t)c * nr + rs) * (mmax + 1) + m] = ev[c] - od[c];
                    }
                } else if (l0 <= lmax) {
                    sval sp, sm;
                    lam2_seed(m, x, g->omz[r], g->sth[r], &sp, &sm);
                    double pp = 0.0, pc = sp.v, mp = 0.0, mc = sm.v;
                    int ep = sp.e, em = sm.e;
                    /* ev/od hold, per (Q,U) pair: [c]=sum F1*(..), [ncomp+c] = sum i F2*(..) */
                    for (int l = l0; l <= lmax; ++l) {
                        if (l > l0) {
                            const double *k = &ca[6 * (l - 1)];
                            double pn = (k[0] * x + k[1]) * pc - k[2] * pp;
                            double mn = (k[3] * x + k[4]) * mc - k[5] * mp;
                            pp = pc; pc = pn; mp = mc; mc = mn;
                            if (fabs(pc) > TWO_P) { pc *= TWO_M; pp *= TWO_M; ep += SC; }
                            if (fabs(mc) > TWO_P) { mc *= TWO_M; mp *= TWO_M; em += SC; }
                        }
                        double lp2 = sget(pc, ep), lm2 = sget(mc, em);
                        if (lp2 == 0.0 && lm2 == 0.0) continue;
                        double f1 = 0.5 * (lp2 + lm2), f2 = 0.5 * (lp2 - lm2);
                        cplx *dst = ((l + m) & 1) ? od : ev;
                        for (int c = 0; c < ncomp; c += 2) {
                            cplx E = alm[c * nlm + almidx(lmax, l, m)];
                            cplx B = alm[(c + 1) * nlm + almidx(lmax, l, m)];
                            dst[c] += f1 * E;               /* Q: F1 part */
                            dst[ncomp + c] += I * f2 * B;   /* Q: F2 part */
                            dst[c + 1] += f1 * B;           /* U: F1 part */
                            dst[ncomp + c + 1] += -I * f2 * E; /* U: F2 part */
                        }
                    }
                    for (int c = 0; c < ncomp; ++c) {
                        /* north: F1 and F2 as is; south: F1 -> par F1, F2 -> -par F2 */
                        F[((int64_t)c * nr + r) * (mmax + 1) + m] =
                            -(ev[c] + od[c] + ev[ncomp + c] + od[ncomp + c]);
                        if (has_s)
                            F[((int64_t)c * nr + rs) * (mmax + 1) + m] =
                                -(ev[c] - od[c] - ev[ncomp + c] + od[ncomp + c]);
                    }
                } else {
                    for (int c = 0; c < ncomp; ++c) {
                        F[((int64_t)c * nr + r) * (mmax + 1) + m] = 0.0;
                        if (has_s) F[((int64_t)c * nr + rs) * (mmax + 1) + m] = 0.0;
                    }
                }
            }
        }
        free(ca); free(ev); free(od);
    }
}

static void fourier_analysis(const geom *g, int lmax, int ncomp, const double *maps,
                             const double *pw, int use_fft, cplx *F)
{
    int nr = g->nrings, mmax = lmax;
    int64_t npix = 12 * (int64_t)g->nside * g->nside;
#pragma omp parallel for schedule(dynamic, 4) collapse(2)
    for (int c = 0; c < ncomp; ++c)
        for (int r = 0; r < nr; ++r)
            ring_analyse(maps + c * npix + g->start[r], pw ? pw + g->start[r] : NULL,
                         g->nphi[r], g->shifted[r], mmax, use_fft,
                         F + ((int64_t)c * nr + r) * (mmax + 1));
}

static void fourier_synthesis(const geom *g, int lmax, int ncomp, const cplx *F,
                              int use_fft, double *maps)
{
    int nr = g->nrings, mmax = lmax;
    int64_t npix = 12 * (int64_t)g->nside * g->nside;
#pragma omp parallel for schedule(dynamic, 4) collapse(2)
    for (int c = 0; c < ncomp; ++c)
        for (int r = 0; r < nr; ++r)
            ring_synthesise(F + ((int64_t)c * nr + r) * (mmax + 1), g->nphi[r],
                            g->shifted[r], mmax, use_fft, maps + c * npix + g->start[r]);
}

int hxo_alm2map(int nside, int lmax, int spin, int ncomp, const cplx *alms, double *maps,
                int use_fft)
{
    if ((spin != 0 && spin != 2) || (spin == 2 && (ncomp & 1))) return -1;
    geom g = make_geom(nside);
    cplx *F = malloc(sizeof(cplx) * (size_t)ncomp * g.nrings * (lmax + 1));
    if (!F) { free_geom(&g); return -2; }
    legendre_synthesis(&g, lmax, spin, ncomp, alms, F);
    fourier_synthesis(&g, lmax, ncomp, F, use_fft, maps);
    free(F);
    free_geom(&g);
    return 0;
}

int hxo_map2alm(int nside, int lmax, int spin, int ncomp, const double *maps, cplx *alms,
                const double *ring_weights, const double *pix_weights, int niter,
                int use_fft)
{
    if ((spin != 0 && spin != 2) || (spin == 2 && (ncomp & 1))) return -1;
    geom g = make_geom(nside);
    int64_t npix = 12 * (int64_t)nside * nside;
    cplx *F = malloc(sizeof(cplx) * (size_t)ncomp * g.nrings * (lmax + 1));
    if (!F) { free_geom(&g); return -2; }
    double t0 = wall();
    fourier_analysis(&g, lmax, ncomp, maps, pix_weights, use_fft, F);
    double t1 = wall();
    legendre_analysis(&g, lmax, spin, ncomp, F, ring_weights, alms, 0);
    g_t_fourier = t1 - t0;
    g_t_legendre = wall() - t1;
    if (niter > 0) {
        double *res = malloc(sizeof(double) * npix * ncomp);
        for (int it = 0; it < niter; ++it) {
            legendre_synthesis(&g, lmax, spin, ncomp, alms, F);
            fourier_synthesis(&g, lmax, ncomp, F, use_fft, res);
            for (int64_t i = 0; i < npix * ncomp; ++i) res[i] = maps[i] - res[i];
            fourier_analysis(&g, lmax, ncomp, res, pix_weights, use_fft, F);
            legendre_analysis(&g, lmax, spin, ncomp, F, ring_weights, alms, 1);
        }
        free(res);
    }
    free(F);
    free_geom(&g);
    return 0;
}

/* The two stages of hxo_map2alm on their own (checker of the m-sharded multi-GPU route, tests/test_distributed_cpu.py):
 * F[comp][ring][m] = ring Fourier stage (all 4 nside - 1 rings, m <= lmax); alm = Legendre stage of a given F. */
int hxo_fourier_analysis(int nside, int lmax, int ncomp, const double *maps, const double *pix_weights, cplx *F)
{
    geom g = make_geom(nside);
    fourier_analysis(&g, lmax, ncomp, maps, pix_weights, 1, F);
    free_geom(&g);
    return 0;
}

int hxo_legendre_analysis(int nside, int lmax, int spin, int ncomp, const cplx *F, const double *ring_weights, cplx *alms)
{
    if ((spin != 0 && spin != 2) || (spin == 2 && (ncomp & 1))) return -1;
    geom g = make_geom(nside);
    legendre_analysis(&g, lmax, spin, ncomp, F, ring_weights, alms, 0);
    free_geom(&g);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Adjoint synthesis at arbitrary points -- the operation heracles/ducc.py:121-128 asks of
 * ducc0.sht.adjoint_synthesis_general (ducc0 is third-party and absent from the reference tree; the reference holds
 * no fixture for it): by definition  alm = sum_p v_p conj(sY_lm(theta_p, phi_p)), evaluated here as a DIRECT SUM
 * over the points with the same lambda recursions as the ring transforms above (every point is a ring of one
 * pixel without a southern partner and with weight 1).  O(npoints lmax^2): for the small cases of tests/.
 * spin 0: values[ncomp][npoints] -> alm[ncomp][nlm];  spin 2: ncomp even, rows (Q, U) -> (E, B).
 * ---------------------------------------------------------------------------------- */
int hxo_points2alm(int lmax, int spin, int ncomp, int64_t npoints, const double *theta, const double *phi,
                   const double *values, cplx *alm)
{
    if ((spin != 0 && spin != 2) || (spin == 2 && (ncomp & 1)) || lmax < 0) return -1;
    int64_t nlm = hxo_nlm(lmax);
    memset(alm, 0, sizeof(cplx) * nlm * ncomp);
#pragma omp parallel
    {
        cplx *acc = malloc(sizeof(cplx) * (lmax + 1) * ncomp);
        double *ca = malloc(sizeof(double) * (lmax + 2) * 6);
        cplx *ff = malloc(sizeof(cplx) * ncomp * 2);
#pragma omp for schedule(dynamic, 1)
        for (int mi = 0; mi <= lmax / g_mstride; ++mi) {
            int m = mi * g_mstride;
            int l0 = spin == 0 ? m : (m > 2 ? m : 2);
            if (l0 > lmax) continue;
            memset(acc, 0, sizeof(cplx) * (lmax + 1) * ncomp);
            if (spin == 0) {
                for (int l = m + 1; l <= lmax; ++l)
                    ca[l] = sqrt((4.0 * l * l - 1.0) / ((double)l * l - (double)m * m));
            } else {
                for (int l = l0; l < lmax; ++l) {
                    wd_coef(l, m, -2, &ca[6 * l], &ca[6 * l + 1], &ca[6 * l + 2]);
                    wd_coef(l, m, +2, &ca[6 * l + 3], &ca[6 * l + 4], &ca[6 * l + 5]);
                }
            }
            for (int64_t p = 0; p < npoints; ++p) {
                /* lambda_lm is ill-conditioned in x = cos(theta) near the poles (d lambda / d x ~ l^2): x and the
                 * recursion values are kept in extended precision, so that the result is the transform AT theta_p to
                 * ~1e-12 even a few arc seconds from a pole (the ring transforms above share their x with the device) */
                long double x = cosl((long double)theta[p]);
                double sth = sin(theta[p]);
                double sh = sin(0.5 * theta[p]), ch = cos(0.5 * theta[p]);
                double omz = x >= 0 ? 2.0 * sh * sh : 2.0 * ch * ch; /* 1 - |x| without cancellation */
                cplx ph = cexp(-I * (double)m * phi[p]);
                if (spin == 0) {
                    sval sd;
                    lam0_seed(m, sth, &sd);
                    long double vp = 0.0L, vc = sd.v;
                    int e = sd.e;
                    for (int c = 0; c < ncomp; ++c) ff[c] = values[(int64_t)c * npoints + p] * ph;
                    for (int l = m; l <= lmax; ++l) {
                        if (l > m) {
                            long double vn = ca[l] * (x * vc - (l - 1 > m ? vp / ca[l - 1] : 0.0L));
                            vp = vc; vc = vn;
                            if (e != 0 && fabsl(vc) > TWO_P) { vc *= TWO_M; vp *= TWO_M; e += SC; }
                        }
                        double lam = e == 0 ? (double)vc : sget((double)vc, e);
                        if (lam == 0.0) continue;
                        for (int c = 0; c < ncomp; ++c) acc[c * (lmax + 1) + l] += lam * ff[c];
                    }
                } else {
                    sval sp, sm;
                    lam2_seed(m, (double)x, omz, sth, &sp, &sm);
                    long double pp = 0.0L, pc = sp.v, mp = 0.0L, mc = sm.v;
                    int ep = sp.e, em = sm.e;
                    for (int c = 0; c < ncomp; c += 2) {
                        cplx q = values[(int64_t)c * npoints + p] * ph, u = values[(int64_t)(c + 1) * npoints + p] * ph;
                        /* E += -(f1 q + i f2 u), B += -(f1 u - i f2 q): the northern-ring terms of legendre_analysis */
                        ff[c] = -q;      ff[ncomp + c] = -I * u;
                        ff[c + 1] = -u;  ff[ncomp + c + 1] = I * q;
                    }
                    for (int l = l0; l <= lmax; ++l) {
                        if (l > l0) {
                            const double *k = &ca[6 * (l - 1)];
                            long double pn = (k[0] * x + k[1]) * pc - k[2] * pp;
                            long double mn = (k[3] * x + k[4]) * mc - k[5] * mp;
                            pp = pc; pc = pn; mp = mc; mc = mn;
                            if (ep != 0 && fabsl(pc) > TWO_P) { pc *= TWO_M; pp *= TWO_M; ep += SC; }
                            if (em != 0 && fabsl(mc) > TWO_P) { mc *= TWO_M; mp *= TWO_M; em += SC; }
                        }
                        double lp2 = sget((double)pc, ep), lm2 = sget((double)mc, em);
                        if (lp2 == 0.0 && lm2 == 0.0) continue;
                        double f1 = 0.5 * (lp2 + lm2), f2 = 0.5 * (lp2 - lm2);
                        for (int c = 0; c < ncomp; ++c)
                            acc[c * (lmax + 1) + l] += f1 * ff[c] + f2 * ff[ncomp + c];
                    }
                }
            }
            for (int c = 0; c < ncomp; ++c)
                for (int l = l0; l <= lmax; ++l)
                    alm[c * nlm + almidx(lmax, l, m)] += acc[c * (lmax + 1) + l];
        }
        free(acc); free(ca); free(ff);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------
 * alm2cl -- follows heracles/twopoint.py:63-101.  The reference keeps a running mean
 * over m; the closed form is cl_l = [Re a_l0 Re b_l0 + 2 sum_{m=1..l} Re(a b*)]/(2l+1)
 * (imaginary part of m=0 ignored, twopoint.py:88).  We evaluate the reference's update
 * rule literally so that rounding matches it as closely as a different loop nest can.
 * ---------------------------------------------------------------------------------- */
void hxo_alm2cl(const cplx *alm1, int lmax1, const cplx *alm2, int lmax2, int lmax_out,
                double *cl)
{
    for (int l = 0; l <= lmax_out; ++l) cl[l] = creal(alm1[l]) * creal(alm2[l]);
    for (int m = 1; m <= lmax_out; ++m) {
        int64_t s1 = almidx(lmax1, m, m), s2 = almidx(lmax2, m, m);
        for (int l = m; l <= lmax_out; ++l) {
            cplx a = alm1[s1 + (l - m)], b = alm2[s2 + (l - m)];
            double t = creal(a) * creal(b) + cimag(a) * cimag(b);
            cl[l] += 2.0 * (t - cl[l]) / (2.0 * m + 1.0);
        }
    }
}

/* ------------------------------------------------------------------------------------
 * Gauss-Legendre nodes/weights (what np.polynomial.legendre.leggauss supplies to
 * transforms.py:29-43), Newton iteration on P_n in long double.
 * ---------------------------------------------------------------------------------- */
void hxo_gauss_legendre(int n, double *x, double *w)
{
    for (int i = 0; i < (n + 1) / 2; ++i) {
        long double t = cosl(M_PI * (i + 0.75L) / (n + 0.5L)), dp = 0, p0, p1;
        for (int it = 0; it < 100; ++it) {
            p0 = 1.0L; p1 = t;
            for (int k = 2; k <= n; ++k) {
                long double p2 = ((2 * k - 1) * t * p1 - (k - 1) * p0) / k;
                p0 = p1; p1 = p2;
            }
            if (n == 1) { p1 = t; p0 = 1.0L; }
            dp = n * (t * p1 - p0) / (t * t - 1.0L);
            long double dt = p1 / dp;
            t -= dt;
            if (fabsl(dt) < 1e-19L) break;
        }
        /* re-evaluate derivative at converged node */
        p0 = 1.0L; p1 = t;
        for (int k = 2; k <= n; ++k) {
            long double p2 = ((2 * k - 1) * t * p1 - (k - 1) * p0) / k;
            p0 = p1; p1 = p2;
        }
        dp = n * (t * p1 - p0) / (t * t - 1.0L);
        long double ww = 2.0L / ((1.0L - t * t) * dp * dp);
        x[i] = (double)(-t); x[n - 1 - i] = (double)t;
        w[i] = (double)ww;   w[n - 1 - i] = (double)ww;
    }
    if (n & 1) x[n / 2] = 0.0;
}

/* ------------------------------------------------------------------------------------
 * Wigner d^l_{ab}(x): three-term recursion in l (un-normalised d).
 * ---------------------------------------------------------------------------------- */
void hxo_wigner_d(int lmax, int a, int b, double x, double *out)
{
    int l0 = abs(a) > abs(b) ? abs(a) : abs(b);
    for (int l = 0; l <= lmax; ++l) out[l] = 0.0;
    if (l0 > lmax) return;
    double d0;
    if (a == 0 && b == 0) d0 = 1.0;
    else if (a == 2 && b == 0) d0 = sqrt(6.0) / 4.0 * (1.0 - x) * (1.0 + x);
    else if (a == 2 && b == 2) d0 = 0.25 * (1.0 + x) * (1.0 + x);
    else if (a == 2 && b == -2) d0 = 0.25 * (1.0 - x) * (1.0 - x);
    else if (a == 1 && b == 1) d0 = 0.5 * (1.0 + x);
    else if (a * b == -1) d0 = 0.5 * (1.0 - x);
    else { out[0] = NAN; return; }
    out[l0] = d0;
    double dp = 0.0, dc = d0;
    for (int l = l0; l < lmax; ++l) {
        double dl = l, lp = l + 1.0, dn;
        if (l == 0) dn = x * dc; /* P_1 = x */
        else {
            double den = dl * sqrt((lp * lp - (double)a * a) * (lp * lp - (double)b * b));
            double c1 = (2.0 * dl + 1.0) * (dl * lp * x - (double)a * b) / den;
            double c2 = lp * sqrt((dl * dl - (double)a * a) * (dl * dl - (double)b * b)) / den;
            dn = c1 * dc - c2 * dp;
        }
        dp = dc; dc = dn;
        out[l + 1] = dc;
    }
}

/* ------------------------------------------------------------------------------------
 * legendre_funcs restatement, heracles/transforms.py:46-112.
 * ---------------------------------------------------------------------------------- */
void hxo_legendre_funcs(int lmax, double x, double *P, double *dP, double *d20,
                        double *d22, double *d2m2)
{
    /* P_l and P_l' (scipy.special.legendre_p_all(lmax, x, diff_n=1), transforms.py:60) */
    P[0] = 1.0; dP[0] = 0.0;
    if (lmax >= 1) { P[1] = x; dP[1] = 1.0; }
    for (int l = 2; l <= lmax; ++l) {
        P[l] = ((2.0 * l - 1.0) * x * P[l - 1] - (l - 1.0) * P[l - 2]) / l;
        dP[l] = dP[l - 2] + (2.0 * l - 1.0) * P[l - 1];
    }
    if (lmax < 2) return;
    double fac1 = 1.0 - x, fac2 = 1.0 + x, fac = fac1 / fac2;
    int n = lmax - 1;
    int small = x > 0.998; /* transforms.py:88 */
    int indser = 0;
    if (small) indser = (int)sqrt((400.0 + 3.0 / (1.0 - x * x)) / 150.0) - 1;
    if (indser > n) indser = n;
    if (indser < 0) indser = 0;
    double sin2 = 1.0 - x * x;
    for (int i = 0; i < n; ++i) {
        double l = i + 2.0, lf = l * (l + 1.0), lf2 = (l + 2.0) * (l - 1.0);
        double p = P[i + 2], dp = dP[i + 2];
        d22[i] = (((4.0 * x - 8.0) / fac2 + lf) * p + 4.0 * fac * (fac2 + (x - 2.0) / lf) * dp) / lf2;
        if (small && i < indser)
            d2m2[i] = lf * lf2 * sin2 * sin2 / 7680.0 * (20.0 + sin2 * (16.0 - lf));
        else
            d2m2[i] = ((lf - (4.0 * x + 8.0) / fac1) * p + 4.0 / fac * (-fac1 + (x + 2.0) / lf) * dp) / lf2;
        d20[i] = (2.0 * x * dp - lf * p) / sqrt(lf * lf2);
    }
}

/* _cl2corr, transforms.py:115-161.  cls is (lmax+1, 4) row-major [l][ix]. */
void hxo_cl2corr(int lmax, const double *cls, const double *xw, double *corrs)
{
    int n = lmax + 1;
#pragma omp parallel
    {
        double *P = malloc(sizeof(double) * (lmax + 1) * 5);
        double *dP = P + (lmax + 1), *d20 = dP + (lmax + 1), *d22 = d20 + (lmax + 1),
               *d2m2 = d22 + (lmax + 1);
#pragma omp for
        for (int i = 0; i < n; ++i) {
            hxo_legendre_funcs(lmax, xw[i], P, dP, d20, d22, d2m2);
            double t = 0, qp = 0, qm = 0, cc = 0;
            for (int l = 0; l <= lmax; ++l) {
                double f = (2.0 * l + 1.0) / (4.0 * M_PI);
                t += f * cls[4 * l] * P[l];
                if (l >= 2) {
                    qp += f * (cls[4 * l + 1] + cls[4 * l + 2]) * d22[l - 2];
                    qm += f * (cls[4 * l + 1] - cls[4 * l + 2]) * d2m2[l - 2];
                    cc += f * cls[4 * l + 3] * d20[l - 2];
                }
            }
            corrs[4 * i] = t; corrs[4 * i + 1] = qp; corrs[4 * i + 2] = qm; corrs[4 * i + 3] = cc;
        }
        free(P);
    }
}

/* _corr2cl, transforms.py:164-204. */
void hxo_corr2cl(int lmax, const double *corrs, const double *xw, double *cls)
{
    int n = lmax + 1;
    const double *wts = xw + n;
    memset(cls, 0, sizeof(double) * 4 * (lmax + 1));
    double *P = malloc(sizeof(double) * (lmax + 1) * 5);
    double *dP = P + (lmax + 1), *d20 = dP + (lmax + 1), *d22 = d20 + (lmax + 1),
           *d2m2 = d22 + (lmax + 1);
    for (int i = 0; i < n; ++i) {
        hxo_legendre_funcs(lmax, xw[i], P, dP, d20, d22, d2m2);
        double w = wts[i];
        for (int l = 0; l <= lmax; ++l) {
            cls[4 * l] += (w * corrs[4 * i]) * P[l];
            if (l >= 2) {
                double T2 = (corrs[4 * i + 1] * w / 2.0) * d22[l - 2];
                double T4 = (corrs[4 * i + 2] * w / 2.0) * d2m2[l - 2];
                cls[4 * l + 1] += T2 + T4;
                cls[4 * l + 2] += T2 - T4;
                cls[4 * l + 3] += (w * corrs[4 * i + 3]) * d20[l - 2];
            }
        }
    }
    for (int i = 0; i < 4 * (lmax + 1); ++i) cls[i] *= 2.0 * M_PI;
    free(P);
}

/* ------------------------------------------------------------------------------------
 * Wigner 3j (l1 l2 l3; m1 m2 m3), all l3, Schulten-Gordon two-sided recursion.
 * ---------------------------------------------------------------------------------- */
static inline double sgA(double j, int l1, int l2, int m3)
{
    double d = (double)(l1 - l2), s = (double)(l1 + l2 + 1);
    return sqrt((j * j - d * d) * (s * s - j * j) * (j * j - (double)m3 * m3));
}
static inline double sgB(double j, int l1, int l2, int m1, int m2, int m3)
{
    return -(2.0 * j + 1.0) * ((double)l1 * (l1 + 1.0) * m3 - (double)l2 * (l2 + 1.0) * m3 -
                               j * (j + 1.0) * (double)(m2 - m1));
}

int hxo_wigner3j_l3(int l1, int l2, int m1, int m2, double *out, int *n_out)
{
    int m3 = -(m1 + m2);
    int jmin = abs(l1 - l2), jmax = l1 + l2;
    if (abs(m3) > jmin) jmin = abs(m3);
    int n = jmax - jmin + 1;
    *n_out = n > 0 ? n : 0;
    if (n <= 0 || abs(m1) > l1 || abs(m2) > l2) { *n_out = 0; return jmin; }
    double *f = malloc(sizeof(double) * n), *b = malloc(sizeof(double) * n);
    /* forward from jmin */
    f[0] = 1.0;
    if (n > 1) {
        if (jmin == 0) {
            /* B(0)=0: use the closed forms (l l 0; m -m 0) and (l l 1; m -m 0) ratio */
            /* (l l 1;m -m 0)/(l l 0; m -m 0) = m sqrt(... )  -> 2m/sqrt(2l(2l+2)) * ... */
            double l = l1;
            f[1] = (double)m1 / sqrt(l * (l + 1.0)); /* ratio W(1)/W(0) = m / sqrt(l(l+1)) */
        } else {
            double j = jmin;
            double a1 = sgA(j + 1.0, l1, l2, m3);
            f[1] = a1 != 0.0 ? -sgB(j, l1, l2, m1, m2, m3) * f[0] / (j * a1) : 0.0;
        }
        for (int i = 1; i + 1 < n; ++i) {
            double j = jmin + i;
            double a1 = sgA(j + 1.0, l1, l2, m3);
            f[i + 1] = -(sgB(j, l1, l2, m1, m2, m3) * f[i] + (j + 1.0) * sgA(j, l1, l2, m3) * f[i - 1]) / (j * a1);
            if (fabs(f[i + 1]) > 1e200) for (int k = 0; k <= i + 1; ++k) f[k] *= 1e-200;
        }
    }
    /* backward from jmax */
    b[n - 1] = 1.0;
    if (n > 1) {
        double j = jmax;
        b[n - 2] = -sgB(j, l1, l2, m1, m2, m3) * b[n - 1] / ((j + 1.0) * sgA(j, l1, l2, m3));
        for (int i = n - 2; i >= 1; --i) {
            double jj = jmin + i;
            double a0 = sgA(jj, l1, l2, m3);
            if (a0 == 0.0) { b[i - 1] = 0.0; continue; }
            b[i - 1] = -(sgB(jj, l1, l2, m1, m2, m3) * b[i] + jj * sgA(jj + 1.0, l1, l2, m3) * b[i + 1]) / ((jj + 1.0) * a0);
            if (fabs(b[i - 1]) > 1e200) for (int k = i - 1; k < n; ++k) b[k] *= 1e-200;
        }
    }
    /* match where both are well-conditioned: maximise |f|/max|f| * |b|/max|b| */
    double fm = 0, bm = 0;
    for (int i = 0; i < n; ++i) { if (fabs(f[i]) > fm) fm = fabs(f[i]); if (fabs(b[i]) > bm) bm = fabs(b[i]); }
    int im = 0; double best = -1;
    for (int i = 0; i < n; ++i) {
        double q = (fabs(f[i]) / fm) * (fabs(b[i]) / bm);
        if (q > best) { best = q; im = i; }
    }
    double sc = f[im] / b[im];
    for (int i = 0; i < n; ++i) out[i] = i <= im ? f[i] : b[i] * sc;
    /* normalise: sum (2j+1) W^2 = 1, sign(W(jmax)) = (-1)^(l1-l2-m3) */
    double nrm = 0;
    for (int i = 0; i < n; ++i) nrm += (2.0 * (jmin + i) + 1.0) * out[i] * out[i];
    nrm = 1.0 / sqrt(nrm);
    int sgn = ((l1 - l2 - m3) & 1) ? -1 : 1;
    if ((out[n - 1] < 0 ? -1 : 1) != sgn) nrm = -nrm;
    for (int i = 0; i < n; ++i) out[i] *= nrm;
    free(f); free(b);
    return jmin;
}

/* Any rectangular block l1lo..l1hi x l2lo..l2hi of the matrix: the recursion runs over l3 per (l1, l2), so a block
 * anywhere costs what the low corner costs.  out[(l1hi-l1lo+1)*(l2hi-l2lo+1)], row-major [l1-l1lo][l2-l2lo]. */
void hxo_mixmat_block(const double *cl, int l1lo, int l1hi, int l2lo, int l2hi, int l3max, int s1, int s2, double *out)
{
    int nc = l2hi - l2lo + 1;
#pragma omp parallel
    {
        double *wa = malloc(sizeof(double) * (l1hi + l2hi + 2));
        double *wb = malloc(sizeof(double) * (l1hi + l2hi + 2));
#pragma omp for schedule(dynamic, 1)
        for (int l1 = l1lo; l1 <= l1hi; ++l1)
            for (int l2 = l2lo; l2 <= l2hi; ++l2) {
                double s = 0.0;
                int na, nb;
                if (l1 >= abs(s1) && l2 >= abs(s1) && l1 >= abs(s2) && l2 >= abs(s2)) {
                    int ja = hxo_wigner3j_l3(l1, l2, s1, -s1, wa, &na);
                    int jb = hxo_wigner3j_l3(l1, l2, s2, -s2, wb, &nb);
                    for (int l3 = ja; l3 < ja + na && l3 <= l3max; ++l3) {
                        if (l3 < jb || l3 >= jb + nb) continue;
                        s += (2.0 * l3 + 1.0) * cl[l3] * wa[l3 - ja] * wb[l3 - jb];
                    }
                }
                out[(int64_t)(l1 - l1lo) * nc + (l2 - l2lo)] = (2.0 * l2 + 1.0) / (4.0 * M_PI) * s;
            }
        free(wa); free(wb);
    }
}

void hxo_mixmat(const double *cl, int l1max, int l2max, int l3max, int s1, int s2, double *out)
{
    hxo_mixmat_block(cl, 0, l1max, 0, l2max, l3max, s1, s2, out);
}

/* the same for the three spin-2 x spin-2 matrices: out[3][rows][cols] */
void hxo_mixmat_eb_block(const double *cl, int l1lo, int l1hi, int l2lo, int l2hi, int l3max, double *out)
{
    int nc = l2hi - l2lo + 1;
    int64_t sz = (int64_t)(l1hi - l1lo + 1) * nc;
#pragma omp parallel
    {
        double *wa = malloc(sizeof(double) * (l1hi + l2hi + 2));
#pragma omp for schedule(dynamic, 1)
        for (int l1 = l1lo; l1 <= l1hi; ++l1)
            for (int l2 = l2lo; l2 <= l2hi; ++l2) {
                double se = 0.0, so = 0.0;
                int na;
                if (l1 >= 2 && l2 >= 2) {
                    int ja = hxo_wigner3j_l3(l1, l2, 2, -2, wa, &na);
                    for (int l3 = ja; l3 < ja + na && l3 <= l3max; ++l3) {
                        double t = (2.0 * l3 + 1.0) * cl[l3] * wa[l3 - ja] * wa[l3 - ja];
                        if ((l1 + l2 + l3) & 1) so += t; else se += t;
                    }
                }
                double f = (2.0 * l2 + 1.0) / (4.0 * M_PI);
                int64_t k = (int64_t)(l1 - l1lo) * nc + (l2 - l2lo);
                out[k] = f * se;
                out[sz + k] = f * so;
                out[2 * sz + k] = f * (se - so);
            }
        free(wa);
    }
}

void hxo_mixmat_eb(const double *cl, int l1max, int l2max, int l3max, double *out)
{
    hxo_mixmat_eb_block(cl, 0, l1max, 0, l2max, l3max, out);
}
