/* hx_oracle.h -- CPU ORACLE for the heracles_amd hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the algorithms behind the reference's hot path
 * (SURVEY.md section 8a).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load it; the product path (heracles_amd/) never does.
 *
 * Parity status:
 *  - alm2cl / alm2lmax / legendre_funcs / cl2corr / corr2cl: PINNED against golden
 *    vectors generated from the reference's own numpy code (tests/golden/make_golden.py).
 *  - map2alm / alm2map (third-party healpy, absent from /root/reference and this image):
 *    "parity unpinned" against healpy itself; pinned by first-principles known-answer
 *    tests (scipy Y_lm, explicit spin-weighted Y_lm sums, brute-force direct sums).
 *  - mixmat / mixmat_eb (third-party convolvecl, absent): "parity unpinned" against
 *    convolvecl itself; pinned by exact sympy Wigner-3j known answers and identities.
 */
#ifndef HX_ORACLE_H
#define HX_ORACLE_H
#include <stdint.h>
#include <complex.h>
#ifdef __cplusplus
extern "C" {
#endif

/* HEALPix RING geometry. ring is 1-based, 1..4*nside-1 (north to south). */
void hxo_ring_info(int nside, int ring, int64_t *startpix, int *nphi, double *z,
                   double *sth, double *phi0);

/* alm index, m-major, mmax == lmax: idx(l,m) = m*(2*lmax+1-m)/2 + l  (twopoint.py:90-99) */
int64_t hxo_nlm(int lmax);

/* Analysis: maps -> alm.  spin 0: maps[ncomp][npix] -> alms[ncomp][nlm].
 * spin 2: ncomp must be even, pairs (Q,U) -> (E,B).
 * ring_weights: NULL (=1) or [2*nside] weights for ring pairs i=1..2nside (north half, mirrored).
 * pix_weights : NULL or full [npix] per-pixel weights multiplying the map (healpy use_pixel_weights).
 * niter: number of Jacobi refinement iterations (healpy iter).
 * use_fft: 1 = FFT per ring, 0 = direct DFT (self-check).  Returns 0 on success. */
int hxo_map2alm(int nside, int lmax, int spin, int ncomp, const double *maps,
                double _Complex *alms, const double *ring_weights,
                const double *pix_weights, int niter, int use_fft);

/* Synthesis: alm -> maps (same layouts). */
int hxo_alm2map(int nside, int lmax, int spin, int ncomp, const double _Complex *alms,
                double *maps, int use_fft);

/* Adjoint synthesis at points (heracles/ducc.py:121-128 -> ducc0.sht.adjoint_synthesis_general): direct sum
 * alm = sum_p values_p conj(sY_lm(theta_p, phi_p)); values[ncomp][npoints]; spin 2: rows (Q, U) -> (E, B). */
int hxo_points2alm(int lmax, int spin, int ncomp, int64_t npoints, const double *theta, const double *phi,
                   const double *values, double _Complex *alm);

/* alm2cl (heracles/twopoint.py:63-101): one component pair; lmax1, lmax2 are the alm
 * sizes, lmax_out = number of output multipoles - 1 (<= min(lmax1,lmax2)). */
void hxo_alm2cl(const double _Complex *alm1, int lmax1, const double _Complex *alm2,
                int lmax2, int lmax_out, double *cl);

/* Gauss-Legendre nodes (ascending) and weights on [-1,1]. */
void hxo_gauss_legendre(int n, double *x, double *w);

/* Wigner d^l_{ab}(x) for l=0..lmax at one x=cos(theta), (a,b) in {(0,0),(2,0),(2,2),(2,-2)}
 * via the three-term recursion in l.  out has lmax+1 entries (zeros below l=max(|a|,|b|)). */
void hxo_wigner_d(int lmax, int a, int b, double x, double *out);

/* legendre_funcs restatement (heracles/transforms.py:46-112): closed forms from P_l, P_l'.
 * P[lmax+1], dP[lmax+1]; d20,d22,d2m2 [lmax-1] starting at l=2. */
void hxo_legendre_funcs(int lmax, double x, double *P, double *dP, double *d20,
                        double *d22, double *d2m2);

/* _cl2corr / _corr2cl restatement (transforms.py:115-204): cls (lmax+1,4) <-> corrs (n,4),
 * n = lmax+1 GL nodes (sampling_factor 1). xw = nodes (n) then weights (n). */
void hxo_cl2corr(int lmax, const double *cls, const double *xw, double *corrs);
void hxo_corr2cl(int lmax, const double *corrs, const double *xw, double *cls);

/* Wigner 3j symbols (l1 l2 l3; m1 m2 -(m1+m2)) for all allowed l3 via Schulten-Gordon
 * recursion.  out[l3 - l3min], returns l3min; n_out = number of values. */
int hxo_wigner3j_l3(int l1, int l2, int m1, int m2, double *out, int *n_out);

/* Mixing matrix via 3j recursion (the formula in SURVEY.md 8a-7):
 * out[(l1max+1)*(l2max+1)], row-major [l1][l2]. */
void hxo_mixmat(const double *cl, int l1max, int l2max, int l3max, int s1, int s2, double *out);
/* three matrices: [0] EE->EE, [1] EE->BB, [2] EB->EB */
void hxo_mixmat_eb(const double *cl, int l1max, int l2max, int l3max, double *out);
/* an arbitrary block l1lo..l1hi x l2lo..l2hi of the same matrices (the 3j recursion runs over l3 per (l1, l2):
 * a block at l ~ 6000 costs what the low corner costs); out row-major [l1 - l1lo][l2 - l2lo], _eb: [3][rows][cols] */
void hxo_mixmat_block(const double *cl, int l1lo, int l1hi, int l2lo, int l2hi, int l3max, int s1, int s2, double *out);
void hxo_mixmat_eb_block(const double *cl, int l1lo, int l1hi, int l2lo, int l2hi, int l3max, double *out);

/* the two stages of hxo_map2alm on their own: F[comp][ring][m] (4 nside - 1 rings, m <= lmax), alm from a given F */
int hxo_fourier_analysis(int nside, int lmax, int ncomp, const double *maps, const double *pix_weights, double _Complex *F);
int hxo_legendre_analysis(int nside, int lmax, int spin, int ncomp, const double _Complex *F, const double *ring_weights, double _Complex *alms);

int hxo_num_threads(void);
/* bench sampling: process only every s-th m in the Legendre stage of map2alm (default 1) */
void hxo_set_mstride(int s);
/* stage timings (seconds) of the last hxo_map2alm call (niter = 0 part) */
void hxo_last_timings(double *t_fourier, double *t_legendre);
#ifdef __cplusplus
}
#endif
#endif
