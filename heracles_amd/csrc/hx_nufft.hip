// hx_nufft.hip -- adjoint spherical-harmonic synthesis at arbitrary points:
//     a_lm = sum_p v_p conj( sY_lm(theta_p, phi_p) ),    s = 0 (one real value per point) or 2 ((Q, U) -> (E, B)),
// the operation behind heracles.ducc.DiscreteMapper.map_values (heracles/ducc.py:92-133: one call of
// ducc0.sht.adjoint_synthesis_general(map, spin, lmax, loc, epsilon) per catalogue page; ducc0 is a third-party dependency that is
// not part of the reference tree -- what is built here is the published construction of that routine: a type-1 non-uniform FFT
// onto the doubly periodic (theta, phi) torus followed by the adjoint Legendre stage on equidistant rings).
//
// lambda_lm(theta), continued to the full circle, is a trigonometric polynomial of degree <= l with
// lambda_lm(2 pi - theta) = (-1)^m lambda_lm(theta) (both spins: d^l_{m,+-2}(-theta) = (-1)^(m-+2) d^l_{m,+-2}(theta)).  Hence, with
//     G(k, m) = sum_p v_p exp(-i (k theta_p + m phi_p)),  |k| <= lmax, 0 <= m <= lmax          (type-1 NUFFT, 2-D)
//     h_m(theta) = sum_k G(k, m) exp(i k theta),
// Parseval on N > 2 lmax equidistant points theta_j = 2 pi (j + 1/2) / N gives EXACTLY
//     a_lm = (1 / N) sum_{j < N/2} lambda_lm(theta_j) [ h_m(theta_j) + (-1)^m h_m(2 pi - theta_j) ],
// i.e. the Legendre analysis kernel of the HEALPix path (hx_analysis.hip) on N / 2 rings with ring "spectra" h_m and weight
// 1 / N -- no quadrature error, no iteration.  The only approximation is the NUFFT:
//   1. spread: every point adds v_p psi(x - theta_p) psi(y - phi_p) to an n1 x n1 grid over [0, 2 pi)^2, n1 = 2 N >= 2 (2 lmax + 1),
//      psi = "exponential of semicircle" kernel exp(beta (sqrt(1 - (2u/W)^2) - 1)) of W grid cells, beta = 2.3 W
//      (Barnett, Magland & af Klinteberg 2019); W = ceil(log10(1 / epsilon)) + 3 <= 16: 1e-12 -> 15 cells, error 4e-14 at the
//      smallest oversampling that can occur (n1 / (2 lmax + 1) >= 2);
//   2. FFT along phi of the rows the points can touch, divided by psi^(m): T[m][row];
//   3. FFT along theta of every m, divided by psi^(k), shifted by half a cell of the N-grid: U[m][k mod N], |k| <= lmax;
//   4. inverse FFT of length N: h_m(theta_j).
// FFTs: the in-LDS power-of-two kernels of the ring stage (hx_fft_core.h); lengths above 8192 points (128 KiB) as a radix-2 / 4
// decimation-in-frequency step over sub-transforms of 8192.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "hx_sht_common.h"

#include "hx_sort.h"

using namespace hx;
using namespace hxfft;

struct hx_pointsht {
    int lmax = 0, N = 0, n1 = 0, W = 0, twN = 2;
    double beta = 0.0, epsilon = 0.0;
    hx_plan *eq = nullptr;
    hx::DevBuf tw, dec_phi, fac_theta, grid, T, U, h, nbad, key, key2, idx, idx2, sort_tmp;
};

namespace hx {

constexpr int NUFFT_LDS_MAX = 8192;  // points of one in-LDS transform
constexpr int NUFFT_WMAX = 16;

__device__ __host__ inline double es_kernel(double u, double inv_hw, double beta)
{
    const double t = u * inv_hw, a = 1.0 - t * t;
    return a > 0.0 ? exp(beta * (sqrt(a) - 1.0)) : 0.0;
}

// One thread per point: W x W atomic additions per component grid.  loc = (theta, phi) pairs in radians (ducc's `loc`).
__global__ __launch_bounds__(256) void k_nufft_spread(long long npts, const double2 *__restrict__ loc, const double *__restrict__ val,
                                                      double *__restrict__ grid, int n1, int W, double beta,
                                                      unsigned long long *__restrict__ nbad)
{
    const double sc = (double)n1 / (2.0 * M_PI), inv_hw = 2.0 / W, hw = 0.5 * W;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npts; p += (long long)gridDim.x * blockDim.x) {
        const double2 tp = loc[p];
        const double v = val[p];
        if (!(tp.x >= 0.0 && tp.x <= M_PI) || !isfinite(tp.y) || !isfinite(v)) {
            atomicAdd(nbad, 1ULL);
            continue;
        }
        if (v == 0.0) continue;
        const double x = tp.x * sc;
        double ph = fmod(tp.y, 2.0 * M_PI);
        if (ph < 0.0) ph += 2.0 * M_PI;
        const double y = ph * sc;
        const long long i0 = (long long)ceil(x - hw), j0 = (long long)ceil(y - hw);
        double wi[NUFFT_WMAX], wj[NUFFT_WMAX];
#pragma unroll
        for (int a = 0; a < NUFFT_WMAX; ++a) {
            wi[a] = a < W ? v * es_kernel((double)(i0 + a) - x, inv_hw, beta) : 0.0;
            wj[a] = a < W ? es_kernel((double)(j0 + a) - y, inv_hw, beta) : 0.0;
        }
        int jj[NUFFT_WMAX];
#pragma unroll
        for (int b = 0; b < NUFFT_WMAX; ++b) jj[b] = (int)(((j0 + b) % n1 + n1) % n1);
#pragma unroll
        for (int a = 0; a < NUFFT_WMAX; ++a) {
            double *row = grid + (((i0 + a) % n1 + n1) % n1) * (long long)n1;
#pragma unroll
            for (int b = 0; b < NUFFT_WMAX; ++b)
                if (a < W && b < W) unsafeAtomicAdd(row + jj[b], wi[a] * wj[b]);
        }
    }
}

// ---- spreading through LDS tiles (large catalogues) ----------------------------------------------------------------
// The support of a point is the W x W block of cells whose lower corner is (i0, j0) = (ceil(x - W/2), ceil(y - W/2)).  Tile
// (ti, tj) = ((i0 + 16) / 64, (j0 + 16) / 64) owns the points whose corner lies in its 64 x 64 cells; their supports fit a
// (64 + 15)^2 window that the work-group accumulates in LDS (hardware LDS atomics) and adds to the grid once
// ((64 + 15)^2 = 6241 global atomics per occupied tile instead of W^2 per point).  Points are brought into tile order by
// one radix sort of (tile, point index) per call -- the order is shared by all components.
constexpr int NUFFT_TS = 64, NUFFT_TPAD = 16, NUFFT_TW = NUFFT_TS + NUFFT_WMAX - 1, NUFFT_TLD = NUFFT_TS + NUFFT_WMAX;

__device__ inline bool nufft_corner(const double2 tp, int n1, int W, double &x, double &y, long long &i0, long long &j0)
{
    if (!(tp.x >= 0.0 && tp.x <= M_PI) || !isfinite(tp.y)) return false;
    const double sc = (double)n1 / (2.0 * M_PI), hw = 0.5 * W;
    x = tp.x * sc;
    double ph = fmod(tp.y, 2.0 * M_PI);
    if (ph < 0.0) ph += 2.0 * M_PI;
    y = ph * sc;
    i0 = (long long)ceil(x - hw);
    j0 = (long long)ceil(y - hw);
    return true;
}

__global__ __launch_bounds__(256) void k_nufft_keys(long long npts, const double2 *__restrict__ loc, int n1, int W, int ntx,
                                                    unsigned *__restrict__ key, unsigned *__restrict__ idx,
                                                    unsigned long long *__restrict__ nbad)
{
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < npts; p += (long long)gridDim.x * blockDim.x) {
        double x, y;
        long long i0, j0;
        unsigned k = 0xffffffffu;  // invalid points sort behind every tile
        if (nufft_corner(loc[p], n1, W, x, y, i0, j0)) k = (unsigned)((i0 + NUFFT_TPAD) / NUFFT_TS) * (unsigned)ntx + (unsigned)((j0 + NUFFT_TPAD) / NUFFT_TS);
        else atomicAdd(nbad, 1ULL);
        key[p] = k;
        idx[p] = (unsigned)p;
    }
}

__global__ __launch_bounds__(256) void k_nufft_spread_tiles(long long npts, const double2 *__restrict__ loc,
                                                            const double *__restrict__ val, const unsigned *__restrict__ key,
                                                            const unsigned *__restrict__ idx, double *__restrict__ grid, int n1,
                                                            int W, double beta, int ntx, unsigned long long *__restrict__ nbad)
{
    __shared__ double tile[NUFFT_TW * NUFFT_TLD];
    __shared__ long long seg[2];
    const unsigned me = blockIdx.x;
    if (threadIdx.x < 2) {  // first sorted position with key >= me (+ 1)
        const unsigned want = me + threadIdx.x;
        long long lo = 0, hi = npts;
        while (lo < hi) {
            const long long mid = (lo + hi) >> 1;
            if (key[mid] < want) lo = mid + 1;
            else hi = mid;
        }
        seg[threadIdx.x] = lo;
    }
    __syncthreads();
    const long long p0 = seg[0], p1 = seg[1];
    if (p0 == p1) return;
    for (int c = threadIdx.x; c < NUFFT_TW * NUFFT_TLD; c += blockDim.x) tile[c] = 0.0;
    __syncthreads();
    const long long ti0 = (long long)(me / ntx) * NUFFT_TS - NUFFT_TPAD, tj0 = (long long)(me % ntx) * NUFFT_TS - NUFFT_TPAD;
    const double inv_hw = 2.0 / W;
    for (long long q = p0 + threadIdx.x; q < p1; q += blockDim.x) {
        const unsigned p = idx[q];
        const double v = val[p];
        if (!isfinite(v)) {
            atomicAdd(nbad, 1ULL);
            continue;
        }
        if (v == 0.0) continue;
        double x, y;
        long long i0, j0;
        nufft_corner(loc[p], n1, W, x, y, i0, j0);
        double wj[NUFFT_WMAX];
#pragma unroll
        for (int b = 0; b < NUFFT_WMAX; ++b) wj[b] = b < W ? es_kernel((double)(j0 + b) - y, inv_hw, beta) : 0.0;
        double *base = tile + (i0 - ti0) * NUFFT_TLD + (j0 - tj0);
        for (int a = 0; a < W; ++a) {
            const double wa = v * es_kernel((double)(i0 + a) - x, inv_hw, beta);
#pragma unroll
            for (int b = 0; b < NUFFT_WMAX; ++b)
                if (b < W) atomicAdd(base + a * NUFFT_TLD + b, wa * wj[b]);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < NUFFT_TW * NUFFT_TW; c += blockDim.x) {
        const int r = c / NUFFT_TW, cc = c % NUFFT_TW;
        const double v = tile[r * NUFFT_TLD + cc];
        if (v != 0.0) unsafeAtomicAdd(grid + (((ti0 + r) % n1 + n1) % n1) * (long long)n1 + (((tj0 + cc) % n1 + n1) % n1), v);
    }
}

struct NufftFft {
    const double *src_real;   // MODE 0: grid
    const double2 *src;       // MODE 1: T, MODE 2: U
    double2 *dst;             // MODE 0: T, MODE 1: U, MODE 2: h
    const double *dec;        // MODE 0: 1 / psi^(m)
    const double2 *fac;       // MODE 1: exp(i pi k / N) / psi^(k), index k + lmax
    const double2 *tw;
    int n, lmax, N, n1, row0, twN;
};

__device__ inline double2 rot4(double2 x, int k)  // x * (-i)^k
{
    switch (k & 3) {
    case 0: return x;
    case 1: return mul_mi(x);
    case 2: return mk(-x.x, -x.y);
    default: return mul_pi(x);
    }
}

// One work-group per row.  Length n = R ns with ns <= 8192 in LDS: X[R k + r] = FFT_ns( (sum_q x[j + q ns] w_R^{q r}) w_n^{j r} )[k].
// MODE 0: real row (row0 + blockIdx.x) mod n1 of the grid -> T[m][row] = X[m] dec[m], m <= lmax
// MODE 1: row m of T -> U[m][k mod N] = X[k mod n1] fac[k], |k| <= lmax
// MODE 2: row m of U, inverse transform -> h[m][j]
template <int MODE>
__global__ __launch_bounds__(512) void k_nufft_fft(NufftFft a)
{
    extern __shared__ double2 buf[];
    __shared__ double2 tw_hi[TW_HI_MAX], tw_lo[64];
    const TwFactored twf = load_tw_factored(tw_hi, tw_lo, a.tw, a.twN);
    const int n = a.n;
    const int R = n > NUFFT_LDS_MAX ? n / NUFFT_LDS_MAX : 1, ns = n / R, p = ilog2(ns);
    const int row = MODE == 0 ? (a.row0 + (int)blockIdx.x) % a.n1 : (int)blockIdx.x;
    const double *xr = MODE == 0 ? a.src_real + (long long)row * n : nullptr;
    const double2 *xc = MODE == 0 ? nullptr : a.src + (long long)row * n;
    for (int r = 0; r < R; ++r) {
        __syncthreads();  // twiddle tables written / previous sub-transform read out
        for (int j = threadIdx.x; j < ns; j += blockDim.x) {
            double2 t = mk(0.0, 0.0);
            for (int q = 0; q < R; ++q) {
                double2 x;
                if (MODE == 0) x = mk(xr[j + q * ns], 0.0);
                else if (MODE == 1) x = xc[j + q * ns];
                else x = cconj(xc[j + q * ns]);
                t = cadd(t, rot4(x, (4 / R) * q * r));
            }
            if (r && j) t = cmul(t, expipi(-2.0 * (double)(((long long)j * r) % n) / (double)n));
            buf[lds_slot(j)] = t;
        }
        __syncthreads();
        lds_fft_dif(buf, ns, twf, a.twN);
        for (int k = threadIdx.x; k < ns; k += blockDim.x) {
            const int idx = R * k + r;
            const double2 v = buf[lds_slot(bitrev(k, p))];
            if (MODE == 0) {
                if (idx <= a.lmax) a.dst[(long long)idx * a.n1 + row] = cscale(v, a.dec[idx]);
            } else if (MODE == 1) {
                int kk;
                if (idx <= a.lmax) kk = idx;
                else if (idx >= n - a.lmax) kk = idx - n;
                else continue;
                a.dst[(long long)row * a.N + ((kk + a.N) % a.N)] = cmul(v, a.fac[kk + a.lmax]);
            } else {
                a.dst[(long long)row * a.N + idx] = cconj(v);
            }
        }
    }
}

// Gauss-Legendre nodes on [-1, 1] (host, Newton on P_n)
static void gl_nodes(int n, std::vector<double> &x, std::vector<double> &w)
{
    x.resize(n); w.resize(n);
    for (int i = 0; i < (n + 1) / 2; ++i) {
        long double z = cosl(3.141592653589793238462643383279502884L * (i + 0.75L) / (n + 0.5L)), pp = 0.0L;
        for (int it = 0; it < 100; ++it) {
            long double p1 = 1.0L, p2 = 0.0L;
            for (int j = 1; j <= n; ++j) {
                const long double p3 = p2;
                p2 = p1;
                p1 = ((2.0L * j - 1.0L) * z * p2 - (j - 1.0L) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0L);
            const long double dz = p1 / pp;
            z -= dz;
            if (fabsl(dz) < 1e-19L) break;
        }
        x[i] = (double)-z; x[n - 1 - i] = (double)z;
        w[i] = w[n - 1 - i] = (double)(2.0L / ((1.0L - z * z) * pp * pp));
    }
}

}  // namespace hx

extern "C" hx_pointsht *hx_pointsht_create(int lmax, double epsilon)
{
    if (ensure_ready() != HX_OK) return nullptr;
    if (lmax < 0 || lmax > 8191 || !(epsilon > 0.0) || !(epsilon < 1.0)) {
        set_error("hx_pointsht_create: lmax in [0, 8191] and 0 < epsilon < 1");
        return nullptr;
    }
    hx_pointsht *ps = new hx_pointsht;
    ps->lmax = lmax;
    ps->epsilon = epsilon;
    int N = 16;
    while (N < 2 * lmax + 2) N *= 2;
    ps->N = N;
    ps->n1 = 2 * N;
    ps->W = std::max(4, std::min(NUFFT_WMAX, (int)ceil(-log10(epsilon)) + 3));
    ps->beta = 2.30 * ps->W;
    ps->twN = std::min(ps->n1, NUFFT_LDS_MAX);
    ps->eq = plan_create_equiangular(N, lmax);
    if (!ps->eq) { delete ps; return nullptr; }
    // psi^(k) = int psi(u) cos(k h u) du, h = 2 pi / n1, by Gauss-Legendre quadrature (the integrand is smooth; 2 W + 32 nodes)
    std::vector<double> xg, wg;
    gl_nodes(2 * ps->W + 32, xg, wg);
    const double hgrid = 2.0 * M_PI / ps->n1, hw = 0.5 * ps->W;
    auto khat = [&](int k) {
        long double s = 0.0L;
        for (size_t i = 0; i < xg.size(); ++i) s += (long double)(wg[i] * hw * es_kernel(xg[i] * hw, 1.0 / hw, ps->beta)) * cosl((long double)k * hgrid * xg[i] * hw);
        return (double)s;
    };
    std::vector<double> dec(lmax + 1);
    std::vector<double2> fac(2 * lmax + 1), tw(std::max(ps->twN / 2, 64));
    for (int m = 0; m <= lmax; ++m) dec[m] = 1.0 / khat(m);
    for (int k = -lmax; k <= lmax; ++k) {
        const double f = dec[k < 0 ? -k : k], ang = M_PI * k / N;
        fac[k + lmax].x = f * cos(ang); fac[k + lmax].y = f * sin(ang);
    }
    for (int k = 0; k < (int)tw.size(); ++k) {
        const long double ang = -2.0L * 3.141592653589793238462643383279502884L * k / ps->twN;
        tw[k].x = (double)cosl(ang); tw[k].y = (double)sinl(ang);
    }
    if (upload(ps->dec_phi, dec) != HX_OK || upload(ps->fac_theta, fac) != HX_OK || upload(ps->tw, tw) != HX_OK || ps->nbad.alloc(8) != HX_OK) {
        hx_plan_destroy(ps->eq);
        delete ps;
        return nullptr;
    }
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_nufft_fft<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_nufft_fft<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_nufft_fft<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024);
    return ps;
}

extern "C" void hx_pointsht_destroy(hx_pointsht *ps)
{
    if (!ps) return;
    if (rt().ready) (void)hipStreamSynchronize(rt().stream);
    if (ps->eq) hx_plan_destroy(ps->eq);
    delete ps;
}

extern "C" int hx_pointsht_info(const hx_pointsht *ps, int *info4)
{
    if (!ps || !info4) return fail(HX_ERR_ARG, "hx_pointsht_info: null argument");
    info4[0] = ps->lmax; info4[1] = ps->N; info4[2] = ps->n1; info4[3] = ps->W;
    return HX_OK;
}

extern "C" int hx_pointsht_adjoint(hx_pointsht *ps, int spin, int ncomp, int64_t npoints, const double *loc, const double *map,
                                   double *alm)
{
    HX_TRY(ensure_ready());
    if (!ps || !alm || (npoints > 0 && (!loc || !map))) return fail(HX_ERR_ARG, "hx_pointsht_adjoint: null argument");
    if (spin != 0 && spin != 2) return fail(HX_ERR_UNSUPPORTED, "spin-%d values not supported", spin);
    if (ncomp < 1 || (spin == 2 && (ncomp & 1)) || npoints < 0) return fail(HX_ERR_ARG, "hx_pointsht_adjoint: bad component or point count");
    hx_plan *eq = ps->eq;
    const int lmax = ps->lmax, N = ps->N, n1 = ps->n1, W = ps->W;
    InView vloc, vmap;
    OutView valm;
    HX_TRY(vloc.bind(loc, sizeof(double) * 2 * (size_t)npoints));
    HX_TRY(vmap.bind(map, sizeof(double) * (size_t)ncomp * npoints));
    HX_TRY(valm.bind(alm, sizeof(double2) * (size_t)ncomp * eq->nlm));
    hipStream_t st = rt().stream;
    const size_t hrow = (size_t)(lmax + 1) * N;
    HX_TRY(ps->grid.alloc(sizeof(double) * (size_t)n1 * n1));
    HX_TRY(ps->T.alloc(sizeof(double2) * (size_t)(lmax + 1) * n1));
    HX_TRY(ps->U.alloc(sizeof(double2) * hrow));
    const int maxb = analysis_max_comp(spin);
    HX_TRY(ps->h.alloc(sizeof(double2) * hrow * std::min(ncomp, maxb)));
    HX_HIP(hipMemsetAsync(ps->nbad.p, 0, 8, st));
    NufftFft a;
    a.src_real = ps->grid.as<double>(); a.dec = ps->dec_phi.as<double>(); a.fac = ps->fac_theta.as<double2>(); a.tw = ps->tw.as<double2>();
    a.lmax = lmax; a.N = N; a.n1 = n1; a.twN = ps->twN;
    const int band = std::min(n1, n1 / 2 + W + 4);  // rows a point with 0 <= theta <= pi can touch
    a.row0 = (n1 - W / 2 - 2) % n1;
    auto launch = [&](int mode, int n, int rows) {
        const int ns = std::min(n, NUFFT_LDS_MAX), threads = std::min(512, std::max(64, ns / 16));  // one radix-16 butterfly per thread and pass
        a.n = n;
        const size_t lds = (size_t)lds_fft_slots(ns) * sizeof(double2);
        if (mode == 0) hipLaunchKernelGGL(k_nufft_fft<0>, dim3(rows), dim3(threads), lds, st, a);
        else if (mode == 1) hipLaunchKernelGGL(k_nufft_fft<1>, dim3(rows), dim3(threads), lds, st, a);
        else hipLaunchKernelGGL(k_nufft_fft<2>, dim3(rows), dim3(threads), lds, st, a);
    };
    // large catalogues are spread through LDS tiles (one sort per call); HX_NUFFT_TILES = 0 / 1 forces the choice
    bool tiles = npoints >= 300000;  // measured cross-over at lmax 6144 (tools/time_pointsht.py)
    if (const char *e = getenv("HX_NUFFT_TILES")) tiles = atoi(e) != 0;
    if (npoints == 0 || npoints > 0xfffffff0ll) tiles = false;
    const int ntx = n1 / NUFFT_TS + 1, nty = (n1 / 2 + W + NUFFT_TPAD) / NUFFT_TS + 1;
    unsigned *skey = nullptr, *sidx = nullptr;  // the sorted (tile key, point index) pairs
    if (tiles) {
        ProfScope pf("nufft_sort");
        HX_TRY(ps->key.alloc(sizeof(unsigned) * (size_t)npoints));
        HX_TRY(ps->key2.alloc(sizeof(unsigned) * (size_t)npoints));
        HX_TRY(ps->idx.alloc(sizeof(unsigned) * (size_t)npoints));
        HX_TRY(ps->idx2.alloc(sizeof(unsigned) * (size_t)npoints));
        const long long nblk = std::min<long long>((npoints + 255) / 256, 1 << 20);
        hipLaunchKernelGGL(k_nufft_keys, dim3((unsigned)nblk), dim3(256), 0, st, (long long)npoints, vloc.as<double2>(), n1, W, ntx,
                           ps->key.as<unsigned>(), ps->idx.as<unsigned>(), ps->nbad.as<unsigned long long>());
        // all 32 bits: invalid points carry the key 0xffffffff
        HX_TRY(rsort::radix_sort_pairs<unsigned>(ps->key.as<unsigned>(), ps->idx.as<unsigned>(), ps->key2.as<unsigned>(), ps->idx2.as<unsigned>(),
                                                 (unsigned long long)npoints, 32, ps->sort_tmp, st, &skey, &sidx));
    }
    for (int c0 = 0, nb = 0; c0 < ncomp; c0 += nb) {
        nb = analysis_next_batch(spin, ncomp - c0);
        for (int c = 0; c < nb; ++c) {
            {
                ProfScope pf("nufft_spread");
                HX_HIP(hipMemsetAsync(ps->grid.p, 0, sizeof(double) * (size_t)n1 * n1, st));
                if (tiles) {
                    hipLaunchKernelGGL(k_nufft_spread_tiles, dim3((unsigned)(ntx * nty)), dim3(256), 0, st, (long long)npoints,
                                       vloc.as<double2>(), vmap.as<double>() + (size_t)(c0 + c) * npoints, skey, sidx, ps->grid.as<double>(), n1, W, ps->beta, ntx,
                                       ps->nbad.as<unsigned long long>());
                } else if (npoints > 0) {
                    const long long nblk = std::min<long long>((npoints + 255) / 256, 1 << 20);
                    hipLaunchKernelGGL(k_nufft_spread, dim3((unsigned)nblk), dim3(256), 0, st, (long long)npoints, vloc.as<double2>(),
                                       vmap.as<double>() + (size_t)(c0 + c) * npoints, ps->grid.as<double>(), n1, W, ps->beta,
                                       ps->nbad.as<unsigned long long>());
                }
            }
            ProfScope pf("nufft_fft");
            HX_HIP(hipMemsetAsync(ps->T.p, 0, sizeof(double2) * (size_t)(lmax + 1) * n1, st));
            HX_HIP(hipMemsetAsync(ps->U.p, 0, sizeof(double2) * hrow, st));
            a.dst = ps->T.as<double2>();
            launch(0, n1, band);
            a.src = ps->T.as<double2>(); a.dst = ps->U.as<double2>();
            launch(1, n1, lmax + 1);
            a.src = ps->U.as<double2>(); a.dst = ps->h.as<double2>() + hrow * c;
            launch(2, N, lmax + 1);
            HX_HIP(hipGetLastError());
        }
        eq->hsrc = ps->h.as<double2>();
        eq->hsrc_stride = (long long)hrow;
        const int rc = analysis_batch(eq, spin, nb, nullptr, valm.as<double2>() + (size_t)c0 * eq->nlm, nullptr, nullptr, nullptr, 0);
        eq->hsrc = nullptr;
        HX_TRY(rc);
    }
    unsigned long long nbad = 0;
    HX_HIP(hipMemcpyAsync(&nbad, ps->nbad.p, 8, hipMemcpyDeviceToHost, st));
    HX_HIP(hipStreamSynchronize(st));
    if (nbad) return fail(HX_ERR_ARG, "hx_pointsht_adjoint: %llu points with colatitude outside [0, pi] or non-finite values", nbad);
    HX_TRY(valm.finish());
    return HX_OK;
}
