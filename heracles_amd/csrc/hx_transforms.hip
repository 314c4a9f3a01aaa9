// hx_transforms.hip -- Cl <-> xi(theta) transforms on Gauss-Legendre nodes.
//
// Replaces heracles.transforms._cl2corr / _corr2cl (heracles/transforms.py:115-204), i.e.
// the Python loop over lmax+1 nodes that calls legendre_funcs (transforms.py:46-112) once
// per node.  The Legendre / Wigner-d values use the same closed forms in P_l, P_l' as the
// reference (including its small-angle series for d^l_{2,-2} at x > 0.998), so results are
// comparable to rounding.  Tables T[ix][l][k] are built once per lmax (one thread per node,
// coalesced stores), then both directions are dense contractions over l or over k.
#include <algorithm>
#include <cmath>

#include "hx_common.h"

namespace hx {

// T[ix][l][k]: ix 0: P_l, 1: d22, 2: d2m2, 3: d20 (zero for l < 2), k < n, row stride kpad
__global__ void k_corr_tables(int lmax, int n, int kpad, const double *__restrict__ x, double *__restrict__ T)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const long long sl = kpad, st = (long long)(lmax + 1) * kpad;
    const double xx = x[k];
    const double fac1 = 1.0 - xx, fac2 = 1.0 + xx, fac = fac1 / fac2;
    const double sin2 = 1.0 - xx * xx;
    const bool small = xx > 0.998;  // transforms.py:88
    int indser = 0;
    if (small) indser = (int)sqrt((400.0 + 3.0 / (1.0 - xx * xx)) / 150.0) - 1;
    indser = max(0, min(indser, lmax - 1));
    double pm2 = 0.0, pm1 = 1.0;    // P_{l-2}, P_{l-1}
    double dm2 = 0.0, dm1 = 0.0;    // P'_{l-2}, P'_{l-1}
    for (int l = 0; l <= lmax; ++l) {
        double p, dp;
        if (l == 0) { p = 1.0; dp = 0.0; }
        else if (l == 1) { p = xx; dp = 1.0; }
        else {
            p = ((2.0 * l - 1.0) * xx * pm1 - (l - 1.0) * pm2) / l;
            dp = dm2 + (2.0 * l - 1.0) * pm1;
        }
        double d22 = 0.0, d2m2 = 0.0, d20 = 0.0;
        if (l >= 2) {
            const double dl = l, lf = dl * (dl + 1.0), lf2 = (dl + 2.0) * (dl - 1.0);
            d22 = (((4.0 * xx - 8.0) / fac2 + lf) * p + 4.0 * fac * (fac2 + (xx - 2.0) / lf) * dp) / lf2;
            if (small && (l - 2) < indser)
                d2m2 = lf * lf2 * sin2 * sin2 / 7680.0 * (20.0 + sin2 * (16.0 - lf));
            else
                d2m2 = ((lf - (4.0 * xx + 8.0) / fac1) * p + 4.0 / fac * (-fac1 + (xx + 2.0) / lf) * dp) / lf2;
            d20 = (2.0 * xx * dp - lf * p) / sqrt(lf * lf2);
        }
        T[0 * st + l * sl + k] = p;
        T[1 * st + l * sl + k] = d22;
        T[2 * st + l * sl + k] = d2m2;
        T[3 * st + l * sl + k] = d20;
        pm2 = pm1; pm1 = p; dm2 = dm1; dm1 = dp;
    }
}

// corrs[spec][k][ix] = sum_l c_ix[spec][l] T[ix][l][k]; coefficients formed on the fly from cls
__global__ void k_cl2corr(int lmax, int n, int kpad, int nspec, const double *__restrict__ T,
                          const double *__restrict__ cls, double *__restrict__ corrs)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    const int spec = blockIdx.y;
    if (k >= n) return;
    const long long st = (long long)(lmax + 1) * kpad;
    const double *c = cls + (long long)spec * (lmax + 1) * 4;
    double t = 0.0, qp = 0.0, qm = 0.0, cc = 0.0;
    for (int l = 0; l <= lmax; ++l) {
        const double f = (2.0 * l + 1.0) / (4.0 * M_PI);
        const double c0 = c[4 * l], c1 = c[4 * l + 1], c2 = c[4 * l + 2], c3 = c[4 * l + 3];
        t = fma(f * c0, T[(long long)l * kpad + k], t);
        if (l >= 2) {
            qp = fma(f * (c1 + c2), T[st + (long long)l * kpad + k], qp);
            qm = fma(f * (c1 - c2), T[2 * st + (long long)l * kpad + k], qm);
            cc = fma(f * c3, T[3 * st + (long long)l * kpad + k], cc);
        }
    }
    double *o = corrs + ((long long)spec * n + k) * 4;
    o[0] = t; o[1] = qp; o[2] = qm; o[3] = cc;
}

// cls[spec][l][ix] = 2 pi sum_k w_k (...) T[..][l][k]   (transforms.py:194-204)
__global__ __launch_bounds__(256) void k_corr2cl(int lmax, int n, int kpad, const double *__restrict__ T,
                                                 const double *__restrict__ w, const double *__restrict__ corrs,
                                                 double *__restrict__ cls)
{
    __shared__ double red[4][256];
    const int l = blockIdx.x, spec = blockIdx.y;
    const long long st = (long long)(lmax + 1) * kpad;
    const double *cr = corrs + (long long)spec * n * 4;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const double wk = w[k];
        s0 = fma(wk * cr[4 * k], T[(long long)l * kpad + k], s0);
        if (l >= 2) {
            const double T2 = (cr[4 * k + 1] * wk / 2.0) * T[st + (long long)l * kpad + k];
            const double T4 = (cr[4 * k + 2] * wk / 2.0) * T[2 * st + (long long)l * kpad + k];
            s1 += T2 + T4;
            s2 += T2 - T4;
            s3 = fma(wk * cr[4 * k + 3], T[3 * st + (long long)l * kpad + k], s3);
        }
    }
    red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2; red[3][threadIdx.x] = s3;
    __syncthreads();
    for (int h = 128; h >= 1; h >>= 1) {
        if ((int)threadIdx.x < h)
            for (int i = 0; i < 4; ++i) red[i][threadIdx.x] += red[i][threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x < 4) cls[((long long)spec * (lmax + 1) + l) * 4 + threadIdx.x] = 2.0 * M_PI * red[threadIdx.x][0];
}


struct CorrCache {
    int lmax = -1, n = 0, kpad = 0;
    DevBuf x, w, T;
};

static int corr_tables(int lmax, CorrCache &c)
{
    if (c.lmax == lmax && c.T.p) return HX_OK;
    const int n = lmax + 1, kpad = (n + 63) / 64 * 64;
    HX_TRY(c.x.alloc(sizeof(double) * n));
    HX_TRY(c.w.alloc(sizeof(double) * n));
    HX_TRY(c.T.alloc(sizeof(double) * (size_t)4 * (lmax + 1) * kpad));
    hipStream_t st = rt().stream;
    HX_TRY(launch_gauss_legendre(n, c.x.as<double>(), c.w.as<double>()));
    HX_HIP(hipMemsetAsync(c.T.p, 0, sizeof(double) * (size_t)4 * (lmax + 1) * kpad, st));
    {
        ProfScope ps("wigner_tables");
        hipLaunchKernelGGL(k_corr_tables, dim3((n + 63) / 64), dim3(64), 0, st, lmax, n, kpad, c.x.as<double>(), c.T.as<double>());
    }
    HX_HIP(hipGetLastError());
    c.lmax = lmax; c.n = n; c.kpad = kpad;
    return HX_OK;
}

static CorrCache &corr_cache()
{
    static CorrCache c;
    return c;
}

}  // namespace hx

using namespace hx;

extern "C" int hx_cl2corr(int lmax, int nspec, const double *cls, double *corrs)
{
    HX_TRY(ensure_ready());
    if (lmax < 0 || nspec < 1 || !cls || !corrs) return fail(HX_ERR_ARG, "hx_cl2corr: bad argument");
    CorrCache &c = corr_cache();
    HX_TRY(corr_tables(lmax, c));
    const size_t sz = sizeof(double) * (size_t)nspec * (lmax + 1) * 4;
    InView vi;
    OutView vo;
    HX_TRY(vi.bind(cls, sz));
    HX_TRY(vo.bind(corrs, sz));
    hipLaunchKernelGGL(k_cl2corr, dim3((c.n + 63) / 64, nspec), dim3(64), 0, rt().stream, lmax, c.n, c.kpad, nspec,
                       c.T.as<double>(), vi.as<double>(), vo.as<double>());
    HX_HIP(hipGetLastError());
    HX_TRY(vo.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

extern "C" int hx_corr2cl(int lmax, int nspec, const double *corrs, double *cls)
{
    HX_TRY(ensure_ready());
    if (lmax < 0 || nspec < 1 || !cls || !corrs) return fail(HX_ERR_ARG, "hx_corr2cl: bad argument");
    CorrCache &c = corr_cache();
    HX_TRY(corr_tables(lmax, c));
    const size_t sz = sizeof(double) * (size_t)nspec * (lmax + 1) * 4;
    InView vi;
    OutView vo;
    HX_TRY(vi.bind(corrs, sz));
    HX_TRY(vo.bind(cls, sz));
    hipLaunchKernelGGL(k_corr2cl, dim3(lmax + 1, nspec), dim3(256), 0, rt().stream, lmax, c.n, c.kpad, c.T.as<double>(),
                       c.w.as<double>(), vi.as<double>(), vo.as<double>());
    HX_HIP(hipGetLastError());
    HX_TRY(vo.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}
