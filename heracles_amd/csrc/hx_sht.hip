// hx_sht.hip -- HEALPix spherical harmonic transforms for gfx950.
//
// Replaces healpy.map2alm / alm2map as called from heracles/healpy.py:183-189 (spin 0 and
// spin 2, RING-ordered maps, mmax == lmax).
//
// Pipeline of one analysis batch (<= 8 map components):
//   1. k_ring_subdft       ring Fourier stage.  A north/south ring pair is packed as
//                          z = f_N + i f_S and transformed as one complex DFT of length
//                          4n, split radix-4 (DIF) into four length-n DFTs that run
//                          entirely in LDS (plain FFT for n = 2^k, Bluestein otherwise).
//   2. k_fourier_combine   un-packs N/S, applies ring phase / quadrature weight, forms
//                          the parity combinations and writes the MFMA B-operand layout
//                          F[m][ring pair][parity][op][16 columns].
//   3. k_legendre_analysis Legendre / Wigner-d stage: lanes = ring pairs run the
//                          three-term recursion in l; 16 l-values x 64 rings of
//                          lambda_lm are transposed through LDS into A-operand tiles of
//                          v_mfma_f64_16x16x4_f64, which contracts over rings against the
//                          F operands of 8 maps (16 real columns) held in registers.
//   4. k_alm_reduce        fixed-order sum of the ring-group partials -> alm (x fl).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "hx_common.h"
#include "hx_fft_core.h"

namespace hx {
using namespace hxfft;

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int LA_WAVES = 8;       // waves (64-ring-pair blocks) per Legendre workgroup
constexpr int TILE_LD = 64;       // LDS row stride (doubles) of a 16 x 64 lambda tile (XOR-swizzled columns)
constexpr int LBLK = 32;          // l values per block (16 even-parity + 16 odd-parity rows)
constexpr int NCOL = 16;          // MFMA N: real columns per batch (8 spin-0 maps / 4 spin-2 fields)
constexpr double SC_BIG = 0x1p+300, SC_SMALL = 0x1p-300;

struct LegTask {
    int m;
    int rb0;          // first 64-ring-pair block
    int nrb;          // blocks (waves) used, 1..LA_WAVES
    int pad;
    long long pout;   // first row of this task in the partial buffer
};

struct MTasks {
    int first, count;
};

// Device-side view of a plan (POD, passed by value to kernels).
struct PlanDev {
    int nside, lmax, nrp, nrp_pad, twN;
    long long npix, ny;
    const double *z, *omz, *sth, *rwdef;
    const int *nsub, *shifted;
    const long long *startN, *startS, *bhat_off;
    const double2 *tw, *bhat;
    const double *mfac, *kfac2;
    const double2 *rec0;
    const double4 *rec2;
};

}  // namespace hx

struct hx_plan {
    int nside = 0, lmax = 0, max_comp = 0;
    int nrp = 0, nrp_pad = 0, nrb = 0, twN = 1;
    long long npix = 0, ny = 0, nlm = 0;
    size_t lds_fft = 0;
    hx::DevBuf z, omz, sth, rwdef, nsub, shifted, startN, startS, bhat_off, tw, bhat, mfac, kfac2, rec0, rec2, cn0, al0, cn2, al2;
    std::vector<double> h_sth, h_z;
    std::vector<int> h_nsub;
    struct TaskSet {
        bool built = false;
        std::vector<hx::LegTask> tasks;
        std::vector<hx::MTasks> of_m;
        hx::DevBuf d_tasks, d_of_m;
        long long rows = 0;
    } ts[2];
    hx::DevBuf Y, F, partial, d_rw, stage_maps, stage_alms, resid, Fsyn;
    hx::PlanDev dev() const;
};

namespace hx {

__host__ __device__ inline long long almidx(int lmax, int l, int m)
{
    return (long long)m * (2 * lmax + 1 - m) / 2 + l;
}

// =====================================================================================
// table initialisation kernels
// =====================================================================================
__global__ void k_init_rec0(int lmax, double2 *__restrict__ rec)
{
    const int m = blockIdx.x;
    for (int l = m + threadIdx.x; l <= lmax; l += blockDim.x) {
        double2 r = make_double2(0.0, 0.0);
        if (l > m) {
            double dl = l, dm = m;
            double a = sqrt((4.0 * dl * dl - 1.0) / (dl * dl - dm * dm));
            double b = 0.0;
            if (l > m + 1) {
                double d1 = l - 1.0;
                double ap = sqrt((4.0 * d1 * d1 - 1.0) / (d1 * d1 - dm * dm));
                b = a / ap;
            }
            r = make_double2(a, b);
        }
        rec[almidx(lmax, l, m)] = r;
    }
}

// Normalised recursions used by the analysis kernel (two FMAs per new value, after the
// scheme of libsharp/ducc's Ylmgen): with lambda_l = alpha_l mu_l,
//   spin 0 (two-step):  mu_{l+2} = (A' x^2 + B') mu_l - mu_{l-2}
//       A = a_{l+1} a_{l+2},  B = -(a_{l+2}/a_{l+1} + a_{l+2} a_{l+1}/a_l^2),  a_l = sqrt((4l^2-1)/(l^2-m^2)),
//       alpha_{l+2} alpha_l = a_{l+2} a_{l+1} / 4,   A' = A alpha_l/alpha_{l+2},  B' likewise;
//       coef[idx(l,m)] = (A', B') is indexed by the SOURCE l, alpha[idx(l,m)] = alpha_l.
//   spin 2 (one-step):  mu_{l+1} = (p' x +- q') mu_l - mu_{l-1}   (+ for d^l_{m,-2}, - for d^l_{m,+2})
//       alpha_{l+1} = r_l alpha_{l-1},  p' = p alpha_l/alpha_{l+1},  q' likewise;
//       coef[idx(l+1,m)] = (p', q') is indexed by the TARGET l.
// One thread per (m, chain): the alpha recursion is sequential in l.
__global__ void k_init_norm0(int lmax, double2 *__restrict__ coef, double *__restrict__ alpha)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = t >> 1, par = t & 1;
    if (m > lmax) return;
    const double dm = m;
    auto a = [dm](double l) { return sqrt((4.0 * l * l - 1.0) / (l * l - dm * dm)); };
    double al = 1.0;
    for (int l = m + par; l <= lmax; l += 2) {
        const double dl = l;
        const double a1 = a(dl + 1.0), a2 = a(dl + 2.0);
        const double A = a1 * a2;
        double B = -a2 / a1;
        if (l > m) {
            const double a0 = a(dl);
            B -= a2 * a1 / (a0 * a0);
        }
        const double an = 0.25 * a2 * a1 / al;
        coef[almidx(lmax, l, m)] = make_double2(A * al / an, B * al / an);
        alpha[almidx(lmax, l, m)] = al;
        al = an;
    }
}

__global__ void k_init_norm2(int lmax, double2 *__restrict__ coef, double *__restrict__ alpha)
{
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m > lmax) return;
    const int l0 = m > 2 ? m : 2;
    double am1 = 1.0, a0 = 1.0;  // alpha_{l-1}, alpha_l
    for (int l = l0; l <= lmax; ++l) {
        alpha[almidx(lmax, l, m)] = a0;
        if (l == lmax) break;
        // coefficients of the step l -> l+1 (n = -2)
        const double k = l, lp = l + 1.0, dm = m, dn = -2.0;
        const double den = k * sqrt((lp * lp - dm * dm) * (lp * lp - dn * dn));
        const double r1 = sqrt((2.0 * k + 3.0) / (2.0 * k + 1.0));
        const double p = r1 * (2.0 * k + 1.0) * k * lp / den;
        const double q = -r1 * (2.0 * k + 1.0) * dm * dn / den;
        double a1 = 1.0;
        if (l > l0) {
            const double r2 = sqrt((2.0 * k + 3.0) / (2.0 * k - 1.0));
            const double r = r2 * lp * sqrt((k * k - dm * dm) * (k * k - dn * dn)) / den;
            a1 = r * am1;
        }
        coef[almidx(lmax, l + 1, m)] = make_double2(p * a0 / a1, q * a0 / a1);
        am1 = a0;
        a0 = a1;
    }
}

// rec2[idx(l,m)] = (c1x, c1c(n=-2), c2, 0): g_l = (c1x x + c1c) g_{l-1} - c2 g_{l-2};
// for n=+2 the sign of c1c flips.  g_l = sqrt((2l+1)/4pi) d^l_{m,n}.
__global__ void k_init_rec2(int lmax, double4 *__restrict__ rec)
{
    const int m = blockIdx.x;
    const int l0 = m > 2 ? m : 2;
    for (int l = m + threadIdx.x; l <= lmax; l += blockDim.x) {
        double4 r = make_double4(0.0, 0.0, 0.0, 0.0);
        if (l > l0) {
            double k = l - 1.0, lp = l, dm = m, dn = -2.0;
            double den = k * sqrt((lp * lp - dm * dm) * (lp * lp - dn * dn));
            double r1 = sqrt((2.0 * k + 3.0) / (2.0 * k + 1.0));
            r.x = r1 * (2.0 * k + 1.0) * k * lp / den;
            r.y = -r1 * (2.0 * k + 1.0) * dm * dn / den;
            double r2 = sqrt((2.0 * k + 3.0) / (2.0 * k - 1.0));
            r.z = r2 * lp * sqrt((k * k - dm * dm) * (k * k - dn * dn)) / den;
        }
        rec[almidx(lmax, l, m)] = r;
    }
}

__device__ inline double2 expipi(double x)
{
    double s, c;
    sincospi(x, &s, &c);
    return make_double2(c, s);
}

// In-LDS FFT drivers (all threads of the block participate)
__device__ inline void lds_fft_dif(double2 *buf, int M, const double2 *__restrict__ tw, int twN)
{
    for (int h = M >> 1; h >= 1; h >>= 1) {
        for (int i = threadIdx.x; i < (M >> 1); i += blockDim.x) dif_butterfly(buf, i, h, tw, twN);
        __syncthreads();
    }
}
__device__ inline void lds_fft_dit_inv(double2 *buf, int M, const double2 *__restrict__ tw, int twN)
{
    for (int h = 1; h <= (M >> 1); h <<= 1) {
        for (int i = threadIdx.x; i < (M >> 1); i += blockDim.x) dit_inv_butterfly(buf, i, h, tw, twN);
        __syncthreads();
    }
}

// Bluestein filter spectra, one block per ring pair whose sub-length is not a power of two
// and is the first ring with that length.
__global__ __launch_bounds__(512) void k_init_bhat(PlanDev P, const int *__restrict__ rp_list,
                                                   double2 *__restrict__ bhat)
{
    extern __shared__ double2 buf[];
    const int rp = rp_list[blockIdx.x];
    const int n = P.nsub[rp];
    const int M = fft_size_for(n);
    for (int j = threadIdx.x; j < M; j += blockDim.x) buf[j] = make_double2(0.0, 0.0);
    __syncthreads();
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        double2 c = expipi((double)chirp_num(j, n) / (double)n);
        buf[j] = c;
        if (j) buf[M - j] = c;
    }
    __syncthreads();
    lds_fft_dif(buf, M, P.tw, P.twN);
    double2 *out = bhat + P.bhat_off[rp];
    for (int j = threadIdx.x; j < M; j += blockDim.x) out[j] = buf[j];
}

// =====================================================================================
// 1. ring Fourier stage: sub-DFTs in LDS
// =====================================================================================
// MODE 0: input = real maps (N ring -> real part, S ring -> imaginary part)
// MODE 1: input = complex spectrum Zc[c][ny-layout natural order] (synthesis: conj trick)
template <int MODE>
__global__ __launch_bounds__(512) void k_ring_subdft(PlanDev P, const double *__restrict__ maps,
                                                     const double *__restrict__ pixw,
                                                     const double2 *__restrict__ zin,
                                                     double2 *__restrict__ Y)
{
    extern __shared__ double2 buf[];
    const int rp = P.nrp - 1 - (int)blockIdx.x;  // long (equatorial) rings first
    const int r = blockIdx.y, c = blockIdx.z;
    const int n = P.nsub[rp];
    const long long sN = P.startN[rp], sS = P.startS[rp];
    const int M = fft_size_for(n);
    const bool blu = M != n;
    const int p = ilog2(M);
    const double *mp = maps + (long long)c * P.npix;
    const double2 *zp = zin + (long long)c * P.ny + sN;

    for (int j = threadIdx.x; j < M; j += blockDim.x) {
        double2 val = make_double2(0.0, 0.0);
        if (j < n) {
            double2 zq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (MODE == 0) {
                    long long iN = sN + j + (long long)q * n;
                    double fn = mp[iN];
                    if (pixw) fn *= pixw[iN];
                    double fs = 0.0;
                    if (sS >= 0) {
                        long long iS = sS + j + (long long)q * n;
                        fs = mp[iS];
                        if (pixw) fs *= pixw[iS];
                    }
                    zq[q] = make_double2(fn, fs);
                } else {
                    zq[q] = zp[j + (long long)q * n];
                }
            }
            double2 t = dif4_combine(zq[0], zq[1], zq[2], zq[3], r);
            unsigned qn = load_phase_num(j, r, n, blu);
            val = qn ? cmul(t, expipi(-(double)qn / (2.0 * n))) : t;
        }
        buf[j] = val;
    }
    __syncthreads();
    lds_fft_dif(buf, M, P.tw, P.twN);
    double2 *out = Y + (long long)c * P.ny + sN + (long long)r * n;
    if (!blu) {
        for (int k = threadIdx.x; k < n; k += blockDim.x) out[k] = buf[bitrev(k, p)];
        return;
    }
    const double2 *bh = P.bhat + P.bhat_off[rp];
    for (int j = threadIdx.x; j < M; j += blockDim.x) buf[j] = cmul(buf[j], bh[j]);
    __syncthreads();
    lds_fft_dit_inv(buf, M, P.tw, P.twN);
    const double inv = 1.0 / M;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        double2 cz = expipi(-(double)chirp_num(k, n) / (double)n);
        out[k] = cscale(cmul(buf[k], cz), inv);
    }
}

// =====================================================================================
// 2. combine: Y -> F operands
// =====================================================================================
__device__ inline double2 ring_mode(const PlanDev &P, const double2 *__restrict__ Yc, int rp, int n,
                                    int mm)
{
    // Z[mm] with X[4k+r] = Y_r[k]
    return Yc[P.startN[rp] + (long long)(mm & 3) * n + (mm >> 2)];
}

// F_N(m), F_S(m) of ring pair rp for component c, including phase and quadrature weight
__device__ inline void ring_modes_ns(const PlanDev &P, const double2 *__restrict__ Y, int c, int rp,
                                     int m, double w, double2 &FN, double2 &FS)
{
    const int n = P.nsub[rp];
    const int nphi = 4 * n;
    const int mm = m % nphi, mc = (nphi - mm) % nphi;
    const double2 *Yc = Y + (long long)c * P.ny;
    const double2 a = ring_mode(P, Yc, rp, n, mm), b = cconj(ring_mode(P, Yc, rp, n, mc));
    double2 xn = cscale(cadd(a, b), 0.5);
    double2 d = cscale(csub(a, b), 0.5);
    double2 xs = mul_mi(d);  // (a-b)/(2i)
    double2 ph = make_double2(w, 0.0);
    if (P.shifted[rp]) ph = cscale(expipi(-(double)(m % (2 * nphi)) / (double)nphi), w);
    FN = cmul(xn, ph);
    FS = P.startS[rp] >= 0 ? cmul(xs, ph) : make_double2(0.0, 0.0);
}

// grid: x = m, y = tiles of 32 ring pairs; block 256 = 32 ring pairs x 8 slots
template <int SPIN>
__global__ __launch_bounds__(256) void k_fourier_combine(PlanDev P, const double2 *__restrict__ Y,
                                                         int ncomp, const double *__restrict__ rw,
                                                         double *__restrict__ F)
{
    const int m = blockIdx.x;
    const int rp = blockIdx.y * 32 + (threadIdx.x >> 3);
    const int slot = threadIdx.x & 7;
    if (rp >= P.nrp_pad) return;
    const bool live = rp < P.nrp;
    const double w = live ? (rw ? rw[rp] : 1.0) * (4.0 * M_PI / (double)P.npix) : 0.0;
    if (SPIN == 0) {
        double2 s = make_double2(0.0, 0.0), d = s;
        if (live && slot < ncomp) {
            double2 fn, fs;
            ring_modes_ns(P, Y, slot, rp, m, w, fn, fs);
            s = cadd(fn, fs);
            d = csub(fn, fs);
        }
        double *base = F + (((long long)m * P.nrp_pad + rp) * 2) * NCOL + 2 * slot;
        *reinterpret_cast<double2 *>(base) = s;
        *reinterpret_cast<double2 *>(base + NCOL) = d;
    } else {
        const int f = slot >> 1, op = slot & 1;
        double4 o0 = make_double4(0.0, 0.0, 0.0, 0.0), o1 = o0;
        if (live && 2 * f + 1 < ncomp) {
            double2 qn, qs, un, us;
            ring_modes_ns(P, Y, 2 * f, rp, m, w, qn, qs);
            ring_modes_ns(P, Y, 2 * f + 1, rp, m, w, un, us);
            // P+ = -(Q + iU)/2, P- = -(Q - iU)/2
            double2 ppn = cscale(cadd(qn, mul_pi(un)), -0.5), pmn = cscale(csub(qn, mul_pi(un)), -0.5);
            double2 pps = cscale(cadd(qs, mul_pi(us)), -0.5), pms = cscale(csub(qs, mul_pi(us)), -0.5);
            // B+(P) = [Pr, Pi, Pi, -Pr]   (E_re, E_im, B_re, B_im columns, lambda^+ operand)
            // B-(P) = [Pr, Pi, -Pi, Pr]   (lambda^- operand)
            double4 x, y;
            if (op == 0) {
                x = make_double4(ppn.x, ppn.y, ppn.y, -ppn.x);  // B+(P+_N)
                y = make_double4(pms.x, pms.y, -pms.y, pms.x);  // B-(P-_S)
            } else {
                x = make_double4(pmn.x, pmn.y, -pmn.y, pmn.x);  // B-(P-_N)
                y = make_double4(pps.x, pps.y, pps.y, -pps.x);  // B+(P+_S)
            }
            o0 = make_double4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
            o1 = make_double4(x.x - y.x, x.y - y.y, x.z - y.z, x.w - y.w);
        }
        double *base = F + ((((long long)m * P.nrp_pad + rp) * 2) * 2 + op) * NCOL + 4 * f;
        *reinterpret_cast<double4 *>(base) = o0;
        *reinterpret_cast<double4 *>(base + 2 * NCOL) = o1;
    }
}

// =====================================================================================
// 3. Legendre analysis on FP64 MFMA
// =====================================================================================
struct SVal {
    double v;
    int e;  // value = v * 2^(300 e)
};

__device__ inline void snorm_small(SVal &s)
{
    if (s.v != 0.0)
        while (fabs(s.v) < SC_SMALL) {
            s.v *= SC_BIG;
            s.e -= 1;
        }
}

// x^n for 0 <= x <= 1 with extended exponent
__device__ inline SVal spow(double x, int n)
{
    SVal r = {1.0, 0}, b = {x, 0};
    while (n) {
        if (n & 1) {
            r.v *= b.v;
            r.e += b.e;
            snorm_small(r);
        }
        n >>= 1;
        if (n) {
            b.v *= b.v;
            b.e *= 2;
            snorm_small(b);
        }
    }
    return r;
}

__device__ inline double sval_true(double v, int e)
{
    return e == 0 ? v : (e == -1 ? v * SC_SMALL : 0.0);
}

struct LegParams {
    PlanDev P;
    const LegTask *__restrict__ tasks;
    const double *__restrict__ F;
    const double2 *__restrict__ rec0;
    const double4 *__restrict__ rec2;
    double *__restrict__ partial;
    int ablate;  // diagnostic only (HX_ABLATE): 1 skip MFMA, 2 skip recursion, 4 skip flush, 8 count paths
    unsigned long long *counters;  // [dead, live, mixed, wave_off] block counts when ablate & 8
};

// broadcast lane `src` of a wave-distributed double to all lanes (-> SGPR pair)
__device__ inline double bcast(double v, int src)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}

template <int SPIN>
__global__ __launch_bounds__(LA_WAVES * 64) void k_legendre_analysis(LegParams A,
                                                                   const double2 *__restrict__ coefn,
                                                                   const double *__restrict__ alphan)
{
    constexpr int NOP = SPIN == 0 ? 1 : 2;
    __shared__ double tiles[LA_WAVES][2][16][TILE_LD];  // 128 KiB; after the MFMA phase the first
                                                        // 4 KiB of each wave's tiles carry its D tiles
    __shared__ double2 coefs[2][LBLK];                  // recursion coefficients of this / the next block
    const PlanDev &P = A.P;
    const LegTask task = A.tasks[blockIdx.x];
    const int m = task.m, lmax = P.lmax;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool wave_on = w < task.nrb;
    const int rb = task.rb0 + (wave_on ? w : 0);
    const int rp = rb * 64 + lane;
    const bool valid = wave_on && rp < P.nrp;
    const double x = valid ? P.z[rp] : 0.0;
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const int off = (l0 + m) & 1;
    const long long cb = almidx(lmax, 0, m);

    // ---- B operands: F[m][rp][par][op][16], lane (k = lane>>4, j = lane&15) ----------
    double fr[NOP][2][16];
    {
        const int j = lane & 15, k = lane >> 4;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const long long row = (long long)m * P.nrp_pad + rb * 64 + q + 16 * k;
#pragma unroll
            for (int par = 0; par < 2; ++par)
#pragma unroll
                for (int op = 0; op < NOP; ++op)
                    fr[op][par][q] = wave_on ? A.F[((row * 2 + par) * NOP + op) * NCOL + j] : 0.0;
        }
    }

    // ---- seeds ----------------------------------------------------------------------
    double vc[NOP], vp[NOP];
    int sc[NOP];
    if (SPIN == 0) {
        SVal s = {0.0, 0};
        if (valid) {
            s = spow(P.sth[rp], m);
            s.v *= P.mfac[m];
        }
        vc[0] = s.v; vp[0] = 0.0; sc[0] = valid ? s.e : -100;
    } else {
        SVal sp = {0.0, -100}, sm = {0.0, -100};
        if (valid) {
            const double sth = P.sth[rp], omx = P.omz[rp], opx = 2.0 - omx;
            const double nrm = sqrt((2.0 * l0 + 1.0) / (4.0 * M_PI));
            if (m == 0) {
                double d = 0.61237243569579452455 * sth * sth;  // sqrt(6)/4
                sp.v = sm.v = nrm * d; sp.e = sm.e = 0;
            } else if (m == 1) {
                sp.v = nrm * (-0.5 * omx * sth); sp.e = 0;
                sm.v = nrm * (0.5 * opx * sth);  sm.e = 0;
            } else {
                SVal b = spow(sth, m - 2);
                b.v *= P.kfac2[m] * nrm * ((m & 1) ? -1.0 : 1.0);
                sp.v = b.v * (0.25 * omx * omx); sp.e = b.e;
                sm.v = b.v * (0.25 * opx * opx); sm.e = b.e;
            }
            snorm_small(sp);
            snorm_small(sm);
        }
        vc[0] = sp.v; vp[0] = 0.0; sc[0] = sp.e;
        if (NOP > 1) { vc[NOP - 1] = sm.v; vp[NOP - 1] = 0.0; sc[NOP - 1] = sm.e; }
    }

    double *mytile = &tiles[w][0][0][0];
    double *myflush = mytile;
    const int ai = lane & 15, ak = lane >> 4;
    // tile element (row r of parity tile p, ring c) lives at ((p*16 + r)*64 + (c ^ r));
    // MFMA q contracts the rings {q, q+16, q+32, q+48} of this wave (k = lane>>4).
    auto mfma_block = [&](int op, double4_t (&acc)[2]) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // tile is private to the wave:
        __builtin_amdgcn_wave_barrier();                        // order LDS writes before reads
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (A.ablate & 1) return;
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int par = 0; par < 2; ++par) {
                const double a = mytile[(par * 16 + ai) * TILE_LD + ((q + 16 * ak) ^ ai)];
                acc[par] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[op][par][q], acc[par], 0, 0, 0);
            }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    auto scale_of = [](int e) { return e == 0 ? 1.0 : (e == -1 ? SC_SMALL : 0.0); };

    // spin 0: the normalised two-step recursion mu_{l+2} = (A' x^2 + B') mu_l - mu_{l-2}
    // (lambda_l = alpha_l mu_l) gives two independent chains (even / odd l - m) per ring,
    // each feeding one parity tile; alpha_l is applied to the output rows at the flush.
    // State of chain p: (wp[p], wc[p]) = mu at l-2, l; alpha_m = alpha_{m+1} = 1.
    double wc[2] = {0.0, 0.0}, wp[2] = {0.0, 0.0};
    const double x2 = x * x;
    if (SPIN == 0) {
        wc[0] = vc[0];
        wc[1] = sqrt(2.0 * m + 3.0) * x * vc[0];  // lambda_{m+1,m} = sqrt(2m+3) x lambda_mm
    }
    double scf[NOP];
#pragma unroll
    for (int op = 0; op < NOP; ++op) scf[op] = scale_of(sc[op]);

    // The coefficients are the same for every wave of the workgroup: each block's 32 entries
    // are fetched one block ahead by threads 0..127 (one double each) and handed over through
    // LDS, so the recursion never waits on global / scalar memory.
    const int coff = SPIN == 0 ? 0 : 1;  // spin-2 coefficients are indexed by the target l
    double cpre = 0.0;
    if (threadIdx.x < 2 * LBLK)
        (&coefs[0][0].x)[threadIdx.x] = reinterpret_cast<const double *>(coefn + cb + l0 + coff)[threadIdx.x];
    __syncthreads();
    int cbuf = 0;
    for (int lb = l0; lb <= lmax; lb += LBLK, cbuf ^= 1) {
        if (threadIdx.x < 2 * LBLK)
            cpre = reinterpret_cast<const double *>(coefn + cb + lb + LBLK + coff)[threadIdx.x];
        const double2 *cf = coefs[cbuf];
        double4_t acc[2];
        acc[0] = (double4_t){0.0, 0.0, 0.0, 0.0};
        acc[1] = (double4_t){0.0, 0.0, 0.0, 0.0};
        if (wave_on && !(A.ablate & 2)) {
            if (SPIN == 0) {
                const bool all_live = __all(sc[0] == 0 || !valid);
                const bool all_dead = __all(sc[0] <= -3 || !valid);
                if ((A.ablate & 8) && lane == 0) atomicAdd(&A.counters[all_dead ? 0 : (all_live ? 1 : 2)], 1ULL);
                auto advance2 = [&](int j) {
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const double2 c = cf[2 * j + p];
                        const double vn = fma(fma(c.x, x2, c.y), wc[p], -wp[p]);
                        wp[p] = wc[p];
                        wc[p] = vn;
                    }
                };
                auto rescale = [&]() {
                    if (__any(fabs(wc[0]) > SC_BIG || fabs(wc[1]) > SC_BIG)) {
                        if (fabs(wc[0]) > SC_BIG || fabs(wc[1]) > SC_BIG) {
                            wc[0] *= SC_SMALL; wp[0] *= SC_SMALL; wc[1] *= SC_SMALL; wp[1] *= SC_SMALL;
                            sc[0] += 1;
                            scf[0] = scale_of(sc[0]);
                        }
                    }
                };
                if (all_dead) {
                    for (int j = 0; j < LBLK / 2; ++j) { advance2(j); rescale(); }
                } else {
                    if (all_live) {
#pragma unroll
                        for (int j = 0; j < LBLK / 2; ++j) {
                            mytile[(0 * 16 + j) * TILE_LD + (lane ^ j)] = wc[0];
                            mytile[(1 * 16 + j) * TILE_LD + (lane ^ j)] = wc[1];
                            advance2(j);
                        }
                    } else {
#pragma unroll 4
                        for (int j = 0; j < LBLK / 2; ++j) {
                            mytile[(0 * 16 + j) * TILE_LD + (lane ^ j)] = wc[0] * scf[0];
                            mytile[(1 * 16 + j) * TILE_LD + (lane ^ j)] = wc[1] * scf[0];
                            advance2(j);
                            rescale();
                        }
                    }
                    mfma_block(0, acc);
                }
            } else {
#pragma unroll
                for (int op = 0; op < NOP; ++op) {
                    const bool all_live = __all(sc[op] == 0 || !valid);
                    const bool all_dead = __all(sc[op] <= -3 || !valid);
                    if ((A.ablate & 8) && lane == 0) atomicAdd(&A.counters[all_dead ? 0 : (all_live ? 1 : 2)], 1ULL);
                    const double sgn = op == 0 ? 1.0 : -1.0;
                    auto advance = [&](int s) {
                        const double2 c = cf[s];
                        const double vn = fma(fma(c.x, x, sgn * c.y), vc[op], -vp[op]);
                        vp[op] = vc[op];
                        vc[op] = vn;
                    };
                    auto rescale = [&]() {
                        if (__any(fabs(vc[op]) > SC_BIG)) {
                            if (fabs(vc[op]) > SC_BIG) {
                                vc[op] *= SC_SMALL; vp[op] *= SC_SMALL;
                                sc[op] += 1;
                                scf[op] = scale_of(sc[op]);
                            }
                        }
                    };
                    if (all_dead) {
                        for (int s = 0; s < LBLK; ++s) { advance(s); rescale(); }
                        continue;
                    }
                    if (all_live) {
#pragma unroll 8
                        for (int s = 0; s < LBLK; ++s) {
                            mytile[(((s + off) & 1) * 16 + (s >> 1)) * TILE_LD + (lane ^ (s >> 1))] = vc[op];
                            advance(s);
                        }
                    } else {
#pragma unroll 4
                        for (int s = 0; s < LBLK; ++s) {
                            mytile[(((s + off) & 1) * 16 + (s >> 1)) * TILE_LD + (lane ^ (s >> 1))] = vc[op] * scf[op];
                            advance(s);
                            rescale();
                        }
                    }
                    mfma_block(op, acc);
                }
            }
        }
        // ---- flush: combine the waves' partial tiles through LDS ---------------------
        // D layout of v_mfma_f64_16x16x4_f64: row = (lane>>4) + 4*reg, col = lane&15
        if (threadIdx.x < 2 * LBLK) (&coefs[cbuf ^ 1][0].x)[threadIdx.x] = cpre;
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                myflush[par * 256 + ((lane >> 4) + 4 * r) * 16 + (lane & 15)] = acc[par][r];
        __syncthreads();
        for (int t = threadIdx.x; t < 512; t += LA_WAVES * 64) {
            const int par = t >> 8, r16 = (t >> 4) & 15, col = t & 15;
            double s = 0.0;
#pragma unroll
            for (int ww = 0; ww < LA_WAVES; ++ww) s += (&tiles[ww][0][0][0])[par * 256 + r16 * 16 + col];
            const int l = lb + 2 * r16 + (par ^ off);
            if (l <= lmax) A.partial[(task.pout + (l - l0)) * NCOL + col] = s * alphan[cb + l];
        }
        __syncthreads();  // D tiles consumed: the tile buffers may be overwritten by the next block
    }
}

// =====================================================================================
// 4. partial sums -> alm
// =====================================================================================
template <int SPIN>
__global__ __launch_bounds__(256) void k_alm_reduce(PlanDev P, const LegTask *__restrict__ tasks,
                                                    const MTasks *__restrict__ of_m,
                                                    const double *__restrict__ partial, int ncomp,
                                                    const double *__restrict__ fl, int add,
                                                    double2 *__restrict__ alm, long long alm_stride)
{
    const int m = blockIdx.x, lmax = P.lmax;
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const MTasks mt = of_m[m];
    const int nl = lmax - m + 1;
    for (int i = threadIdx.x; i < nl * 8; i += blockDim.x) {
        const int l = m + (i >> 3), c = i & 7;
        if (c >= ncomp) continue;
        double2 v = make_double2(0.0, 0.0);
        if (l >= l0) {
            // spin 0: comp c -> cols 2c,2c+1; spin 2: comp 2f+e -> cols 4f+2e, +1 (== 2c)
            for (int t = 0; t < mt.count; ++t) {
                const double *p = partial + (tasks[mt.first + t].pout + (l - l0)) * NCOL + 2 * c;
                v.x += p[0];
                v.y += p[1];
            }
            if (fl) { v.x *= fl[l]; v.y *= fl[l]; }
        }
        double2 *dst = alm + (long long)c * alm_stride + almidx(lmax, l, m);
        if (add) { double2 o = *dst; v.x += o.x; v.y += o.y; }
        *dst = v;
    }
}

// =====================================================================================
// Legendre synthesis (VALU, lanes = ring pairs): F_m(r) = sum_l a_lm lambda_lm(r)
// =====================================================================================
// alm operands staged per m as AL[m-offset ...]: we read alm directly: for each l the
// 2*ncomp reals of (c, re/im) are wave-uniform scalar loads.
// Output: Fsyn[m][rp][ns][16]: ns = 0 north, 1 south; columns = 2c+{re,im} of F_m.
// spin 2: columns 4f + {Qre,Qim,Ure,Uim}.
template <int SPIN>
__global__ __launch_bounds__(256) void k_legendre_synthesis(PlanDev P, const double2 *__restrict__ alm,
                                                            long long alm_stride, int ncomp,
                                                            const MTasks *__restrict__ first_rp,
                                                            double *__restrict__ Fsyn)
{
    constexpr int NOP = SPIN == 0 ? 1 : 2;
    const int m = blockIdx.x, lmax = P.lmax;
    const int rp = blockIdx.y * 256 + threadIdx.x;
    const bool valid = rp < P.nrp;
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const long long cb = almidx(lmax, 0, m);
    // whole block below the first active ring pair: write zeros
    const bool block_dead = (int)(blockIdx.y * 256 + 255) < first_rp[m].first;
    double ev[NCOL], od[NCOL];
#pragma unroll
    for (int i = 0; i < NCOL; ++i) ev[i] = od[i] = 0.0;
    if (!block_dead && l0 <= lmax) {
        const double x = valid ? P.z[rp] : 0.0;
        double vc[NOP], vp[NOP];
        int sc[NOP];
        if (SPIN == 0) {
            SVal s = {0.0, -100};
            if (valid) {
                s = spow(P.sth[rp], m);
                s.v *= P.mfac[m];
            }
            vc[0] = s.v; vp[0] = 0.0; sc[0] = s.e;
        } else {
            SVal sp = {0.0, -100}, sm = {0.0, -100};
            if (valid) {
                const double sth = P.sth[rp], omx = P.omz[rp], opx = 2.0 - omx;
                const double nrm = sqrt((2.0 * l0 + 1.0) / (4.0 * M_PI));
                if (m == 0) {
                    double d = 0.61237243569579452455 * sth * sth;
                    sp.v = sm.v = nrm * d; sp.e = sm.e = 0;
                } else if (m == 1) {
                    sp.v = nrm * (-0.5 * omx * sth); sp.e = 0;
                    sm.v = nrm * (0.5 * opx * sth);  sm.e = 0;
                } else {
                    SVal b = spow(sth, m - 2);
                    b.v *= P.kfac2[m] * nrm * ((m & 1) ? -1.0 : 1.0);
                    sp.v = b.v * (0.25 * omx * omx); sp.e = b.e;
                    sm.v = b.v * (0.25 * opx * opx); sm.e = b.e;
                }
                snorm_small(sp);
                snorm_small(sm);
            }
            vc[0] = sp.v; vp[0] = 0.0; sc[0] = sp.e;
            if (NOP > 1) { vc[NOP - 1] = sm.v; vp[NOP - 1] = 0.0; sc[NOP - 1] = sm.e; }
        }
        for (int l = l0; l <= lmax; ++l) {
            const bool odd = (l + m) & 1;
            double lam[NOP];
#pragma unroll
            for (int op = 0; op < NOP; ++op) lam[op] = sval_true(vc[op], sc[op]);
            if (SPIN == 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (c < ncomp) {
                        const double2 a = alm[(long long)c * alm_stride + cb + l];
                        if (odd) { od[2 * c] = fma(lam[0], a.x, od[2 * c]); od[2 * c + 1] = fma(lam[0], a.y, od[2 * c + 1]); }
                        else     { ev[2 * c] = fma(lam[0], a.x, ev[2 * c]); ev[2 * c + 1] = fma(lam[0], a.y, ev[2 * c + 1]); }
                    }
                }
            } else {
                // Q_m = -sum (E F1 + i B F2), U_m = -sum (B F1 - i E F2)
                // F1 = (lam+ + lam-)/2, F2 = (lam+ - lam-)/2.  South: F1 -> p F1, F2 -> -p F2.
                // ev/od hold the F1 parts in [4f..4f+3] of ev/od and the F2 parts in evod2
                const double f1 = 0.5 * (lam[0] + lam[NOP - 1]), f2 = 0.5 * (lam[0] - lam[NOP - 1]);
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    if (2 * f + 1 < ncomp) {
                        const double2 E = alm[(long long)(2 * f) * alm_stride + cb + l];
                        const double2 B = alm[(long long)(2 * f + 1) * alm_stride + cb + l];
                        // north contribution n = -(E f1 + i B f2) etc.; south uses s1 = p f1, s2 = -p f2
                        // accumulate A = f1-part, C = f2-part separately by parity:
                        //   parity even: north A + C, south A - C ; parity odd: north A + C, south -A + C
                        const double qa_r = -f1 * E.x, qa_i = -f1 * E.y;   // -(E f1)
                        const double qc_r = f2 * B.y, qc_i = -f2 * B.x;    // -(i B f2)
                        const double ua_r = -f1 * B.x, ua_i = -f1 * B.y;   // -(B f1)
                        const double uc_r = -f2 * E.y, uc_i = f2 * E.x;    // +(i E f2)
                        // ev := quantity that is the same north and south; od := flips sign
                        if (!odd) {
                            ev[4 * f] += qa_r; ev[4 * f + 1] += qa_i; ev[4 * f + 2] += ua_r; ev[4 * f + 3] += ua_i;
                            od[4 * f] += qc_r; od[4 * f + 1] += qc_i; od[4 * f + 2] += uc_r; od[4 * f + 3] += uc_i;
                        } else {
                            od[4 * f] += qa_r; od[4 * f + 1] += qa_i; od[4 * f + 2] += ua_r; od[4 * f + 3] += ua_i;
                            ev[4 * f] += qc_r; ev[4 * f + 1] += qc_i; ev[4 * f + 2] += uc_r; ev[4 * f + 3] += uc_i;
                        }
                    }
                }
            }
            // advance
#pragma unroll
            for (int op = 0; op < NOP; ++op) {
                double vn;
                if (SPIN == 0) {
                    const double2 c = P.rec0[cb + l + 1];
                    vn = fma(c.x * x, vc[op], -c.y * vp[op]);
                } else {
                    const double4 c = P.rec2[cb + l + 1];
                    const double cc = op == 0 ? c.y : -c.y;
                    vn = fma(fma(c.x, x, cc), vc[op], -c.z * vp[op]);
                }
                vp[op] = vc[op]; vc[op] = vn;
                if (fabs(vc[op]) > SC_BIG) { vc[op] *= SC_SMALL; vp[op] *= SC_SMALL; sc[op] += 1; }
            }
        }
    }
    if (rp < P.nrp_pad) {
        double *base = Fsyn + (((long long)m * P.nrp_pad + rp) * 2) * NCOL;
#pragma unroll
        for (int i = 0; i < NCOL; ++i) {
            base[i] = ev[i] + od[i];          // north
            base[NCOL + i] = ev[i] - od[i];   // south
        }
    }
}

// Fsyn -> conj(Z) spectra of the packed ring pair z = f_N + i f_S.
// X[k] = sum_{m == k mod nphi} (c_m/2) Ft_m + sum_{m == -k} (c_m/2) conj(Ft_m), Ft = F e^{i m phi0}
// grid: x = ring pair, y = comp; block loops over k.  Output Zc[c][startN + k] = conj(X_N + i X_S)
__global__ __launch_bounds__(256) void k_synth_spectrum(PlanDev P, const double *__restrict__ Fsyn,
                                                        int lmax, double2 *__restrict__ Zc)
{
    const int rp = blockIdx.x, c = blockIdx.y;
    const int n = P.nsub[rp], nphi = 4 * n;
    const bool shifted = P.shifted[rp] != 0;
    for (int k = threadIdx.x; k < nphi; k += blockDim.x) {
        double2 xn = make_double2(0.0, 0.0), xs = xn;
        // m == k (mod nphi)
        for (int m = k; m <= lmax; m += nphi) {
            const double *b = Fsyn + (((long long)m * P.nrp_pad + rp) * 2) * NCOL + 2 * c;
            double2 ph = make_double2(1.0, 0.0);
            if (shifted) ph = expipi((double)(m % (2 * nphi)) / (double)nphi);
            double2 fn = cmul(make_double2(b[0], b[1]), ph), fs = cmul(make_double2(b[NCOL], b[NCOL + 1]), ph);
            const double sc = m == 0 ? 0.5 : 1.0;  // c_m / 2
            xn = cadd(xn, cscale(fn, sc));
            xs = cadd(xs, cscale(fs, sc));
        }
        // m == -k (mod nphi)
        for (int m = (nphi - k) % nphi; m <= lmax; m += nphi) {
            const double *b = Fsyn + (((long long)m * P.nrp_pad + rp) * 2) * NCOL + 2 * c;
            double2 ph = make_double2(1.0, 0.0);
            if (shifted) ph = expipi((double)(m % (2 * nphi)) / (double)nphi);
            double2 fn = cconj(cmul(make_double2(b[0], b[1]), ph)), fs = cconj(cmul(make_double2(b[NCOL], b[NCOL + 1]), ph));
            const double sc = m == 0 ? 0.5 : 1.0;
            xn = cadd(xn, cscale(fn, sc));
            xs = cadd(xs, cscale(fs, sc));
        }
        // Z = X_N + i X_S ; store conj(Z)
        double2 zz = cadd(xn, mul_pi(xs));
        Zc[(long long)c * P.ny + P.startN[rp] + k] = cconj(zz);
    }
}

// Y_r[k] = DFT(conj Z)[4k+r] = conj(z[4k+r]) -> f_N = Re, f_S = -Im
__global__ __launch_bounds__(256) void k_synth_scatter(PlanDev P, const double2 *__restrict__ Y,
                                                       double *__restrict__ maps, int accumulate_neg,
                                                       const double *__restrict__ ref)
{
    const int rp = blockIdx.x, c = blockIdx.y;
    const int n = P.nsub[rp];
    const long long sN = P.startN[rp], sS = P.startS[rp];
    const double2 *y = Y + (long long)c * P.ny + sN;
    double *mp = maps + (long long)c * P.npix;
    const double *rf = ref ? ref + (long long)c * P.npix : nullptr;
    for (int i = threadIdx.x; i < 4 * n; i += blockDim.x) {
        const int r = i / n, k = i - r * n;
        const int j = 4 * k + r;
        const double2 v = y[i];
        double fn = v.x, fs = -v.y;
        if (accumulate_neg) {  // residual: ref - synthesised
            fn = rf[sN + j] - fn;
            if (sS >= 0) fs = rf[sS + j] - fs;
        }
        mp[sN + j] = fn;
        if (sS >= 0) mp[sS + j] = fs;
    }
}

}  // namespace hx

using namespace hx;

// =====================================================================================
// plan
// =====================================================================================
PlanDev hx_plan::dev() const
{
    PlanDev P;
    P.nside = nside; P.lmax = lmax; P.nrp = nrp; P.nrp_pad = nrp_pad; P.twN = twN;
    P.npix = npix; P.ny = ny;
    P.z = z.as<double>(); P.omz = omz.as<double>(); P.sth = sth.as<double>(); P.rwdef = rwdef.as<double>();
    P.nsub = nsub.as<int>(); P.shifted = shifted.as<int>();
    P.startN = startN.as<long long>(); P.startS = startS.as<long long>(); P.bhat_off = bhat_off.as<long long>();
    P.tw = tw.as<double2>(); P.bhat = bhat.as<double2>();
    P.mfac = mfac.as<double>(); P.kfac2 = kfac2.as<double>();
    P.rec0 = rec0.as<double2>(); P.rec2 = rec2.as<double4>();
    return P;
}

template <class T>
static int upload(DevBuf &b, const std::vector<T> &v)
{
    HX_TRY(b.alloc(sizeof(T) * std::max<size_t>(v.size(), 1)));
    if (!v.empty()) HX_HIP(hipMemcpy(b.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return HX_OK;
}

// libsharp's published heuristic for the largest m that contributes on a ring
// (sharp_get_mlim): rings with m > mlim are skipped.
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

static int build_tasks(hx_plan *pl, int spin)
{
    hx_plan::TaskSet &ts = pl->ts[spin ? 1 : 0];
    if (ts.built) return HX_OK;
    const int lmax = pl->lmax;
    ts.tasks.clear();
    ts.of_m.assign(lmax + 1, MTasks{0, 0});
    std::vector<std::vector<LegTask>> per_m(lmax + 1);
    for (int m = 0; m <= lmax; ++m) {
        const int l0 = spin == 0 ? m : std::max(m, 2);
        if (l0 > lmax) continue;
        int first = pl->nrp;  // first active ring pair (rings are ordered pole -> equator)
        for (int rp = 0; rp < pl->nrp; ++rp)
            if (ring_mlim(lmax, spin, pl->h_sth[rp], pl->h_z[rp]) >= m) { first = rp; break; }
        if (first >= pl->nrp) first = pl->nrp - 1;
        int rb = first / 64;
        while (rb < pl->nrb) {
            LegTask t;
            t.m = m; t.rb0 = rb; t.nrb = std::min(LA_WAVES, pl->nrb - rb); t.pad = 0; t.pout = 0;
            per_m[m].push_back(t);
            rb += t.nrb;
        }
    }
    // partial-buffer rows and per-m index (tasks of one m stay contiguous)
    long long rows = 0;
    for (int m = 0; m <= lmax; ++m) {
        const int l0 = spin == 0 ? m : std::max(m, 2);
        ts.of_m[m].first = (int)ts.tasks.size();
        ts.of_m[m].count = (int)per_m[m].size();
        for (auto &t : per_m[m]) {
            t.pout = rows;
            rows += (lmax - l0 + 1);
            ts.tasks.push_back(t);
        }
    }
    ts.rows = rows;
    HX_TRY(upload(ts.d_tasks, ts.tasks));
    HX_TRY(upload(ts.d_of_m, ts.of_m));
    ts.built = true;
    return HX_OK;
}

extern "C" hx_plan *hx_plan_create(int nside, int lmax, int max_comp)
{
    if (ensure_ready() != HX_OK) return nullptr;
    if (nside < 1 || lmax < 0 || max_comp < 1) {
        set_error("hx_plan_create: bad argument");
        return nullptr;
    }
    hx_plan *pl = new hx_plan;
    pl->nside = nside; pl->lmax = lmax; pl->max_comp = max_comp;
    pl->npix = 12LL * nside * nside;
    pl->nrp = 2 * nside;
    pl->nrp_pad = (pl->nrp + 63) / 64 * 64;
    pl->nrb = pl->nrp_pad / 64;
    pl->nlm = (long long)(lmax + 1) * (lmax + 2) / 2;
    const long long ns = nside, ncap = 2 * ns * (ns - 1);
    std::vector<double> z(pl->nrp), omz(pl->nrp), sth(pl->nrp), rw(pl->nrp, 1.0);
    std::vector<int> nsub(pl->nrp), shifted(pl->nrp);
    std::vector<long long> sN(pl->nrp), sS(pl->nrp), boff(pl->nrp, -1);
    const double fact2 = 4.0 / (double)pl->npix, fact1 = (double)(2 * ns) * fact2;
    int maxM = 1;
    for (int rp = 0; rp < pl->nrp; ++rp) {
        const int i = rp + 1;
        if (i < nside) {
            double tmp = (double)i * (double)i * fact2;
            z[rp] = 1.0 - tmp; omz[rp] = tmp; sth[rp] = sqrt(tmp * (2.0 - tmp));
            nsub[rp] = i; sN[rp] = 2LL * i * (i - 1); shifted[rp] = 1;
        } else {
            z[rp] = (double)(2 * nside - i) * fact1; omz[rp] = 1.0 - z[rp];
            sth[rp] = sqrt((1.0 - z[rp]) * (1.0 + z[rp]));
            nsub[rp] = nside; sN[rp] = ncap + (long long)(i - nside) * 4 * ns;
            shifted[rp] = ((i - nside) & 1) == 0;
        }
        sS[rp] = i == 2 * nside ? -1 : pl->npix - sN[rp] - 4LL * nsub[rp];
        maxM = std::max(maxM, fft_size_for(nsub[rp]));
    }
    pl->ny = sN[pl->nrp - 1] + 4LL * nsub[pl->nrp - 1];
    if ((size_t)maxM * sizeof(double2) > 160 * 1024) {
        set_error("hx_plan_create: nside=%d needs an in-LDS FFT of %d points (> 8192); unsupported", nside, maxM);
        delete pl;
        return nullptr;
    }
    pl->twN = std::max(maxM, 2);
    pl->lds_fft = (size_t)maxM * sizeof(double2);
    pl->h_sth = sth; pl->h_z = z; pl->h_nsub = nsub;
    // Bluestein tables: one spectrum per distinct non-power-of-two sub-length
    std::vector<int> blu_list;
    long long btot = 0;
    {
        std::map<int, long long> off_of_n;
        for (int rp = 0; rp < pl->nrp; ++rp) {
            int n = nsub[rp], M = fft_size_for(n);
            if (M == n) continue;
            auto it = off_of_n.find(n);
            if (it == off_of_n.end()) {
                it = off_of_n.emplace(n, btot).first;
                btot += M;
                blu_list.push_back(rp);
            }
            boff[rp] = it->second;
        }
    }
    std::vector<double2> tw(pl->twN / 2);
    for (int k = 0; k < pl->twN / 2; ++k) {
        long double a = -2.0L * 3.141592653589793238462643383279502884L * k / pl->twN;
        tw[k].x = (double)cosl(a); tw[k].y = (double)sinl(a);
    }
    // mfac[m] = (-1)^m sqrt((2m+1)/(4pi) prod_{k<=m} (2k-1)/(2k));  kfac2[m] = K_m 2^-(m-2)
    std::vector<double> mfac(lmax + 1), kfac2(lmax + 3, 0.0);
    {
        long double p = 1.0L;
        for (int m = 0; m <= lmax; ++m) {
            if (m > 0) p *= (2.0L * m - 1.0L) / (2.0L * m);
            long double v = sqrtl((2.0L * m + 1.0L) / (4.0L * 3.141592653589793238462643383279502884L) * p);
            mfac[m] = (double)((m & 1) ? -v : v);
        }
        long double k = 1.0L;
        for (int m = 2; m <= lmax + 2; ++m) {
            if (m > 2) k *= sqrtl((2.0L * m) * (2.0L * m - 1.0L) / ((m - 2.0L) * (m + 2.0L))) / 2.0L;
            kfac2[m] = (double)k;
        }
    }
    int rc = HX_OK;
    auto chk = [&](int r) { if (rc == HX_OK) rc = r; };
    chk(upload(pl->z, z)); chk(upload(pl->omz, omz)); chk(upload(pl->sth, sth)); chk(upload(pl->rwdef, rw));
    chk(upload(pl->nsub, nsub)); chk(upload(pl->shifted, shifted));
    chk(upload(pl->startN, sN)); chk(upload(pl->startS, sS)); chk(upload(pl->bhat_off, boff));
    chk(upload(pl->tw, tw)); chk(upload(pl->mfac, mfac)); chk(upload(pl->kfac2, kfac2));
    chk(pl->bhat.alloc(sizeof(double2) * std::max<long long>(btot, 1)));
    chk(pl->rec0.alloc(sizeof(double2) * (pl->nlm + 128)));
    if (rc != HX_OK) { delete pl; return nullptr; }
    hipStream_t st = rt().stream;
    (void)hipMemsetAsync(pl->rec0.p, 0, sizeof(double2) * (pl->nlm + 128), st);
    hipLaunchKernelGGL(k_init_rec0, dim3(lmax + 1), dim3(256), 0, st, lmax, pl->rec0.as<double2>());
    if (pl->cn0.alloc(sizeof(double2) * (pl->nlm + 128)) != HX_OK || pl->al0.alloc(sizeof(double) * (pl->nlm + 128)) != HX_OK) {
        delete pl;
        return nullptr;
    }
    (void)hipMemsetAsync(pl->cn0.p, 0, sizeof(double2) * (pl->nlm + 128), st);
    (void)hipMemsetAsync(pl->al0.p, 0, sizeof(double) * (pl->nlm + 128), st);
    hipLaunchKernelGGL(k_init_norm0, dim3((2 * (lmax + 1) + 63) / 64), dim3(64), 0, st, lmax, pl->cn0.as<double2>(), pl->al0.as<double>());
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_init_bhat), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_subdft<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_ring_subdft<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (!blu_list.empty()) {
        DevBuf d_list;
        if (upload(d_list, blu_list) != HX_OK) { delete pl; return nullptr; }
        hipLaunchKernelGGL(k_init_bhat, dim3((unsigned)blu_list.size()), dim3(512), pl->lds_fft, st, pl->dev(),
                           d_list.as<int>(), pl->bhat.as<double2>());
        (void)hipStreamSynchronize(st);
    }
    if (hipStreamSynchronize(st) != hipSuccess || hipGetLastError() != hipSuccess) {
        set_error("hx_plan_create: table initialisation failed");
        delete pl;
        return nullptr;
    }
    return pl;
}

extern "C" void hx_plan_destroy(hx_plan *plan)
{
    if (!plan) return;
    if (rt().ready) (void)hipStreamSynchronize(rt().stream);
    delete plan;
}

extern "C" int64_t hx_plan_scratch_bytes(const hx_plan *pl)
{
    if (!pl) return 0;
    return (int64_t)(pl->Y.bytes + pl->F.bytes + pl->partial.bytes + pl->rec0.bytes + pl->rec2.bytes + pl->cn0.bytes + pl->al0.bytes + pl->cn2.bytes + pl->al2.bytes +
                     pl->bhat.bytes + pl->stage_maps.bytes + pl->stage_alms.bytes + pl->resid.bytes + pl->Fsyn.bytes);
}

static int ensure_rec2(hx_plan *pl)
{
    if (pl->rec2.p) return HX_OK;
    HX_TRY(pl->rec2.alloc(sizeof(double4) * (pl->nlm + 128)));
    HX_HIP(hipMemsetAsync(pl->rec2.p, 0, sizeof(double4) * (pl->nlm + 128), rt().stream));
    hipLaunchKernelGGL(k_init_rec2, dim3(pl->lmax + 1), dim3(256), 0, rt().stream, pl->lmax, pl->rec2.as<double4>());
    HX_TRY(pl->cn2.alloc(sizeof(double2) * (pl->nlm + 128)));
    HX_TRY(pl->al2.alloc(sizeof(double) * (pl->nlm + 128)));
    HX_HIP(hipMemsetAsync(pl->cn2.p, 0, sizeof(double2) * (pl->nlm + 128), rt().stream));
    HX_HIP(hipMemsetAsync(pl->al2.p, 0, sizeof(double) * (pl->nlm + 128), rt().stream));
    hipLaunchKernelGGL(k_init_norm2, dim3((pl->lmax + 64) / 64), dim3(64), 0, rt().stream, pl->lmax, pl->cn2.as<double2>(), pl->al2.as<double>());
    HX_HIP(hipGetLastError());
    return HX_OK;
}

// ---- one analysis pass over a batch of <= 8 components (device pointers) -------------
static int analysis_batch(hx_plan *pl, int spin, int nb, const double *d_maps, double2 *d_alms,
                          const double *d_rw, const double *d_pw, const double *d_fl, int add)
{
    hipStream_t st = rt().stream;
    const int sidx = spin ? 1 : 0, nop = spin ? 2 : 1;
    HX_TRY(build_tasks(pl, spin));
    if (spin) HX_TRY(ensure_rec2(pl));
    hx_plan::TaskSet &ts = pl->ts[sidx];
    HX_TRY(pl->Y.alloc(sizeof(double2) * (size_t)pl->ny * 8));
    HX_TRY(pl->F.alloc(sizeof(double) * (size_t)(pl->lmax + 1) * pl->nrp_pad * 2 * 2 * NCOL));
    HX_TRY(pl->partial.alloc(sizeof(double) * (size_t)std::max<long long>(pl->ts[0].rows, pl->ts[1].rows) * NCOL));
    PlanDev P = pl->dev();
    {
        ProfScope ps("ring_fft");
        hipLaunchKernelGGL(k_ring_subdft<0>, dim3(pl->nrp, 4, nb), dim3(512), pl->lds_fft, st, P, d_maps, d_pw,
                           (const double2 *)nullptr, pl->Y.as<double2>());
    }
    {
        ProfScope ps("fourier_combine");
        dim3 grid(pl->lmax + 1, pl->nrp_pad / 32);
        if (spin == 0)
            hipLaunchKernelGGL(k_fourier_combine<0>, grid, dim3(256), 0, st, P, pl->Y.as<double2>(), nb, d_rw, pl->F.as<double>());
        else
            hipLaunchKernelGGL(k_fourier_combine<2>, grid, dim3(256), 0, st, P, pl->Y.as<double2>(), nb, d_rw, pl->F.as<double>());
    }
    {
        ProfScope ps("legendre_analysis");
        ProfScope ps2(spin == 0 ? "legendre_analysis_s0" : "legendre_analysis_s2");
        LegParams A;
        A.P = P; A.tasks = ts.d_tasks.as<LegTask>(); A.F = pl->F.as<double>(); A.partial = pl->partial.as<double>();
        A.rec0 = pl->rec0.as<double2>(); A.rec2 = pl->rec2.as<double4>();
        {
            const char *e = getenv("HX_ABLATE");
            A.ablate = e ? atoi(e) : 0;
            A.counters = nullptr;
            if (A.ablate & 8) {
                HX_TRY(pl->d_rw.alloc(64));
                HX_HIP(hipMemsetAsync(pl->d_rw.p, 0, 64, st));
                A.counters = pl->d_rw.as<unsigned long long>();
            }
        }
        if (spin == 0)
            hipLaunchKernelGGL(k_legendre_analysis<0>, dim3((unsigned)ts.tasks.size()), dim3(LA_WAVES * 64), 0, st, A, (const double2 *)pl->cn0.as<double2>(), (const double *)pl->al0.as<double>());
        else
            hipLaunchKernelGGL(k_legendre_analysis<2>, dim3((unsigned)ts.tasks.size()), dim3(LA_WAVES * 64), 0, st, A, (const double2 *)pl->cn2.as<double2>(), (const double *)pl->al2.as<double>());
    }
    if (pl->d_rw.p && getenv("HX_ABLATE") && (atoi(getenv("HX_ABLATE")) & 8)) {
        unsigned long long h[4] = {0, 0, 0, 0};
        HX_HIP(hipStreamSynchronize(st));
        HX_HIP(hipMemcpy(h, pl->d_rw.p, 32, hipMemcpyDeviceToHost));
        fprintf(stderr, "[hx] spin %d legendre wave-blocks: dead %llu live %llu mixed %llu\n", spin, h[0], h[1], h[2]);
    }
    {
        ProfScope ps("alm_reduce");
        if (spin == 0)
            hipLaunchKernelGGL(k_alm_reduce<0>, dim3(pl->lmax + 1), dim3(256), 0, st, P, ts.d_tasks.as<LegTask>(),
                               ts.d_of_m.as<MTasks>(), pl->partial.as<double>(), nb, d_fl, add, d_alms, pl->nlm);
        else
            hipLaunchKernelGGL(k_alm_reduce<2>, dim3(pl->lmax + 1), dim3(256), 0, st, P, ts.d_tasks.as<LegTask>(),
                               ts.d_of_m.as<MTasks>(), pl->partial.as<double>(), nb, d_fl, add, d_alms, pl->nlm);
    }
    (void)nop;
    HX_HIP(hipGetLastError());
    return HX_OK;
}

// ---- one synthesis pass over a batch (device pointers).  If d_ref != NULL the output is
// the residual ref - synth (Jacobi iteration). -------------------------------------------
static int synthesis_batch(hx_plan *pl, int spin, int nb, const double2 *d_alms, double *d_maps,
                           const double *d_ref)
{
    hipStream_t st = rt().stream;
    HX_TRY(build_tasks(pl, spin));
    if (spin) HX_TRY(ensure_rec2(pl));
    hx_plan::TaskSet &ts = pl->ts[spin ? 1 : 0];
    HX_TRY(pl->Y.alloc(sizeof(double2) * (size_t)pl->ny * 8));
    HX_TRY(pl->resid.alloc(sizeof(double2) * (size_t)pl->ny * 8));  // conj(Z) spectra
    HX_TRY(pl->Fsyn.alloc(sizeof(double) * (size_t)(pl->lmax + 1) * pl->nrp_pad * 2 * NCOL));
    PlanDev P = pl->dev();
    // first active ring pair per m (reuse MTasks.first as "first ring pair")
    static thread_local std::vector<MTasks> fr;
    fr.assign(pl->lmax + 1, MTasks{0, 0});
    for (int m = 0; m <= pl->lmax; ++m) {
        int first = 0;
        if (ts.of_m[m].count > 0) first = ts.tasks[ts.of_m[m].first].rb0 * 64;
        fr[m].first = first;
    }
    DevBuf d_fr;
    HX_TRY(upload(d_fr, fr));
    {
        ProfScope ps("legendre_synthesis");
        dim3 grid(pl->lmax + 1, (pl->nrp_pad + 255) / 256);
        if (spin == 0)
            hipLaunchKernelGGL(k_legendre_synthesis<0>, grid, dim3(256), 0, st, P, d_alms, pl->nlm, nb, d_fr.as<MTasks>(), pl->Fsyn.as<double>());
        else
            hipLaunchKernelGGL(k_legendre_synthesis<2>, grid, dim3(256), 0, st, P, d_alms, pl->nlm, nb, d_fr.as<MTasks>(), pl->Fsyn.as<double>());
    }
    {
        ProfScope ps("ring_fft");
        hipLaunchKernelGGL(k_synth_spectrum, dim3(pl->nrp, nb), dim3(256), 0, st, P, pl->Fsyn.as<double>(), pl->lmax, pl->resid.as<double2>());
        hipLaunchKernelGGL(k_ring_subdft<1>, dim3(pl->nrp, 4, nb), dim3(512), pl->lds_fft, st, P, (const double *)nullptr,
                           (const double *)nullptr, pl->resid.as<double2>(), pl->Y.as<double2>());
        hipLaunchKernelGGL(k_synth_scatter, dim3(pl->nrp, nb), dim3(256), 0, st, P, pl->Y.as<double2>(), d_maps, d_ref ? 1 : 0, d_ref);
    }
    HX_HIP(hipGetLastError());
    HX_HIP(hipStreamSynchronize(st));  // d_fr is freed on return
    return HX_OK;
}

static int check_sht_args(hx_plan *pl, int spin, int ncomp, const void *a, const void *b)
{
    if (!pl || !a || !b) return fail(HX_ERR_ARG, "null plan or buffer");
    if (spin != 0 && spin != 2) return fail(HX_ERR_UNSUPPORTED, "spin-%d maps not yet supported", spin);
    if (ncomp < 1 || (spin == 2 && (ncomp & 1))) return fail(HX_ERR_ARG, "bad component count %d for spin %d", ncomp, spin);
    return HX_OK;
}

namespace hx {
__global__ void k_apply_fl(int lmax, int ncomp, long long nlm, const double *__restrict__ fl, double2 *__restrict__ alm)
{
    const int m = blockIdx.x;
    for (int i = threadIdx.x; i < (lmax - m + 1) * ncomp; i += blockDim.x) {
        const int c = i / (lmax - m + 1), l = m + i % (lmax - m + 1);
        double2 *p = alm + c * nlm + almidx(lmax, l, m);
        p->x *= fl[l];
        p->y *= fl[l];
    }
}
}  // namespace hx

static int apply_fl(hx_plan *pl, int nb, double2 *alm, const double *fl)
{
    hipLaunchKernelGGL(k_apply_fl, dim3(pl->lmax + 1), dim3(256), 0, rt().stream, pl->lmax, nb, pl->nlm, fl, alm);
    HX_HIP(hipGetLastError());
    return HX_OK;
}

extern "C" int hx_map2alm(hx_plan *pl, int spin, int ncomp, const double *maps, double *alms,
                          const double *ring_weights, const double *pix_weights, const double *fl, int niter)
{
    HX_TRY(ensure_ready());
    HX_TRY(check_sht_args(pl, spin, ncomp, maps, alms));
    if (niter < 0) return fail(HX_ERR_ARG, "niter < 0");
    InView vmaps, vrw, vpw, vfl;
    OutView valms;
    HX_TRY(vmaps.bind(maps, sizeof(double) * (size_t)ncomp * pl->npix));
    HX_TRY(vrw.bind(ring_weights, sizeof(double) * pl->nrp));
    HX_TRY(vpw.bind(pix_weights, sizeof(double) * (size_t)pl->npix));
    HX_TRY(vfl.bind(fl, sizeof(double) * (pl->lmax + 1)));
    HX_TRY(valms.bind(alms, sizeof(double2) * (size_t)ncomp * pl->nlm));
    DevBuf resid;
    if (niter > 0) HX_TRY(resid.alloc(sizeof(double) * (size_t)8 * pl->npix));
    for (int c0 = 0; c0 < ncomp; c0 += 8) {
        const int nb = std::min(8, ncomp - c0);
        const double *dm = vmaps.as<double>() + (size_t)c0 * pl->npix;
        double2 *da = valms.as<double2>() + (size_t)c0 * pl->nlm;
        // the filter fl is applied once, after the last iteration
        HX_TRY(analysis_batch(pl, spin, nb, dm, da, vrw.as<double>(), vpw.as<double>(), niter == 0 ? vfl.as<double>() : nullptr, 0));
        for (int it = 0; it < niter; ++it) {
            HX_TRY(synthesis_batch(pl, spin, nb, da, resid.as<double>(), dm));
            HX_TRY(analysis_batch(pl, spin, nb, resid.as<double>(), da, vrw.as<double>(), vpw.as<double>(), nullptr, 1));
        }
        if (niter > 0 && fl) HX_TRY(apply_fl(pl, nb, da, vfl.as<double>()));
    }
    HX_TRY(valms.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));  // staging buffers are released on return
    return HX_OK;
}



extern "C" int hx_alm2map(hx_plan *pl, int spin, int ncomp, const double *alms, double *maps)
{
    HX_TRY(ensure_ready());
    HX_TRY(check_sht_args(pl, spin, ncomp, alms, maps));
    InView valms;
    OutView vmaps;
    HX_TRY(valms.bind(alms, sizeof(double2) * (size_t)ncomp * pl->nlm));
    HX_TRY(vmaps.bind(maps, sizeof(double) * (size_t)ncomp * pl->npix));
    for (int c0 = 0; c0 < ncomp; c0 += 8) {
        const int nb = std::min(8, ncomp - c0);
        HX_TRY(synthesis_batch(pl, spin, nb, valms.as<double2>() + (size_t)c0 * pl->nlm,
                               vmaps.as<double>() + (size_t)c0 * pl->npix, nullptr));
    }
    HX_TRY(vmaps.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}
