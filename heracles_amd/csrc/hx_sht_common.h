// hx_sht_common.h -- plan structure, device-side plan view and scaled-arithmetic helpers
// shared by hx_sht.hip (plan, ring Fourier stage, synthesis, C ABI) and hx_analysis.hip
// (Legendre analysis on FP64 MFMA).
#pragma once
#include <map>
#include <vector>

#include "hx_common.h"
#include "hx_fft_core.h"

namespace hx {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int LBLK = 32;   // l values per block (16 even-parity + 16 odd-parity rows)
constexpr int NCOL = 16;   // MFMA N: real columns per group (8 spin-0 maps / 4 spin-2 fields)
// entries past idx(lmax, lmax) of the recursion tables: the kernels prefetch one flush span (up to 4 x 32 l)
// ahead and read whole spans, i.e. up to 2 x 128 entries beyond the last valid one
constexpr int TABLE_PAD = 512;
constexpr int NGMAX = 2;   // column groups per analysis launch
constexpr int RBLK = 32;   // ring pairs per wave of the Legendre analysis kernel
constexpr double SC_BIG = 0x1p+300, SC_SMALL = 0x1p-300;

struct LegTask {
    int m;
    int rb0;          // first 32-ring-pair block
    int nrb;          // blocks (waves) used
    int pad;
    long long pout;   // first row of this task in the partial buffer
};

struct MTasks {
    int first, count;
};

// Device-side view of a plan (POD, passed by value to kernels).
// what a work item of the ring Fourier kernels needs to know about its ring pair, in the order of hx_plan::fft_rp_list
struct alignas(16) RingDesc {
    long long sN, sS, bhat_off;
    int n, rp;
};
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
    double wnorm;              // quadrature normalisation of a ring value: 4 pi / npix (HEALPix), 1 / N (equiangular rings)
    const double2 *hsrc;       // equiangular plan: ring spectra h_m(theta_j) [component][m][N] of the current call, else null
    long long hsrc_stride;
    int hN;
    // m-sharded route (hx_legendre_from_modes): per component a block [m - ns_m0][nrp_pad] of (F_N.re, F_N.im, F_S.re, F_S.im) with
    // phase and quadrature weight applied, as another rank's hx_ring_modes produced it; null otherwise
    const double4 *const *nssrc;
    int ns_m0, ns_ms;          // the block holds the orders ns_m0 + k ns_ms
};

__host__ __device__ inline long long almidx(int lmax, int l, int m)
{
    return (long long)m * (2 * lmax + 1 - m) / 2 + l;
}

__device__ inline double2 expipi(double x)
{
    double s, c;
    sincospi(x, &s, &c);
    return make_double2(c, s);
}

// value = v * 2^(300 e): lambda_mm ~ (sin theta)^m underflows f64 for m ~ 10^3..10^4
struct SVal {
    double v;
    int e;
};

__device__ inline void snorm_small(SVal &s)
{
    if (s.v != 0.0)
        while (fabs(s.v) < SC_SMALL) {
            s.v *= SC_BIG;
            s.e -= 1;
        }
}

// (v, e) with value = v 2^(300 e), |v| >= 2^-300  ->  value = v 2^(SB e), |v| >= 2^-SB (SB divides 300).  A chain is "live" -- its values enter
// the matrix products -- once e = 0, i.e. from 2^-SB on: the kernels that skip the matrix work of blocks whose rings are all below that
// threshold take SB = 100 (7.9e-31; libsharp drops what is below 2^-60 of its scaled values), which ends the lead-in of a ring earlier
// than SB = 300 (4.9e-91) does.
template <int SB>
__device__ inline void sval_rebase(double &v, int &e)
{
    static_assert(300 % SB == 0, "the step must divide 300");
    if (SB != 300) {
        constexpr double big = SB == 100 ? 0x1p+100 : SB == 150 ? 0x1p+150 : SB == 75 ? 0x1p+75 : SB == 60 ? 0x1p+60 : 0x1p+50;
        static_assert(SB == 300 || SB == 100 || SB == 150 || SB == 75 || SB == 60 || SB == 50, "unsupported step");
        e *= 300 / SB;
        if (v != 0.0)
            while (fabs(v) < 1.0 / big) {
                v *= big;
                e -= 1;
            }
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

// seeds of the spin-2 recursions at l0 = max(m,2): sqrt((2 l0+1)/4pi) d^{l0}_{m,-2} and d^{l0}_{m,+2}
__device__ inline void spin2_seeds(int m, double sth, double omx, double kfac2m, SVal &sp, SVal &sm)
{
    const int l0 = m > 2 ? m : 2;
    const double opx = 2.0 - omx;
    const double nrm = sqrt((2.0 * l0 + 1.0) / (4.0 * M_PI));
    if (m == 0) {
        const double d = 0.61237243569579452455 * sth * sth;  // sqrt(6)/4 sin^2
        sp.v = sm.v = nrm * d; sp.e = sm.e = 0;
    } else if (m == 1) {
        sp.v = nrm * (-0.5 * omx * sth); sp.e = 0;
        sm.v = nrm * (0.5 * opx * sth);  sm.e = 0;
    } else {
        SVal b = spow(sth, m - 2);
        b.v *= kfac2m * nrm * ((m & 1) ? -1.0 : 1.0);
        sp.v = b.v * (0.25 * omx * omx); sp.e = b.e;
        sm.v = b.v * (0.25 * opx * opx); sm.e = b.e;
    }
    snorm_small(sp);
    snorm_small(sm);
}

#ifdef __HIPCC__
using namespace hxfft;
// What a ring pair needs at one m, whatever the component: where Z[m] and Z[-m] sit in its spectrum (X[4k+r] = Y_r[k]) and
// the phase x quadrature weight.  One thread per ring pair of the block computes it (integer divisions, one sincospi) and
// hands it over through LDS: done per (ring pair, m, component slot) it was half of the kernel's time (3.7e9 vector
// instructions per spin-2 sweep, profiles/r02_pmc_summary.md).
struct RingAtM {
    double2 ph;        // w e^{-i m phi_0} (w = 0 for padding ring pairs)
    long long i0, i1;  // Z[m mod nphi], Z[-m mod nphi] relative to the component's spectrum
    int hasS;
};

// F_N(m), F_S(m) of ring pair rp for component c, including phase and quadrature weight
__device__ inline void ring_modes_ns(const PlanDev &P, const double2 *__restrict__ Y, int c, int rp, int m,
                                     const RingAtM &r, double2 &FN, double2 &FS)
{
    if (P.nssrc) {
        const double4 v = P.nssrc[c][(long long)((m - P.ns_m0) / P.ns_ms) * P.nrp_pad + rp];
        FN = make_double2(v.x, v.y);
        FS = make_double2(v.z, v.w);
        return;
    }
    if (P.hsrc) {
        // equiangular rings theta_j = 2 pi (j + 1/2) / N of the point transform: the spectrum h_m is given on the full circle,
        // lambda_lm(2 pi - theta) = (-1)^m lambda_lm(theta) (both spins) folds the second half onto the rings
        const double w = r.ph.x;
        const double2 *h = P.hsrc + (long long)c * P.hsrc_stride + (long long)m * P.hN;
        const double sg = (m & 1) ? -w : w;
        const double2 a = h[rp], b = h[P.hN - 1 - rp], cN = h[P.hN / 2 - 1 - rp], d = h[P.hN / 2 + rp];
        FN = make_double2(w * a.x + sg * b.x, w * a.y + sg * b.y);
        FS = make_double2(w * cN.x + sg * d.x, w * cN.y + sg * d.y);
        return;
    }
    const double2 *Yc = Y + (long long)c * P.ny;
    const double2 a = Yc[r.i0], b = cconj(Yc[r.i1]);
    const double2 xn = cscale(cadd(a, b), 0.5);
    const double2 xs = mul_mi(cscale(csub(a, b), 0.5));  // (a-b)/(2i)
    FN = cmul(xn, r.ph);
    FS = r.hasS ? cmul(xs, r.ph) : make_double2(0.0, 0.0);
}

// RingAtM of ring pair r (pole -> equator) at order m; rw = ring quadrature weights or null (unit weights)
__device__ inline RingAtM ring_at_m_of(const PlanDev &P, int r, int m, const double *__restrict__ rw)
{
    RingAtM q;
    const bool live = r < P.nrp;
    const double w = live ? (rw ? rw[r] : 1.0) * P.wnorm : 0.0;
    q.ph = make_double2(w, 0.0);
    q.i0 = q.i1 = 0;
    q.hasS = 0;
    if (live && !P.hsrc) {
        const int n = P.nsub[r], nphi = 4 * n;
        const int mm = m % nphi, mc = (nphi - mm) % nphi;
        q.i0 = P.startN[r] + (long long)(mm & 3) * n + (mm >> 2);  // Z[mm], X[4k+r] = Y_r[k]
        q.i1 = P.startN[r] + (long long)(mc & 3) * n + (mc >> 2);
        if (P.shifted[r]) q.ph = cscale(expipi(-(double)(m % (2 * nphi)) / (double)nphi), w);
        q.hasS = P.startS[r] >= 0;
    }
    return q;
}

// In-LDS FFT drivers on the padded buffer of hx_fft_core.h (element e in slot lds_slot(e)): fused radix-2^K passes, K <= 4,
// schedule fft_sched_k(log2 M, pass) (tests/csrc/test_fft_core.cpp runs this exact schedule on the host).  Thread gt of the gn threads
// that share one transform; EVERY thread of the block must call (one __syncthreads per pass), all with the same M.
template <class TW>
__device__ __forceinline__ void lds_fft_pass_dif(double2 *buf, int M, int K, int h, int gt, int gn, TW tw, int twN)
{
    switch (K) {
    case 4:
_Pragma("unroll 1")
        for (int i = gt; i < (M >> 4); i += gn) dif_pass_butterfly<4>(buf, i, h, tw, twN); break;
    case 3:
_Pragma("unroll 1")
        for (int i = gt; i < (M >> 3); i += gn) dif_pass_butterfly<3>(buf, i, h, tw, twN); break;
    case 2:
_Pragma("unroll 1")
        for (int i = gt; i < (M >> 2); i += gn) dif_pass_butterfly<2>(buf, i, h, tw, twN); break;
    default: for (int i = gt; i < (M >> 1); i += gn) dif_pass_butterfly<1>(buf, i, h, tw, twN); break;
    }
}
template <class TW>
__device__ __forceinline__ void lds_fft_pass_dit_inv(double2 *buf, int M, int K, int h, int gt, int gn, TW tw, int twN)
{
    switch (K) {
    case 4:
_Pragma("unroll 1")
        for (int i = gt; i < (M >> 4); i += gn) dit_inv_pass_butterfly<4>(buf, i, h, tw, twN); break;
    case 3:
_Pragma("unroll 1")
        for (int i = gt; i < (M >> 3); i += gn) dit_inv_pass_butterfly<3>(buf, i, h, tw, twN); break;
    case 2:
_Pragma("unroll 1")
        for (int i = gt; i < (M >> 2); i += gn) dit_inv_pass_butterfly<2>(buf, i, h, tw, twN); break;
    default: for (int i = gt; i < (M >> 1); i += gn) dit_inv_pass_butterfly<1>(buf, i, h, tw, twN); break;
    }
}
// forward passes first .. last-1 of the schedule (skip_last: the caller fuses the h = 1 pass with what follows)
template <class TW>
__device__ __forceinline__ void lds_fft_dif(double2 *buf, int M, TW tw, int twN, int gt, int gn, bool skip_last = false)
{
    const int p = ilog2(M), np = fft_sched_np(p);
    int h = M;
    for (int a = 0; a < np - (skip_last ? 1 : 0); ++a) {
        const int K = fft_sched_k(p, a);
        h >>= K;
        lds_fft_pass_dif(buf, M, K, h, gt, gn, tw, twN);
        __syncthreads();
    }
}
template <class TW>
__device__ __forceinline__ void lds_fft_dit_inv(double2 *buf, int M, TW tw, int twN, int gt, int gn, bool skip_first = false)
{
    const int p = ilog2(M), np = fft_sched_np(p);
    int h = 1;
    for (int a = np - 1; a >= 0; --a) {
        const int K = fft_sched_k(p, a);
        if (!(skip_first && a == np - 1)) {
            lds_fft_pass_dit_inv(buf, M, K, h, gt, gn, tw, twN);
            __syncthreads();
        }
        h <<= K;
    }
}
template <class TW>
__device__ __forceinline__ void lds_fft_dif(double2 *buf, int M, TW tw, int twN) { lds_fft_dif(buf, M, tw, twN, (int)threadIdx.x, (int)blockDim.x); }
template <class TW>
__device__ __forceinline__ void lds_fft_dit_inv(double2 *buf, int M, TW tw, int twN) { lds_fft_dit_inv(buf, M, tw, twN, (int)threadIdx.x, (int)blockDim.x); }

// Fills the factored twiddle tables of TwFactored from the plan's full table (twN/2 entries).
// LDS: hi[twN/128], lo[64].  The caller synchronises before the first butterfly.
constexpr int TW_HI_MAX = 128;  // twN <= 16384
__device__ inline TwFactored load_tw_factored(double2 *hi, double2 *lo, const double2 *__restrict__ tw, int twN)
{
    for (int i = threadIdx.x; i < (twN >= 128 ? twN / 128 : 1); i += blockDim.x) hi[i] = tw[i * 64];  // hi[0] = 1
    for (int i = threadIdx.x; i < 64; i += blockDim.x) lo[i] = tw[i];
    TwFactored f;
    f.hi = hi;
    f.lo = lo;
    return f;
}

#endif

template <class T>
inline int upload(DevBuf &b, const std::vector<T> &v)
{
    HX_TRY(b.alloc(sizeof(T) * (v.size() > 0 ? v.size() : 1)));
    if (!v.empty()) HX_HIP(hipMemcpy(b.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return HX_OK;
}

}  // namespace hx

struct hx_plan {
    int nside = 0, lmax = 0, max_comp = 0;
    int nrp = 0, nrp_pad = 0, twN = 1;
    int fft_cap = 8192;   // longest in-LDS FFT of this plan
    int last_chunks = 0;  // m-chunks of the most recent analysis sweep (hx_plan_last_chunks)
    int eqN = 0;          // equiangular plan (nside == 0): points of the full circle in theta
    double wnorm = 0.0;
    const double2 *hsrc = nullptr;  // ring spectra of the current point-transform call (equiangular plan)
    long long hsrc_stride = 0;
    const double4 *const *nssrc = nullptr;  // m-sharded route: device array of per-component mode blocks of the current call
    int ns_m0 = 0;
    // orders the analysis sweeps cover: m_lo, m_lo + m_step, ... < m_hi (m_hi < 0 = lmax + 1).  The m-sharded route gives rank q of N
    // the orders q, q + N, ...: every rank gets every size of work-group, and as many of them as a single GPU's launch / N
    int m_lo = 0, m_hi = -1, m_step = 1;
    long long npix = 0, ny = 0, nlm = 0;
    size_t lds_fft = 0;
    hx::DevBuf z, omz, sth, rwdef, nsub, shifted, startN, startS, bhat_off, tw, bhat, mfac, kfac2, rec0, rec2, cn0, al0, cn2, al2;
    std::vector<double> h_sth, h_z;
    std::vector<int> h_nsub;
    std::vector<long long> h_startN, h_startS;   // first pixel of the northern / southern ring of a pair (-1: the equator has no southern ring)
    struct TaskSet {
        bool built = false;
        std::vector<hx::LegTask> tasks;       // ordered by m; tasks of one m contiguous
        std::vector<hx::MTasks> of_m;
        std::vector<long long> rows_before_m; // partial rows of all tasks with smaller m (size lmax+2)
        std::vector<long long> arow;          // the same with ONE span of rows per m (pipelined kernel: ring groups summed in place)
        hx::DevBuf d_tasks, d_of_m, d_arow;
    } ts[8];  // spin 0, spin 2, spin 0 with half-size work-groups, spin 2 with one ring set per wave (4 ring blocks per task),
       // spin 2 / spin 0 on the vector unit (hx_legendre_valu.hip: 2 R ring blocks per task), spin 2 / spin 0 synthesis of several
       // maps per sweep (8 ring blocks per task)
    struct FftClass { int M, first, count, big; };
    std::vector<FftClass> fft_classes;   // ring pairs grouped by in-LDS FFT length
    hx::DevBuf fft_rp_list;
    hx::DevBuf fft_desc;                 // RingDesc of every entry of fft_rp_list (one 32-byte read per work item instead of two dependent ones)
    std::vector<int> h_fft_rp_list;      // (host copy: ring pairs in DESCENDING order within a class)
    hx::DevBuf Y, F, partial, d_dbg, resid_maps, pw_sym;  // (F doubles as the scratch of a synthesis: ring modes + ring spectra)
    hx::DevBuf syn_mlim0, syn_mlim2;  // per ring pair: the highest m its rows of Fv are written for (task pruning by blocks of 32 ring pairs)
    hx::DevBuf syn_tab, syn_boff0, syn_boff2;  // batched synthesis on the matrix unit (hx_synth_duo.hip): B-operand table of a sweep, first table block of every m
    const double *pw_checked = nullptr;       // the pixel-weight array of the current call that pw_mode describes
    int pw_mode = 0;                          // 1: one weight per pixel; 2: the array repeats over the quadrants of every ring and from north to south (healpy's weights)
    static constexpr int NSTAGE = 3;          // staging buffers of the upload pipeline (hx_map2alm_multi / _list; hx_map2alm of host maps is one job of it)
    hx::DevBuf stage[NSTAGE];                 // maps of one sweep each: host input uploaded sweep by sweep
    hipEvent_t stage_done[NSTAGE] = {nullptr, nullptr, nullptr};   // the transform of the sweep in buffer i has been queued behind this
    static constexpr int NUNIT_EV = 4;        // upload events in flight: a unit (a sweep, or a slab of rings of a streamed sweep) waits for its own before the next but one is recorded
    hipEvent_t unit_up[NUNIT_EV] = {nullptr, nullptr, nullptr, nullptr};
    hx::PlanDev dev() const;
};

namespace hx {
// hx_sht.hip
hx_plan *plan_create_equiangular(int N, int lmax);
int ensure_rec2(hx_plan *pl);
int launch_ring_subdft_maps(hx_plan *pl, int nb, const double *d_maps, const double *d_pw, double2 *Y, int rp_lo = 0, int rp_hi = 0x7fffffff);  // ring pairs [rp_lo, rp_hi) only
int classify_pixel_weights(hx_plan *pl, const double *d_pw);
// hx_analysis.hip
int build_tasks(hx_plan *pl, int spin);
int analysis_batch(hx_plan *pl, int spin, int nb, const double *d_maps, double2 *d_alms, const double *d_rw,
                   const double *d_pw, const double *d_fl, int add);
int analysis_max_comp(int spin);
int analysis_next_batch(int spin, int remaining);
int analysis_max_batch(int spin, int ncomp);
// One analysis sweep whose rings ARRIVE IN SLABS (hx_map2alm_multi on host maps): the ring FFT, the operand pass and the ring groups of the
// Legendre kernel run slab by slab behind the upload, the change of layout once at the end -- the same sums in the same order as the sweep
// over resident maps (bit-identical alms)
struct StreamSweep {
    hx_plan *pl = nullptr;
    int spin = 0, nb = 0, nslab = 0;
    const double *d_maps = nullptr, *d_rw = nullptr, *d_pw = nullptr, *d_fl = nullptr;
    double2 *d_alms = nullptr;
    std::vector<int> rp_edge;   // slab k = ring pairs [rp_edge[k], rp_edge[k + 1]): whole 32-ring-pair blocks, about equal numbers of pixels
    DevBuf d_of_m;              // [nslab][lmax + 1]: the ring groups of order m that are complete with slab k
};
bool analysis_can_stream(hx_plan *pl, int spin, int nb);
int analysis_stream_plan(hx_plan *pl, int spin, int nb, int nslab, StreamSweep &s);   // slab edges, task tables, scratch (grown, never shrunk): before anything of the call is queued
int analysis_stream_start(StreamSweep &s);  // zeroes the accumulation rows (queued on the library stream, in front of slab 0)
int analysis_stream_slab(StreamSweep &s, int k);
int analysis_stream_end(StreamSweep &s);
// hx_legendre_valu.hip: one map (spin 0) / one field (spin 2) per sweep on the FP64 vector unit
int launch_valu_chunk(hx_plan *pl, int spin, hx_plan::TaskSet &ts, int m0, int m1, int c0, const double *d_rw);
int valu_task_blocks(int spin);      // 32-ring-pair blocks per task
int valu_partial_cols(int spin);     // doubles per row of the partial buffer
int valu_operand_doubles(int spin);  // doubles per (m, ring pair) of the operand array
int valu_exec_flops(unsigned long long *v, bool reset);
int valu_tasks(hx_plan *pl, int spin, hx_plan::TaskSet **ts, int blocks = 0);  // (hx_analysis.hip) task set of the vector-unit kernels (blocks: 32-ring-pair blocks per task, 0 = valu_task_blocks), built on first use
int synth_valu_max_units(int spin);                // maps (spin 0) / fields (spin 2) per sweep of the synthesis kernel: 1, 2, .. a power of two
int synth_valu_task_blocks(int spin, int units);   // ring blocks per task of that sweep
int launch_synth_valu(hx_plan *pl, int spin, int units, hx_plan::TaskSet &ts, const double2 *d_alm, double *d_Fv);
// hx_synth_duo.hip: up to 20 maps / 10 fields per sweep on the matrix unit; Fv[m][rp][synth_duo_rowlen] in the lane order of that kernel
int synth_duo_max_units(int spin);
int synth_duo_rowlen(int spin, int units);
size_t synth_duo_table_bytes(hx_plan *pl, int spin, int units);
int launch_synth_duo(hx_plan *pl, int spin, int units, hx_plan::TaskSet &ts, const double2 *d_alm, double *d_tab, double *d_Fv);
int synth_duo_tasks(hx_plan *pl, int spin, hx_plan::TaskSet **ts);  // (hx_analysis.hip) tasks of 8 (spin 0) / 4 (spin 2) ring blocks: the ring groups of k_legendre_duo  // alm -> Fv[m][rp][4 per component]  // FP64 vector flops executed by the vector-unit kernels since the last reset
}  // namespace hx
