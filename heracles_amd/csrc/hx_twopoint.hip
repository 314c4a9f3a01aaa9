// hx_twopoint.hip -- all-pairs alm x alm -> Cl reduction (HBM-bound).
//
// Replaces heracles.twopoint.alm2cl (heracles/twopoint.py:63-101) for a list of component
// pairs:   cl_l = [Re a_l0 Re b_l0 + 2 sum_{m=1..l} Re(a_lm conj b_lm)] / (2l+1)
// (the imaginary part of m=0 is ignored, twopoint.py:88).
//
// Layout: alm is m-major, so for fixed m consecutive l are contiguous: lanes map to l
// (1 KiB coalesced per wave-load), the sum over m runs in-lane.  Components are tiled 6 x 6, one wave per tile, 16 tiles per
// work-group walking the orders together, so that a row is fetched from HBM once per l-block; the orders are split over a few
// work-groups whose partial sums are added in a fixed order (bit-reproducible run to run).
#include <algorithm>

#include "hx_common.h"

namespace hx {

constexpr int CL_T = 6;        // component tile edge: 36 accumulators per lane
constexpr int CL_TW = 8;       // tiles (= waves) per work-group (16 would leave each wave 128 registers: 90 spilled)
constexpr int CL_LB = 64;      // l values per work-group
constexpr int CL_SYNC = 8;     // orders between two work-group barriers (keeps the waves on the same rows)

struct ClTile {
    int i0, j0;               // first component of the tile along each axis
    int out[CL_T * CL_T];     // pair index of (i0+a, j0+b) or -1
};

// A work-group = one block of 64 l x one share of the orders m x up to 8 component tiles, ONE WAVE PER TILE: the waves walk the
// orders together (a barrier every CL_SYNC orders), so a row of an alm -- 64 consecutive l of one m, 1 KiB -- is fetched from HBM by
// the first wave that needs it and comes from the CU's L1 / the XCD's L2 for the other tiles that share the component.  Every
// (l, m) of every component belongs to exactly one work-group: HBM traffic = the alms once (30 GB -> 9 GB for the bench's 30
// components; tiles as independent work-groups re-read each alm ncomp / 6 times: 50 GB, 30 GB with XCD-aware ordering).
// The shares of the orders (msplit) are summed in a fixed order by k_alm2cl_finish: bitwise repeatable.
__global__ __launch_bounds__(CL_TW * 64) void k_alm2cl_rows(
    const double2 *const *__restrict__ comp, const int *__restrict__ comp_lmax, int ncomp,
    const ClTile *__restrict__ tiles, int ntiles, int ngroups, int nmsplit, int lmax_out, int nlblk, double *__restrict__ part, long long part_stride,
    int m_lo, int m_hi, int m_step)
{
    // heavy (high-l) blocks first.  Work-groups go to the 8 XCDs round-robin (blockIdx % 8): the tile groups of one (l-block, share of
    // the orders) read the same rows of the same components, so they are given block indices 8 apart -- the same XCD, dispatched
    // together -- and find each other's rows in that XCD's L2 (round 5; before, they sat nmsplit apart on different XCDs and every
    // group fetched its rows from HBM: 2 x the alms for the bench's 15 tiles)
    const int nq = nlblk * nmsplit;
    const int q = ((int)blockIdx.x / (8 * ngroups)) * 8 + ((int)blockIdx.x & 7), g = ((int)blockIdx.x >> 3) % ngroups;
    if (q >= nq) return;
    const int ms = q % nmsplit, lblk = nlblk - 1 - q / nmsplit;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ti = g * CL_TW + w;
    const bool have = ti < ntiles;
    const ClTile tile = tiles[have ? ti : 0];
    const int l = lblk * CL_LB + lane;
    const int lhi = min(lblk * CL_LB + CL_LB - 1, lmax_out);
    // A tile may hold components whose own band limit is below lmax_out (mixed-lmax alms: only the components of REQUESTED pairs
    // are checked against lmax_out by the host).  Their loads are predicated on (m, l) lying inside their own triangle.
    const double2 *pa[CL_T], *pb[CL_T];
    long long La[CL_T], Lb[CL_T];
#pragma unroll
    for (int t = 0; t < CL_T; ++t) {
        int ia = min(tile.i0 + t, ncomp - 1), ib = min(tile.j0 + t, ncomp - 1);
        pa[t] = comp[ia]; La[t] = comp_lmax[ia];
        pb[t] = comp[ib]; Lb[t] = comp_lmax[ib];
    }
    double acc[CL_T * CL_T];
#pragma unroll
    for (int t = 0; t < CL_T * CL_T; ++t) acc[t] = 0.0;
    // the orders of this share: m_lo + (k nmsplit + ms) m_step, k = 0, 1, ...  (< m_hi, <= the largest l of the block)
    const int mend = min(lhi, m_hi - 1);
    int k = 0;
    for (int m = m_lo + ms * m_step; m <= mend; m += nmsplit * m_step, ++k) {
        if (have && m <= l && l <= lmax_out) {
            double2 a[CL_T], b[CL_T];
#pragma unroll
            for (int t = 0; t < CL_T; ++t) {
                a[t] = l <= La[t] ? pa[t][(long long)m * (2 * La[t] + 1 - m) / 2 + l] : make_double2(0.0, 0.0);
                b[t] = l <= Lb[t] ? pb[t][(long long)m * (2 * Lb[t] + 1 - m) / 2 + l] : make_double2(0.0, 0.0);
            }
            const double wgt = m == 0 ? 1.0 : 2.0;
#pragma unroll
            for (int s = 0; s < CL_T; ++s)
#pragma unroll
                for (int t = 0; t < CL_T; ++t) {
                    double p = a[s].x * b[t].x;
                    if (m != 0) p = fma(a[s].y, b[t].y, p);
                    acc[s * CL_T + t] = fma(wgt, p, acc[s * CL_T + t]);
                }
        }
        if ((k & (CL_SYNC - 1)) == CL_SYNC - 1) __syncthreads();  // (the trip count is the same for every wave of the group)
    }
    if (have && l <= lmax_out) {
#pragma unroll
        for (int t = 0; t < CL_T * CL_T; ++t) {
            const int o = tile.out[t];
            if (o >= 0) part[(long long)ms * part_stride + (long long)o * (lmax_out + 1) + l] = acc[t];
        }
    }
}

// cls[pair][l] = (sum over the shares, in order) / (2 l + 1)
__global__ __launch_bounds__(256) void k_alm2cl_finish(const double *__restrict__ part, long long part_stride, int nmsplit, long long n, int lmax_out,
                                                        double *__restrict__ cls)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int q = 0; q < nmsplit; ++q) s += part[(long long)q * part_stride + i];
    cls[i] = s / (2.0 * (double)(i % (lmax_out + 1)) + 1.0);
}

}  // namespace hx

using namespace hx;

namespace hx {
struct Alm2clScratch {
    DevBuf ptrs, lmax, tiles, part, out;
};
static Alm2clScratch *g_alm2cl_scratch = nullptr;  // heap, never destroyed at exit (the HIP runtime may be gone by then)
static Alm2clScratch &alm2cl_scratch()
{
    if (!g_alm2cl_scratch) g_alm2cl_scratch = new Alm2clScratch;
    return *g_alm2cl_scratch;
}
void alm2cl_drop_cache()
{
    if (!g_alm2cl_scratch) return;
    g_alm2cl_scratch->ptrs.release(); g_alm2cl_scratch->lmax.release(); g_alm2cl_scratch->tiles.release();
    g_alm2cl_scratch->part.release(); g_alm2cl_scratch->out.release();
}
}  // namespace hx

extern "C" int hx_alm2cl_pairs_range(int ncomp, const int *lmax_i, const double *const *alms, int lmax_out, int npairs, const int *pair_i,
                                     const int *pair_j, int m0, int m1, int mstep, double *cls);

extern "C" int hx_alm2cl_pairs(int ncomp, const int *lmax_i, const double *const *alms,
                               int lmax_out, int npairs, const int *pair_i, const int *pair_j,
                               double *cls)
{
    return hx_alm2cl_pairs_range(ncomp, lmax_i, alms, lmax_out, npairs, pair_i, pair_j, 0, lmax_out + 1, 1, cls);
}

// The same sum restricted to the orders m0, m0 + mstep, ... < m1 (still divided by 2l + 1): the partial spectra of disjoint sets add
// up to hx_alm2cl_pairs -- what a rank of the m-sharded route contributes before the all-reduce.
extern "C" int hx_alm2cl_pairs_range(int ncomp, const int *lmax_i, const double *const *alms, int lmax_out, int npairs, const int *pair_i,
                                     const int *pair_j, int m0, int m1, int mstep, double *cls)
{
    HX_TRY(ensure_ready());
    if (m0 < 0 || m1 < m0 || mstep < 1) return fail(HX_ERR_ARG, "hx_alm2cl_pairs_range: bad orders %d, %d + %d, ... < %d", m0, m0, mstep, m1);
    if (ncomp <= 0 || npairs < 0 || !lmax_i || !alms || !pair_i || !pair_j || !cls || lmax_out < 0)
        return fail(HX_ERR_ARG, "hx_alm2cl_pairs: bad argument");
    if (npairs == 0) return HX_OK;
    for (int c = 0; c < ncomp; ++c)
        if (!alms[c] || lmax_i[c] < 0) return fail(HX_ERR_ARG, "hx_alm2cl_pairs: component %d invalid", c);
    for (int p = 0; p < npairs; ++p) {
        if (pair_i[p] < 0 || pair_i[p] >= ncomp || pair_j[p] < 0 || pair_j[p] >= ncomp)
            return fail(HX_ERR_ARG, "hx_alm2cl_pairs: pair %d out of range", p);
        if (lmax_i[pair_i[p]] < lmax_out || lmax_i[pair_j[p]] < lmax_out)
            return fail(HX_ERR_ARG, "hx_alm2cl_pairs: lmax_out exceeds lmax of pair %d", p);
    }
    hipStream_t st = rt().stream;

    // stage components (device pointers are used in place)
    std::vector<InView> views(ncomp);
    std::vector<const double2 *> ptrs(ncomp);
    for (int c = 0; c < ncomp; ++c) {
        size_t nlm = (size_t)(lmax_i[c] + 1) * (lmax_i[c] + 2) / 2;
        HX_TRY(views[c].bind(alms[c], nlm * sizeof(double2)));
        ptrs[c] = views[c].as<double2>();
    }
    // tiles
    const int nb = (ncomp + CL_T - 1) / CL_T;
    std::map<std::pair<int, int>, int> tile_of;
    std::vector<ClTile> tiles;
    for (int p = 0; p < npairs; ++p) {
        // Cl(a, b) = Cl(b, a) term by term (the products commute: bit-identical), so a pair is filed under the tile with block(i) <=
        // block(j) whichever way it was given: a buffer that holds its spin-2 components in front of its spin-0 ones (distributed.py
        // since round 4) hands the spin-0 x spin-2 pairs over as (high, low) and would otherwise open 19 tiles instead of 15
        int ci = pair_i[p], cj = pair_j[p];
        if (ci / CL_T > cj / CL_T) std::swap(ci, cj);
        int bi = ci / CL_T, bj = cj / CL_T;
        auto key = std::make_pair(bi, bj);
        auto it = tile_of.find(key);
        if (it == tile_of.end()) {
            ClTile t;
            t.i0 = bi * CL_T;
            t.j0 = bj * CL_T;
            for (int k = 0; k < CL_T * CL_T; ++k) t.out[k] = -1;
            tiles.push_back(t);
            it = tile_of.emplace(key, (int)tiles.size() - 1).first;
        }
        ClTile &t = tiles[it->second];
        int slot = (ci - t.i0) * CL_T + (cj - t.j0);
        if (t.out[slot] >= 0) {
            // duplicate pair in the list: give it its own tile so each output is written
            ClTile d;
            d.i0 = t.i0; d.j0 = t.j0;
            for (int k = 0; k < CL_T * CL_T; ++k) d.out[k] = -1;
            d.out[slot] = p;
            tiles.push_back(d);
        } else
            t.out[slot] = p;
    }
    (void)nb;
    // tiles of one work-group share components when they are neighbours in (i0, j0) order
    std::stable_sort(tiles.begin(), tiles.end(), [](const ClTile &x, const ClTile &y) { return x.i0 != y.i0 ? x.i0 < y.i0 : x.j0 < y.j0; });
    // tables, partial sums and the staging buffer of a host result are kept between calls (a loop over steps or jackknife regions calls with
    // the same sizes: four hipMalloc / hipFree pairs, one of 70 MB, were 0.6 ms of every call); freed by hx_init on another device
    Alm2clScratch &sc_ = alm2cl_scratch();
    DevBuf &d_ptrs = sc_.ptrs, &d_lmax = sc_.lmax, &d_tiles = sc_.tiles;
    HX_TRY(d_ptrs.alloc(sizeof(void *) * ncomp));
    HX_TRY(d_lmax.alloc(sizeof(int) * ncomp));
    HX_TRY(d_tiles.alloc(sizeof(ClTile) * tiles.size()));
    HX_HIP(hipMemcpyAsync(d_ptrs.p, ptrs.data(), sizeof(void *) * ncomp, hipMemcpyHostToDevice, st));
    HX_HIP(hipMemcpyAsync(d_lmax.p, lmax_i, sizeof(int) * ncomp, hipMemcpyHostToDevice, st));
    HX_HIP(hipMemcpyAsync(d_tiles.p, tiles.data(), sizeof(ClTile) * tiles.size(), hipMemcpyHostToDevice, st));
    // (OutView with the kept buffer as its staging area)
    const size_t out_bytes = sizeof(double) * (size_t)npairs * (lmax_out + 1);
    double *out_dev = cls;
    const bool out_on_host = !is_device_ptr(cls);
    if (out_on_host) {
        HX_TRY(sc_.out.alloc(out_bytes));
        out_dev = sc_.out.as<double>();
    }

    const int nlblk = (lmax_out + CL_LB) / CL_LB;
    const int ngroups = ((int)tiles.size() + CL_TW - 1) / CL_TW;
    // shares of the orders per (l-block, tile group): enough work-groups for ~2.5 rounds of the CUs (one work-group fills a CU)
    int nmsplit = (int)((2.5 * rt().cus) / ((double)nlblk * ngroups) + 0.5);
    nmsplit = std::max(1, std::min(nmsplit, 8));
    const long long nout = (long long)npairs * (lmax_out + 1);
    DevBuf &d_part = sc_.part;
    HX_TRY(d_part.alloc(sizeof(double) * (size_t)nout * nmsplit));
    HX_HIP(hipMemsetAsync(d_part.p, 0, sizeof(double) * (size_t)nout * nmsplit, st));  // (shares without orders write nothing)
    {
        ProfScope ps("alm2cl");
        hipLaunchKernelGGL(k_alm2cl_rows, dim3((unsigned)((nlblk * nmsplit + 7) / 8 * 8 * ngroups)), dim3(CL_TW * 64), 0, st,
                           d_ptrs.as<const double2 *>(), d_lmax.as<int>(), ncomp, d_tiles.as<ClTile>(), (int)tiles.size(), ngroups, nmsplit,
                           lmax_out, nlblk, d_part.as<double>(), nout, m0, m1, mstep);
        hipLaunchKernelGGL(k_alm2cl_finish, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, st, d_part.as<double>(), nout, nmsplit, nout,
                           lmax_out, out_dev);
    }
    HX_HIP(hipGetLastError());
    if (out_on_host) HX_TRY(copy_d2h(cls, out_dev, out_bytes));  // synchronous on return
    // temporaries (views of host components) are freed on return: make sure the kernel is done
    HX_HIP(hipStreamSynchronize(st));
    if (sc_.part.bytes + sc_.out.bytes > ((size_t)512 << 20)) alm2cl_drop_cache();  // (an unusually large call does not keep its buffers)
    return HX_OK;
}

// ---- alm re-pack between band limits (DiscreteMapper.resample, heracles/ducc.py:145-162) ---------
namespace hx {
__global__ __launch_bounds__(256) void k_alm_resample(int lmax_in, int lmax_out, long long nlm_in, long long nlm_out,
                                                      const double2 *__restrict__ in, double2 *__restrict__ out)
{
    // one block per (m of the OUTPUT layout, component): rows l = m..lmax_out, zero beyond the input
    const int m = blockIdx.x;
    const double2 *src = in + (long long)blockIdx.y * nlm_in;
    double2 *dst = out + (long long)blockIdx.y * nlm_out;
    const long long bo = (long long)m * (2 * lmax_out + 1 - m) / 2, bi = (long long)m * (2 * lmax_in + 1 - m) / 2;
    for (int l = m + threadIdx.x; l <= lmax_out; l += blockDim.x)
        dst[bo + l] = (m <= lmax_in && l <= lmax_in) ? src[bi + l] : make_double2(0.0, 0.0);
}
}  // namespace hx

extern "C" int hx_alm_resample(int lmax_in, int lmax_out, int ncomp, const double *alm_in, double *alm_out)
{
    using namespace hx;
    HX_TRY(ensure_ready());
    if (lmax_in < 0 || lmax_out < 0 || ncomp < 0 || (ncomp > 0 && (!alm_in || !alm_out)))
        return fail(HX_ERR_ARG, "hx_alm_resample: bad arguments");
    if (ncomp == 0) return HX_OK;
    const long long ni = (long long)(lmax_in + 1) * (lmax_in + 2) / 2, no = (long long)(lmax_out + 1) * (lmax_out + 2) / 2;
    InView vin;
    OutView vout;
    HX_TRY(vin.bind(alm_in, sizeof(double2) * ni * ncomp));
    HX_TRY(vout.bind(alm_out, sizeof(double2) * no * ncomp));
    hipLaunchKernelGGL(k_alm_resample, dim3(lmax_out + 1, ncomp), dim3(256), 0, rt().stream, lmax_in, lmax_out, ni, no,
                       vin.as<double2>(), vout.as<double2>());
    HX_HIP(hipGetLastError());
    HX_TRY(vout.finish());
    return finish_call();
}


// ---- jackknife helpers (heracles/dices/jackknife.py:222-233, :253-263): delete-k alms and region maps in HBM ----------
namespace hx {
// out = full - sum_i subs[i]  (complex values as interleaved doubles; nsub <= 4)
struct SubList {
    const double2 *p[4];
};
__global__ __launch_bounds__(256) void k_alm_subtract(long long n, const double2 *__restrict__ full, int nsub, SubList subs,
                                                      double2 *__restrict__ out)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long st = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        double2 v = full[i];
        for (int k = 0; k < nsub; ++k) {
            const double2 s = subs.p[k][i];
            v.x -= s.x;
            v.y -= s.y;
        }
        out[i] = v;
    }
}

// out[c][p] = maps[c][p] if region[p] == k else 0  (region map as doubles, the type the reference compares with float(jk))
__global__ __launch_bounds__(256) void k_region_maps(long long npix, int ncomp, const double *__restrict__ maps,
                                                     const double *__restrict__ region, double k, double *__restrict__ out)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long st = (long long)gridDim.x * blockDim.x;
    for (; i < npix; i += st) {
        const bool in = region[i] == k;
        for (int c = 0; c < ncomp; ++c) out[(long long)c * npix + i] = in ? maps[(long long)c * npix + i] : 0.0;
    }
}
}  // namespace hx

// out = full - sum of nsub arrays, n complex values each (host or device pointers; out may alias full)
extern "C" int hx_alm_subtract(int64_t n, const double *full, int nsub, const double *const *subs, double *out)
{
    using namespace hx;
    HX_TRY(ensure_ready());
    if (n < 0 || nsub < 0 || nsub > 4 || (n > 0 && (!full || !out || (nsub > 0 && !subs))))
        return fail(HX_ERR_ARG, "hx_alm_subtract: bad arguments (at most 4 arrays to subtract)");
    if (n == 0) return HX_OK;
    InView vf, vs[4];
    OutView vo;
    HX_TRY(vf.bind(full, sizeof(double2) * n));
    SubList sl;
    for (int k = 0; k < 4; ++k) sl.p[k] = nullptr;
    for (int k = 0; k < nsub; ++k) {
        if (!subs[k]) return fail(HX_ERR_ARG, "hx_alm_subtract: null array %d", k);
        HX_TRY(vs[k].bind(subs[k], sizeof(double2) * n));
        sl.p[k] = vs[k].as<double2>();
    }
    HX_TRY(vo.bind(out, sizeof(double2) * n));
    hipLaunchKernelGGL(k_alm_subtract, dim3(4096), dim3(256), 0, rt().stream, (long long)n, vf.as<double2>(), nsub, sl, vo.as<double2>());
    HX_HIP(hipGetLastError());
    HX_TRY(vo.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

// maps restricted to jackknife region k: out[c][p] = maps[c][p] where region[p] == k, else 0
extern "C" int hx_region_maps(int64_t npix, int ncomp, const double *maps, const double *region, double k, double *out)
{
    using namespace hx;
    HX_TRY(ensure_ready());
    if (npix < 0 || ncomp < 0 || (npix > 0 && ncomp > 0 && (!maps || !region || !out))) return fail(HX_ERR_ARG, "hx_region_maps: bad arguments");
    if (npix == 0 || ncomp == 0) return HX_OK;
    InView vm, vr;
    OutView vo;
    HX_TRY(vm.bind(maps, sizeof(double) * npix * ncomp));
    HX_TRY(vr.bind(region, sizeof(double) * npix));
    HX_TRY(vo.bind(out, sizeof(double) * npix * ncomp));
    hipLaunchKernelGGL(k_region_maps, dim3(4096), dim3(256), 0, rt().stream, (long long)npix, ncomp, vm.as<double>(), vr.as<double>(), k, vo.as<double>());
    HX_HIP(hipGetLastError());
    HX_TRY(vo.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}
