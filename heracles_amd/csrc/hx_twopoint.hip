// hx_twopoint.hip -- all-pairs alm x alm -> Cl reduction (HBM-bound).
//
// Replaces heracles.twopoint.alm2cl (heracles/twopoint.py:63-101) for a list of component
// pairs:   cl_l = [Re a_l0 Re b_l0 + 2 sum_{m=1..l} Re(a_lm conj b_lm)] / (2l+1)
// (the imaginary part of m=0 is ignored, twopoint.py:88).
//
// Layout: alm is m-major, so for fixed m consecutive l are contiguous: lanes map to l
// (1 KiB coalesced per wave-load), the sum over m runs in-lane; the waves of a
// workgroup split m round-robin and are combined through LDS in a fixed order, so results
// are bit-reproducible run to run.  Components are tiled T x T so every alm value loaded is
// used for T products (the reference re-reads each alm once per partner).
#include "hx_common.h"

namespace hx {

constexpr int CL_T = 6;        // component tile edge (4 -> 6: each alm is fetched ncomp/6 times; 13.4 -> 9.7 ms, 8 gains no more)
constexpr int CL_WAVES = 3;    // waves per workgroup (m split); 3 x 36 x 64 doubles of LDS
constexpr int CL_LB = 64;      // l values per workgroup

struct ClTile {
    int i0, j0;               // first component of the tile along each axis
    int out[CL_T * CL_T];     // pair index of (i0+a, j0+b) or -1
};

__global__ __launch_bounds__(CL_WAVES * 64) void k_alm2cl_tiles(
    const double2 *const *__restrict__ comp, const int *__restrict__ comp_lmax, int ncomp,
    const ClTile *__restrict__ tiles, int ntiles, int lmax_out, int nlblk, double *__restrict__ cls, int m_lo, int m_hi, int m_step)
{
    __shared__ double red[CL_WAVES][CL_T * CL_T][CL_LB];
    // Work-groups are dispatched to the 8 XCDs round-robin: XCD x = blockIdx % 8 walks the l-blocks 8 k + x, and for each of them
    // ALL tiles back to back -- the tiles of one l-block read the same rows of the same alms at about the same time, so a row comes
    // from HBM once and from that XCD's L2 for the other tiles (tile-major order re-read every alm ncomp / 6 times from HBM:
    // 50 GB for the 9 GB of the bench).  Heavy (high-l) blocks first.
    const int xcd = blockIdx.x & 7, seq = blockIdx.x >> 3;
    const int lblk = nlblk - 1 - ((seq / ntiles) * 8 + xcd);
    if (lblk < 0) return;
    const ClTile tile = tiles[seq % ntiles];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int l = lblk * CL_LB + lane;
    const int lhi = min(lblk * CL_LB + CL_LB - 1, lmax_out);

    // A tile may hold components whose own band limit is below lmax_out (mixed-lmax alms: only the
    // components of REQUESTED pairs are checked against lmax_out by the host).  Their loads are
    // predicated on (m, l) lying inside their own triangle -- the index would otherwise leave the buffer.
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

    // the orders summed: m_lo, m_lo + m_step, ... < m_hi (everything for hx_alm2cl_pairs; one rank's share on the m-sharded route)
    for (int m = m_lo + w * m_step; m <= min(lhi, m_hi - 1); m += CL_WAVES * m_step) {
        if (m <= l && l <= lmax_out) {
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
    }
#pragma unroll
    for (int t = 0; t < CL_T * CL_T; ++t) red[w][t][lane] = acc[t];
    __syncthreads();
    // fixed-order combine; thread (w, lane) finishes pairs t = w, w + CL_WAVES, ...
    if (l <= lmax_out) {
        for (int t = w; t < CL_T * CL_T; t += CL_WAVES) {
            int o = tile.out[t];
            if (o < 0) continue;
            double s = 0.0;
#pragma unroll
            for (int ww = 0; ww < CL_WAVES; ++ww) s += red[ww][t][lane];
            cls[(long long)o * (lmax_out + 1) + l] = s / (2.0 * l + 1.0);
        }
    }
}

}  // namespace hx

using namespace hx;

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
        int bi = pair_i[p] / CL_T, bj = pair_j[p] / CL_T;
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
        int slot = (pair_i[p] - t.i0) * CL_T + (pair_j[p] - t.j0);
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
    DevBuf d_ptrs, d_lmax, d_tiles;
    HX_TRY(d_ptrs.alloc(sizeof(void *) * ncomp));
    HX_TRY(d_lmax.alloc(sizeof(int) * ncomp));
    HX_TRY(d_tiles.alloc(sizeof(ClTile) * tiles.size()));
    HX_HIP(hipMemcpyAsync(d_ptrs.p, ptrs.data(), sizeof(void *) * ncomp, hipMemcpyHostToDevice, st));
    HX_HIP(hipMemcpyAsync(d_lmax.p, lmax_i, sizeof(int) * ncomp, hipMemcpyHostToDevice, st));
    HX_HIP(hipMemcpyAsync(d_tiles.p, tiles.data(), sizeof(ClTile) * tiles.size(), hipMemcpyHostToDevice, st));
    OutView out;
    HX_TRY(out.bind(cls, sizeof(double) * (size_t)npairs * (lmax_out + 1)));

    const int nlblk = (lmax_out + CL_LB) / CL_LB;
    {
        ProfScope ps("alm2cl");
        const unsigned nlb8 = (unsigned)((nlblk + 7) / 8);  // l-blocks per XCD
        hipLaunchKernelGGL(k_alm2cl_tiles, dim3(nlb8 * 8u * (unsigned)tiles.size()), dim3(CL_WAVES * 64), 0, st,
                           d_ptrs.as<const double2 *>(), d_lmax.as<int>(), ncomp, d_tiles.as<ClTile>(), (int)tiles.size(),
                           lmax_out, nlblk, out.as<double>(), m0, m1, mstep);
    }
    HX_HIP(hipGetLastError());
    HX_TRY(out.finish());
    // temporaries (views, tables) are freed on return: make sure the kernel is done
    HX_HIP(hipStreamSynchronize(st));
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
