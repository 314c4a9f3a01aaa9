// hx_mshard.hip -- the two halves of hx_map2alm as separate entry points, for the m-sharded multi-GPU route (SURVEY.md 8e,
// "lower-traffic alternative"; the loops it replaces: heracles/mapping.py:151-172 and heracles/twopoint.py:198-215).
//
// A fixed job of a few tens of maps cannot be sped up by dealing MAPS to 8 ranks: two or three maps per rank leave the matrix
// instructions with 4-8 columns to contract against.  Sharded by m instead, every rank runs the Legendre stage of ALL components
// -- the full-batch kernels at their best shape -- on 1/N of the orders m:
//     rank r:  ring FFT of its own maps  ->  hx_ring_modes: (F_N, F_S)(m, ring pair) of its components, cut by destination range
//     all-to-all over xGMI (32 B per component, m and ring pair: 1.6 GB per component in all)
//     rank q:  hx_legendre_from_modes on its range [m0, m1) for every component  ->  alm(l, m in range)
//     local all-pairs Cl over its m (alm2cl is a sum over m), all-reduce of the small Cl blocks.
// The mode blocks carry the ring phase and the quadrature weights, so the receiver needs no map-side data.
#include <algorithm>
#include <vector>

#include "hx_sht_common.h"

namespace hx {
using namespace hxfft;

// out[(c (m1 - m0) + (m - m0)) nrp_pad + rp] = (F_N.re, F_N.im, F_S.re, F_S.im) of component c0 + c
// grid: x = m - m0, y = blocks of 256 ring pairs; one thread per ring pair, looping over the components of the batch
__global__ __launch_bounds__(256) void k_ring_modes(PlanDev P, const double2 *__restrict__ Y, int nb, int c0, int ncomp, int m0, int m1,
                                                    const double *__restrict__ rw, const LegTask *__restrict__ tasks,
                                                    const MTasks *__restrict__ of_m, const LegTask *__restrict__ tasks2,
                                                    const MTasks *__restrict__ of_m2, double4 *__restrict__ out)
{
    const int m = m0 + blockIdx.x;
    const MTasks mt = of_m[m], mt2 = of_m2[m];
    // ring pairs in front of the first task of this m -- of the spin-0 AND of the spin-2 list: their pruning limits differ by a
    // ring or two -- are pruned by every Legendre kernel: never read by the receiver
    int rb0 = 1 << 30;
    if (mt.count) rb0 = tasks[mt.first].rb0;
    if (mt2.count) rb0 = min(rb0, tasks2[mt2.first].rb0);
    if (((int)blockIdx.y + 1) * 256 <= (long long)rb0 * RBLK) return;
    const int rp = blockIdx.y * 256 + threadIdx.x;
    if (rp >= P.nrp_pad) return;
    const RingAtM ram = ring_at_m_of(P, rp, m, rw);
    for (int c = 0; c < nb; ++c) {
        double2 fn = make_double2(0.0, 0.0), fs = fn;
        if (rp < P.nrp) ring_modes_ns(P, Y, c, rp, m, ram, fn, fs);
        out[((long long)(c0 + c) * (m1 - m0) + (m - m0)) * P.nrp_pad + rp] = make_double4(fn.x, fn.y, fs.x, fs.y);
    }
    (void)ncomp;
}

}  // namespace hx

using namespace hx;

// Relative cost of order m in the Legendre stage (ring blocks that are not pruned x 32-l blocks), for cutting [0, lmax] into
// ranges of equal work.
extern "C" int hx_plan_m_cost(hx_plan *pl, int spin, double *cost)
{
    if (!pl || !cost || (spin != 0 && spin != 2)) return fail(HX_ERR_ARG, "hx_plan_m_cost: bad arguments");
    HX_TRY(ensure_ready());
    HX_TRY(build_tasks(pl, spin));
    const hx_plan::TaskSet &ts = pl->ts[spin ? 1 : 0];
    const int l0min = spin ? 2 : 0;
    for (int m = 0; m <= pl->lmax; ++m) {
        double c = 0.0;
        const int l0 = std::max(m, l0min);
        for (int t = ts.of_m[m].first; t < ts.of_m[m].first + ts.of_m[m].count; ++t)
            c += (double)ts.tasks[t].nrb * ((pl->lmax - l0) / LBLK + 1);
        cost[m] = c;
    }
    return HX_OK;
}

extern "C" int64_t hx_ring_modes_size(const hx_plan *pl, int m0, int m1)
{
    if (!pl || m0 < 0 || m1 < m0 || m1 > pl->lmax + 1) return -1;
    return (int64_t)(m1 - m0) * pl->nrp_pad * 4;
}

extern "C" int hx_ring_modes(hx_plan *pl, int ncomp, const double *maps, const double *pix_weights, const double *ring_weights, int nranges,
                             const int *mbounds, double *const *outs)
{
    HX_TRY(ensure_ready());
    if (!pl || pl->nside < 1 || ncomp < 1 || !maps || nranges < 1 || !mbounds || !outs) return fail(HX_ERR_ARG, "hx_ring_modes: bad arguments");
    for (int q = 0; q < nranges; ++q) {
        if (mbounds[q] < 0 || mbounds[q + 1] < mbounds[q] || mbounds[q + 1] > pl->lmax + 1) return fail(HX_ERR_ARG, "hx_ring_modes: bad m ranges");
        if (mbounds[q + 1] > mbounds[q] && (!outs[q] || !is_device_ptr(outs[q]))) return fail(HX_ERR_ARG, "hx_ring_modes: outputs must be device buffers");
    }
    HX_TRY(build_tasks(pl, 0));
    HX_TRY(build_tasks(pl, 2));
    const hx_plan::TaskSet &ts = pl->ts[0], &ts2 = pl->ts[1];
    InView vmaps, vrw, vpw;
    HX_TRY(vmaps.bind(maps, sizeof(double) * (size_t)ncomp * pl->npix));
    HX_TRY(vrw.bind(ring_weights, sizeof(double) * pl->nrp));
    HX_TRY(vpw.bind(pix_weights, sizeof(double) * (size_t)pl->npix));
    PlanDev P = pl->dev();
    P.nssrc = nullptr;
    P.hsrc = nullptr;
    const int batch = 16;  // components per ring-FFT launch (Y: 1.6 GB per component at nside 4096)
    for (int c0 = 0; c0 < ncomp; c0 += batch) {
        const int nb = std::min(batch, ncomp - c0);
        HX_TRY(pl->Y.alloc(sizeof(double2) * (size_t)pl->ny * nb));
        HX_TRY(launch_ring_subdft_maps(pl, nb, vmaps.as<double>() + (size_t)c0 * pl->npix, vpw.as<double>(), pl->Y.as<double2>()));
        ProfScope ps("ring_modes");
        for (int q = 0; q < nranges; ++q) {
            const int m0 = mbounds[q], m1 = mbounds[q + 1];
            if (m1 <= m0) continue;
            dim3 grid(m1 - m0, (pl->nrp_pad + 255) / 256);
            hipLaunchKernelGGL(k_ring_modes, grid, dim3(256), 0, rt().stream, P, pl->Y.as<double2>(), nb, c0, ncomp, m0, m1, vrw.as<double>(),
                               ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(), ts2.d_tasks.as<LegTask>(), ts2.d_of_m.as<MTasks>(),
                               reinterpret_cast<double4 *>(outs[q]));
        }
        HX_HIP(hipGetLastError());
    }
    HX_HIP(hipStreamSynchronize(rt().stream));  // the blocks are handed to a collective next
    return HX_OK;
}

extern "C" int hx_legendre_from_modes(hx_plan *pl, int spin, int ncomp, const double *const *comp_modes, int m0, int m1, double *alms,
                                      const double *fl)
{
    HX_TRY(ensure_ready());
    if (!pl || pl->nside < 1 || !comp_modes || !alms) return fail(HX_ERR_ARG, "hx_legendre_from_modes: bad arguments");
    if (spin != 0 && spin != 2) return fail(HX_ERR_UNSUPPORTED, "spin-%d maps not yet supported", spin);
    if (ncomp < 1 || (spin == 2 && (ncomp & 1))) return fail(HX_ERR_ARG, "bad component count %d for spin %d", ncomp, spin);
    if (m0 < 0 || m1 <= m0 || m1 > pl->lmax + 1) return fail(HX_ERR_ARG, "hx_legendre_from_modes: bad m range [%d, %d)", m0, m1);
    if (!is_device_ptr(alms)) return fail(HX_ERR_ARG, "hx_legendre_from_modes: alms must be a device buffer (only m in the range is written)");
    for (int c = 0; c < ncomp; ++c)
        if (!comp_modes[c] || !is_device_ptr(comp_modes[c])) return fail(HX_ERR_ARG, "hx_legendre_from_modes: mode blocks must be device buffers");
    InView vfl;
    HX_TRY(vfl.bind(fl, sizeof(double) * (pl->lmax + 1)));
    DevBuf d_tab;
    HX_TRY(d_tab.alloc(sizeof(void *) * ncomp));
    HX_HIP(hipMemcpy(d_tab.p, comp_modes, sizeof(void *) * ncomp, hipMemcpyHostToDevice));
    const double4 *const *tab = d_tab.as<const double4 *>();
    int rc = HX_OK;
    pl->ns_m0 = m0;
    pl->m_lo = m0;
    pl->m_hi = m1;
    for (int c0 = 0, nb = 0; c0 < ncomp && rc == HX_OK; c0 += nb) {
        nb = analysis_next_batch(spin, ncomp - c0);
        pl->nssrc = tab + c0;
        rc = analysis_batch(pl, spin, nb, nullptr, reinterpret_cast<double2 *>(alms) + (size_t)c0 * pl->nlm, nullptr, nullptr, vfl.as<double>(), 0);
    }
    pl->nssrc = nullptr;
    pl->m_lo = 0;
    pl->m_hi = -1;
    if (hipStreamSynchronize(rt().stream) != hipSuccess && rc == HX_OK) rc = fail(HX_ERR_HIP, "hx_legendre_from_modes: stream error");  // d_tab dies here
    return rc;
}
