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

// A set of orders is (first, count, step): m = first + k step, k < count.  Rank q of N owns (q, ., N): every rank gets work-groups
// of every length (the cost of an order falls steadily with m) and as many of them as one GPU's launch / N -- contiguous ranges of
// equal cost gave the rank of the lowest orders 317 work-groups for 256 compute units (108 ms against 78 on the others).
//
// out[(c count + k) nrp_pad + rp] = (F_N.re, F_N.im, F_S.re, F_S.im) of component c0 + c at order first + k step
// grid: x = k, y = blocks of 256 ring pairs; one thread per ring pair, looping over the components of the batch
__global__ __launch_bounds__(256) void k_ring_modes(PlanDev P, const double2 *__restrict__ Y, int nb, int c0, int first, int count, int step,
                                                    const double *__restrict__ rw, const LegTask *__restrict__ tasks,
                                                    const MTasks *__restrict__ of_m, const LegTask *__restrict__ tasks2,
                                                    const MTasks *__restrict__ of_m2, double4 *__restrict__ out)
{
    const int m = first + blockIdx.x * step;
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
        out[((long long)(c0 + c) * count + blockIdx.x) * P.nrp_pad + rp] = make_double4(fn.x, fn.y, fs.x, fs.y);
    }
}

}  // namespace hx

using namespace hx;

static int check_orders(const hx_plan *pl, int first, int count, int step, const char *who)
{
    if (first < 0 || count < 0 || step < 1 || (count > 0 && first + (long long)(count - 1) * step > pl->lmax))
        return fail(HX_ERR_ARG, "%s: bad set of orders (first %d, count %d, step %d, lmax %d)", who, first, count, step, pl->lmax);
    return HX_OK;
}

extern "C" int64_t hx_ring_modes_size(const hx_plan *pl, int count)
{
    if (!pl || count < 0) return -1;
    return (int64_t)count * pl->nrp_pad * 4;
}

extern "C" int hx_ring_modes(hx_plan *pl, int ncomp, const double *maps, const double *pix_weights, const double *ring_weights, int nsets,
                             const int *m_first, const int *m_count, int m_step, double *const *outs)
{
    HX_TRY(ensure_ready());
    if (!pl || pl->nside < 1 || ncomp < 1 || !maps || nsets < 1 || !m_first || !m_count || !outs) return fail(HX_ERR_ARG, "hx_ring_modes: bad arguments");
    for (int q = 0; q < nsets; ++q) {
        HX_TRY(check_orders(pl, m_first[q], m_count[q], m_step, "hx_ring_modes"));
        if (m_count[q] > 0 && (!outs[q] || !is_device_ptr(outs[q]))) return fail(HX_ERR_ARG, "hx_ring_modes: outputs must be device buffers");
    }
    HX_TRY(build_tasks(pl, 0));
    HX_TRY(build_tasks(pl, 2));
    const hx_plan::TaskSet &ts = pl->ts[0], &ts2 = pl->ts[1];
    InView vmaps, vrw, vpw;
    HX_TRY(vmaps.bind(maps, sizeof(double) * (size_t)ncomp * pl->npix));
    HX_TRY(vrw.bind(ring_weights, sizeof(double) * pl->nrp));
    HX_TRY(vpw.bind(pix_weights, sizeof(double) * (size_t)pl->npix));
    HX_TRY(classify_pixel_weights(pl, vpw.as<double>()));
    PlanDev P = pl->dev();
    P.nssrc = nullptr;
    P.hsrc = nullptr;
    const int batch = 16;  // components per ring-FFT launch (Y: 1.6 GB per component at nside 4096)
    for (int c0 = 0; c0 < ncomp; c0 += batch) {
        const int nb = std::min(batch, ncomp - c0);
        HX_TRY(pl->Y.alloc(sizeof(double2) * (size_t)pl->ny * nb));
        HX_TRY(launch_ring_subdft_maps(pl, nb, vmaps.as<double>() + (size_t)c0 * pl->npix, vpw.as<double>(), pl->Y.as<double2>()));
        ProfScope ps("ring_modes");
        for (int q = 0; q < nsets; ++q) {
            if (m_count[q] <= 0) continue;
            dim3 grid(m_count[q], (pl->nrp_pad + 255) / 256);
            hipLaunchKernelGGL(k_ring_modes, grid, dim3(256), 0, rt().stream, P, pl->Y.as<double2>(), nb, c0, m_first[q], m_count[q], m_step,
                               vrw.as<double>(), ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(), ts2.d_tasks.as<LegTask>(), ts2.d_of_m.as<MTasks>(),
                               reinterpret_cast<double4 *>(outs[q]));
        }
        HX_HIP(hipGetLastError());
    }
    HX_HIP(hipStreamSynchronize(rt().stream));  // the blocks are handed to a collective next
    return HX_OK;
}

extern "C" int hx_legendre_from_modes(hx_plan *pl, int spin, int ncomp, const double *const *comp_modes, int m_first, int m_count, int m_step,
                                      double *alms, const double *fl)
{
    HX_TRY(ensure_ready());
    if (!pl || pl->nside < 1 || !comp_modes || !alms) return fail(HX_ERR_ARG, "hx_legendre_from_modes: bad arguments");
    if (spin != 0 && spin != 2) return fail(HX_ERR_UNSUPPORTED, "spin-%d maps not yet supported", spin);
    if (ncomp < 1 || (spin == 2 && (ncomp & 1))) return fail(HX_ERR_ARG, "bad component count %d for spin %d", ncomp, spin);
    HX_TRY(check_orders(pl, m_first, m_count, m_step, "hx_legendre_from_modes"));
    if (m_count == 0) return HX_OK;
    if (!is_device_ptr(alms)) return fail(HX_ERR_ARG, "hx_legendre_from_modes: alms must be a device buffer (only the given orders are written)");
    for (int c = 0; c < ncomp; ++c)
        if (!comp_modes[c] || !is_device_ptr(comp_modes[c])) return fail(HX_ERR_ARG, "hx_legendre_from_modes: mode blocks must be device buffers");
    InView vfl;
    HX_TRY(vfl.bind(fl, sizeof(double) * (pl->lmax + 1)));
    DevBuf d_tab;
    HX_TRY(d_tab.alloc(sizeof(void *) * ncomp));
    HX_HIP(hipMemcpy(d_tab.p, comp_modes, sizeof(void *) * ncomp, hipMemcpyHostToDevice));
    const double4 *const *tab = d_tab.as<const double4 *>();
    int rc = HX_OK;
    pl->ns_m0 = m_first;
    pl->m_lo = m_first;
    pl->m_hi = m_first + (m_count - 1) * m_step + 1;
    pl->m_step = m_step;
    for (int c0 = 0, nb = 0; c0 < ncomp && rc == HX_OK; c0 += nb) {
        nb = analysis_next_batch(spin, ncomp - c0);
        pl->nssrc = tab + c0;
        rc = analysis_batch(pl, spin, nb, nullptr, reinterpret_cast<double2 *>(alms) + (size_t)c0 * pl->nlm, nullptr, nullptr, vfl.as<double>(), 0);
    }
    pl->nssrc = nullptr;
    pl->m_lo = 0;
    pl->m_hi = -1;
    pl->m_step = 1;
    if (hipStreamSynchronize(rt().stream) != hipSuccess && rc == HX_OK) rc = fail(HX_ERR_HIP, "hx_legendre_from_modes: stream error");  // d_tab dies here
    return rc;
}

// ---- all-gather of alm shards over RCCL for hosts without torch.distributed --------------------------------------
// The Python layer's collectives live in torch.distributed (heracles_amd/distributed.py).  A host in another language that shards the
// maps of a job over the GPUs of a node (one process per GPU) gathers the alms with this entry point: in place on the buffer
// hx_map2alm wrote, shards of unequal length (a rank's components: spin-2 maps count double), one ncclBroadcast per shard inside a
// group -- over xGMI every shard then travels on all links of its owner at once (a ring all-gather of padded shards is bound by one
// link).  The library does NO bootstrap: `comm` is an ncclComm_t the host created (ncclCommInitRank with an id it exchanged its own
// way).  RCCL is bound at the first call (dlopen of librccl.so.1: the copy already in the process if there is one), so the library
// itself does not depend on it.
#include <dlfcn.h>
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*group_start)() = nullptr;
    int (*group_end)() = nullptr;
    int (*broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    const char *(*error_string)(int) = nullptr;
    bool tried = false;
};
Rccl &rccl()
{
    static Rccl r;
    if (!r.tried) {
        r.tried = true;
        for (const char *name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.lib) break;
        }
        if (r.lib) {
            r.group_start = reinterpret_cast<int (*)()>(dlsym(r.lib, "ncclGroupStart"));
            r.group_end = reinterpret_cast<int (*)()>(dlsym(r.lib, "ncclGroupEnd"));
            r.broadcast = reinterpret_cast<int (*)(const void *, void *, size_t, int, int, void *, hipStream_t)>(dlsym(r.lib, "ncclBroadcast"));
            r.error_string = reinterpret_cast<const char *(*)(int)>(dlsym(r.lib, "ncclGetErrorString"));
        }
    }
    return r;
}
}  // namespace

// counts [nranks]: complex elements of every rank's shard; buf: device, interleaved complex, sum(counts) elements, rank q's shard at
// offset sum_{p < q} counts[p] (this rank's own shard already in place).  Stream-ordered on the library's stream; complete on return
// unless hx_set_async(1).
extern "C" int hx_allgather_alms(void *comm, int nranks, const int64_t *counts, double *buf)
{
    using namespace hx;
    HX_TRY(ensure_ready());
    if (!comm || nranks < 1 || !counts || !buf) return fail(HX_ERR_ARG, "hx_allgather_alms: bad argument");
    if (!is_device_ptr(buf)) return fail(HX_ERR_ARG, "hx_allgather_alms: buf must be device memory (the buffer hx_map2alm wrote)");
    for (int q = 0; q < nranks; ++q)
        if (counts[q] < 0) return fail(HX_ERR_ARG, "hx_allgather_alms: counts[%d] < 0", q);
    Rccl &r = rccl();
    if (!r.lib || !r.group_start || !r.group_end || !r.broadcast)
        return fail(HX_ERR_UNSUPPORTED, "hx_allgather_alms: librccl.so.1 could not be loaded (%s)", r.lib ? "symbols missing" : dlerror());
    auto check = [&](int rc, const char *what) {
        if (rc == 0) return HX_OK;
        return fail(HX_ERR_HIP, "hx_allgather_alms: %s failed: %s", what, r.error_string ? r.error_string(rc) : "RCCL error");
    };
    HX_TRY(check(r.group_start(), "ncclGroupStart"));
    int64_t off = 0;
    int rc = HX_OK;
    for (int q = 0; q < nranks && rc == HX_OK; ++q) {
        if (counts[q] > 0) {
            double *p = buf + 2 * off;
            rc = check(r.broadcast(p, p, (size_t)(2 * counts[q]), /* ncclDouble */ 8, q, comm, rt().stream), "ncclBroadcast");
        }
        off += counts[q];
    }
    const int rc_end = check(r.group_end(), "ncclGroupEnd");
    if (rc != HX_OK) return rc;
    HX_TRY(rc_end);
    return finish_call();
}
