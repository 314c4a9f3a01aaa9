// hx_mapper.hip -- catalogue -> HEALPix map accumulation on the GPU: ang2pix (RING) and the
// ordered scatter-add of HealpixMapper.map_values (heracles/healpy.py:58-66, :144-160).
//
// Reference semantics: ipix = hp.ang2pix(nside, lon, lat, lonlat=True), then the compiled
// loop `for j, i in enumerate(ipix): maps[..., i] += values[..., j]` -- each pixel receives
// its points in catalogue order.  Floating-point addition does not commute with that order,
// so the default path here reproduces it exactly: a stable radix sort of the point indices
// by pixel (rocPRIM), then one thread per occupied pixel adds its run of points front to
// back.  HX_MAP_ATOMIC trades that guarantee for a single pass of hardware f64 atomics.
//
// ang2pix follows the published HEALPix algorithm (Gorski et al. 2005; healpix_cxx
// T_Healpix_Base::loc2pix, RING branch), the third-party code healpy calls; it is absent
// from /root/reference, so index parity is pinned on the test-side numpy restatement
// and on the pix2ang round trip of every pixel centre (tests/test_gpu_mapper.py).
#include "hx_common.h"

#include "hx_sort.h"

namespace hx {

namespace {

constexpr double kInvHalfPi = 0.6366197723675813430755350534900574;
constexpr double kDeg2Rad = 0.017453292519943295769236907684886;  // numpy: NPY_PI / 180
constexpr double kHalfPi = 1.5707963267948966192313216916398;
constexpr double kTwoThird = 2.0 / 3.0;

// healpix_cxx fmodulo(v, 4.0)
__device__ inline double fmodulo4(double v)
{
#pragma clang fp contract(off)
    if (v >= 0.0) return (v < 4.0) ? v : fmod(v, 4.0);
    double t = fmod(v, 4.0) + 4.0;
    return (t == 4.0) ? 0.0 : t;
}

__device__ inline long long ang2pix_ring_one(long long nside, double lon_deg, double lat_deg)
{
    // every operation rounds on its own, as in numpy / healpix_cxx built without FMA contraction: a fused
    // pi/2 - lat * (pi/180) gives theta = -6e-17 instead of 0 for lat = 90
#pragma clang fp contract(off)
    // healpy lonlat2thetaphi: theta = pi/2 - radians(lat), phi = radians(lon)
    double theta = kHalfPi - lat_deg * kDeg2Rad;
    double phi = lon_deg * kDeg2Rad;
    double z = cos(theta);
    bool have_sth = (theta < 0.01) || (theta > 3.14159 - 0.01);
    double sth = have_sth ? sin(theta) : 0.0;
    double za = fabs(z);
    double tt = fmodulo4(phi * kInvHalfPi);
    long long npix = 12 * nside * nside, ncap = 2 * nside * (nside - 1), nl4 = 4 * nside;
    if (za <= kTwoThird) {
        double temp1 = (double)nside * (0.5 + tt);
        double temp2 = (double)nside * z * 0.75;
        long long jp = (long long)(temp1 - temp2);
        long long jm = (long long)(temp1 + temp2);
        long long ir = nside + 1 + jp - jm;
        long long kshift = 1 - (ir & 1);
        long long t1 = jp + jm - nside + kshift + 1 + nl4 + nl4;
        long long ip = (t1 >> 1) % nl4;
        return ncap + (ir - 1) * nl4 + ip;
    }
    double tp = tt - (double)(long long)tt;
    double tmp = ((za < 0.99) || !have_sth) ? (double)nside * sqrt(3.0 * (1.0 - za))
                                           : (double)nside * sth / sqrt((1.0 + za) / 3.0);
    long long jp = (long long)(tp * tmp);
    long long jm = (long long)((1.0 - tp) * tmp);
    long long ir = jp + jm + 1;
    long long ip = (long long)(tt * (double)ir);
    if (ip >= 4 * ir) ip = 4 * ir - 1;  // healpix_cxx asserts this never happens; stay in range
    return (z > 0.0) ? 2 * ir * (ir - 1) + ip : npix - 2 * ir * (ir + 1) + ip;
}

// healpy.ang2pix validates its input (check_theta_valid: 0 <= theta <= pi + 1e-5, which a NaN fails) and
// raises ValueError before any pixel is touched; the formulas above yield negative or out-of-range
// indices for such points.  Invalid points (a non-finite longitude included) get pixel -1 and are counted
// in *nbad; the entry points return HX_ERR_ARG before anything is scattered.
__device__ inline bool lonlat_valid(double lon_deg, double lat_deg)
{
#pragma clang fp contract(off)
    const double theta = kHalfPi - lat_deg * kDeg2Rad;
    return isfinite(lon_deg) && theta >= 0.0 && theta <= 3.14159265358979323846 + 1e-5;
}

__global__ __launch_bounds__(256) void k_ang2pix(long long nside, long long n, const double *__restrict__ lon,
                                                 const double *__restrict__ lat, long long *__restrict__ ipix,
                                                 unsigned *__restrict__ order, unsigned long long *__restrict__ nbad)
{
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const double lo = lon[j], la = lat[j];
    const bool ok = lonlat_valid(lo, la);
    const long long npix = 12 * nside * nside;
    long long p = ok ? ang2pix_ring_one(nside, lo, la) : -1;
    if (ok && (p < 0 || p >= npix)) p = -1;  // never scatter outside the map
    if (p < 0) atomicAdd(nbad, 1ULL);
    ipix[j] = p;
    if (order) order[j] = (unsigned)j;
}

// one thread per sorted position; the first position of each run of equal pixels owns it
template <class PIX>
__global__ __launch_bounds__(256) void k_run_add(long long n, const PIX *__restrict__ pix_sorted,
                                                 const unsigned *__restrict__ idx_sorted, int nval,
                                                 const double *__restrict__ values, long long vstride,
                                                 double *__restrict__ maps, long long npix)
{
    long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    const PIX p = pix_sorted[s];
    if (s > 0 && pix_sorted[s - 1] == p) return;
    long long e = s + 1;
    while (e < n && pix_sorted[e] == p) ++e;
    for (int v = 0; v < nval; ++v) {
        double acc = maps[(long long)v * npix + p];
        const double *val = values + (long long)v * vstride;
        for (long long t = s; t < e; ++t) acc += val[idx_sorted[t]];
        maps[(long long)v * npix + p] = acc;
    }
}

__global__ __launch_bounds__(256) void k_scatter_atomic(long long n, const long long *__restrict__ ipix, int nval,
                                                        const double *__restrict__ values, long long vstride,
                                                        double *__restrict__ maps, long long npix)
{
    long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    long long p = ipix[j];
    for (int v = 0; v < nval; ++v)
        unsafeAtomicAdd(&maps[(long long)v * npix + p], values[(long long)v * vstride + j]);
}

// ---- RING <-> NEST (healpix_cxx ring2xyf / xyf2nest / nest2xyf / xyf2ring), nside = 2^order ----
__constant__ int c_jrll[12] = {2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4};
__constant__ int c_jpll[12] = {1, 3, 5, 7, 0, 2, 4, 6, 1, 3, 5, 7};

__device__ inline long long spread_bits(long long v)  // bit b -> bit 2b (v < 2^16)
{
    v = (v | (v << 8)) & 0x00ff00ffll;
    v = (v | (v << 4)) & 0x0f0f0f0fll;
    v = (v | (v << 2)) & 0x33333333ll;
    v = (v | (v << 1)) & 0x55555555ll;
    return v;
}
__device__ inline long long compress_bits(long long v)  // bit 2b -> bit b
{
    v &= 0x55555555ll;
    v = (v | (v >> 1)) & 0x33333333ll;
    v = (v | (v >> 2)) & 0x0f0f0f0fll;
    v = (v | (v >> 4)) & 0x00ff00ffll;
    v = (v | (v >> 8)) & 0x0000ffffll;
    return v;
}
__device__ inline long long isqrt_ll(long long v) { return (long long)sqrt((double)v + 0.5); }

__device__ inline long long nest2ring_dev(int order, long long pnest)
{
    const long long nside = 1ll << order, npface = nside * nside, npix = 12 * npface;
    const long long ncap = 2 * nside * (nside - 1), nl4 = 4 * nside;
    const int face = (int)(pnest >> (2 * order));
    const long long pf = pnest & (npface - 1);
    const long long ix = compress_bits(pf), iy = compress_bits(pf >> 1);
    const long long jr = ((long long)c_jrll[face] << order) - ix - iy - 1;
    long long nr, n_before, kshift;
    if (jr < nside) { nr = jr; n_before = 2 * nr * (nr - 1); kshift = 0; }
    else if (jr > 3 * nside) { nr = nl4 - jr; n_before = npix - 2 * (nr + 1) * nr; kshift = 0; }
    else { nr = nside; n_before = ncap + (jr - nside) * nl4; kshift = (jr - nside) & 1; }
    long long jp = (c_jpll[face] * nr + ix - iy + 1 + kshift) / 2;
    if (jp > nl4) jp -= nl4;
    else if (jp < 1) jp += nl4;
    return n_before + jp - 1;
}

__device__ inline long long ring2nest_dev(int order, long long pring)
{
    const long long nside = 1ll << order, npface = nside * nside, npix = 12 * npface;
    const long long ncap = 2 * nside * (nside - 1), nl4 = 4 * nside;
    long long iring, iphi, kshift, nr;
    int face;
    if (pring < ncap) {
        iring = (1 + isqrt_ll(1 + 2 * pring)) >> 1;
        iphi = (pring + 1) - 2 * iring * (iring - 1);
        kshift = 0; nr = iring;
        face = (int)((iphi - 1) / nr);
    } else if (pring < npix - ncap) {
        const long long ip = pring - ncap, tmp = ip >> (order + 2);
        iring = tmp + nside;
        iphi = ip - tmp * nl4 + 1;
        kshift = (iring + nside) & 1; nr = nside;
        const long long ire = tmp + 1, irm = 2 * nside + 1 - tmp;
        const long long ifm = (iphi - ire / 2 + nside - 1) >> order, ifp = (iphi - irm / 2 + nside - 1) >> order;
        face = (int)((ifp == ifm) ? (ifp | 4) : ((ifp < ifm) ? ifp : (ifm + 8)));
    } else {
        const long long ip = npix - pring;
        const long long ir = (1 + isqrt_ll(2 * ip - 1)) >> 1;
        iphi = 4 * ir + 1 - (ip - 2 * ir * (ir - 1));
        kshift = 0; nr = ir;
        iring = nl4 - ir;
        face = (int)(8 + (iphi - 1) / nr);
    }
    const long long irt = iring - c_jrll[face] * nside + 1;
    long long ipt = 2 * iphi - c_jpll[face] * nr - kshift - 1;
    if (ipt >= 2 * nside) ipt -= 8 * nside;
    const long long ix = (ipt - irt) >> 1, iy = (-ipt - irt) >> 1;
    return face * npface + spread_bits(ix) + (spread_bits(iy) << 1);
}

// numpy's pairwise summation (the arithmetic behind healpy's np.sum(axis=1) in _ud_grade_core):
// n < 8 sequential from 0; n <= 128 eight strided accumulators; else split at n/2 rounded to 8.
template <class Load>
__device__ double numpy_pairwise_sum(long long i0, long long n, Load load)
{
    if (n < 8) {
        double res = 0.0;
        for (long long i = 0; i < n; ++i) res += load(i0 + i);
        return res;
    }
    if (n <= 128) {
        double r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = load(i0 + k);
        long long i = 8;
        for (; i < n - (n % 8); i += 8)
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] += load(i0 + i + k);
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += load(i0 + i);
        return res;
    }
    long long n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_pairwise_sum(i0, n2, load) + numpy_pairwise_sum(i0 + n2, n - n2, load);
}

constexpr double kUnseen = -1.6375e30;

// healpy.ud_grade, RING -> RING, pess=False, power=None.  One thread per output pixel.
__global__ __launch_bounds__(256) void k_ud_grade(int order_in, int order_out, const double *__restrict__ in,
                                                  double *__restrict__ out)
{
    const long long npix_out = 12ll << (2 * order_out), npix_in = 12ll << (2 * order_in);
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix_out) return;
    const double *src = in + (long long)blockIdx.y * npix_in;
    double *dst = out + (long long)blockIdx.y * npix_out;
    const long long q = ring2nest_dev(order_out, p);
    if (order_out >= order_in) {
        dst[p] = src[nest2ring_dev(order_in, q >> (2 * (order_out - order_in)))];
        return;
    }
    const long long rat2 = 1ll << (2 * (order_in - order_out));
    long long nhit = 0;
    // mask_bad(rtol=1e-5, atol=1e-15) | ~isfinite; a masked value enters the sum as value * 0
    auto load = [&](long long c) {
        const double v = src[nest2ring_dev(order_in, q * rat2 + c)];
        const bool good = !(fabs(v - kUnseen) <= 1e-15 + 1e-5 * fabs(kUnseen)) && isfinite(v);
        nhit += good;
        return good ? v : v * 0.0;
    };
    const double s = numpy_pairwise_sum(0, rat2, load);
    dst[p] = nhit ? s / (double)nhit : kUnseen;
}

bool nside_ok(int nside) { return nside >= 1 && nside <= (1 << 24); }
bool nside_pow2(int nside) { return nside >= 1 && nside <= 8192 && (nside & (nside - 1)) == 0; }

// Pixel indices of n points; fails with HX_ERR_ARG (the reference: ValueError from healpy) if any point is
// outside 0 <= theta <= pi or not finite.  Synchronises the stream (the count is read back).
int launch_ang2pix(int nside, long long n, const double *lon, const double *lat, long long *ipix, unsigned *order,
                   const char *who)
{
    if (n == 0) return HX_OK;
    static DevBuf d_bad;  // 8 bytes, lives with the process
    HX_TRY(d_bad.alloc(sizeof(unsigned long long)));
    hipStream_t st = rt().stream;
    HX_HIP(hipMemsetAsync(d_bad.p, 0, sizeof(unsigned long long), st));
    unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_ang2pix, dim3(blocks), dim3(256), 0, st, (long long)nside, n, lon, lat, ipix, order,
                       d_bad.as<unsigned long long>());
    HX_HIP(hipGetLastError());
    unsigned long long nbad = 0;
    HX_HIP(hipMemcpyAsync(&nbad, d_bad.p, sizeof(nbad), hipMemcpyDeviceToHost, st));
    HX_HIP(hipStreamSynchronize(st));
    if (nbad)
        return fail(HX_ERR_ARG, "%s: %llu of %lld points have a latitude outside [-90, 90] or a non-finite coordinate "
                    "(healpy: THETA is out of range [0,pi])", who, nbad, n);
    return HX_OK;
}

}  // namespace

}  // namespace hx

using namespace hx;

extern "C" int hx_ang2pix_ring(int nside, int64_t n, const double *lon, const double *lat, int64_t *ipix)
{
    HX_TRY(ensure_ready());
    if (!nside_ok(nside) || n < 0 || (n > 0 && (!lon || !lat || !ipix)))
        return fail(HX_ERR_ARG, "hx_ang2pix_ring: bad arguments (nside=%d n=%lld)", nside, (long long)n);
    if (n == 0) return HX_OK;
    InView vlon, vlat;
    OutView vout;
    HX_TRY(vlon.bind(lon, sizeof(double) * n));
    HX_TRY(vlat.bind(lat, sizeof(double) * n));
    HX_TRY(vout.bind(ipix, sizeof(int64_t) * n));
    {
        ProfScope ps("ang2pix");
        HX_TRY(launch_ang2pix(nside, n, vlon.as<double>(), vlat.as<double>(), (long long *)vout.as<int64_t>(), nullptr, "hx_ang2pix_ring"));
    }
    HX_TRY(vout.finish());
    return finish_call();
}

extern "C" int hx_map_values(int nside, int64_t n, const double *lon, const double *lat, int nval,
                             const double *values, double *maps, int flags)
{
    HX_TRY(ensure_ready());
    if (!nside_ok(nside) || n < 0 || nval < 0 || (n > 0 && nval > 0 && (!lon || !lat || !values || !maps)))
        return fail(HX_ERR_ARG, "hx_map_values: bad arguments (nside=%d n=%lld nval=%d)", nside, (long long)n, nval);
    if (n > 0xfffffff0ll)
        return fail(HX_ERR_UNSUPPORTED, "hx_map_values: at most 2^32-16 points per call (got %lld); page the catalogue",
                    (long long)n);
    if (n == 0 || nval == 0) return HX_OK;
    const long long npix = 12ll * nside * nside;
    hipStream_t st = rt().stream;
    InView vlon, vlat, vval;
    HX_TRY(vlon.bind(lon, sizeof(double) * n));
    HX_TRY(vlat.bind(lat, sizeof(double) * n));
    HX_TRY(vval.bind(values, sizeof(double) * n * nval));
    // maps are read-modify-write: a host array makes the round trip through a device copy
    DevBuf mtmp;
    double *dmaps = maps;
    const bool host_maps = !is_device_ptr(maps);
    if (host_maps) {
        HX_TRY(mtmp.alloc(sizeof(double) * npix * nval));
        HX_TRY(copy_h2d(mtmp.p, maps, sizeof(double) * npix * nval));
        dmaps = mtmp.as<double>();
    }
    DevBuf bpix, bord, bpix2, bord2, btmp;
    long long *spix64 = nullptr;
    unsigned *spix = nullptr;    // the sorted (pixel, point index) pairs: whichever buffers the last pass wrote (keys narrowed to 32 bits)
    unsigned *sord = nullptr;
    HX_TRY(bpix.alloc(sizeof(long long) * n));
    const bool ordered = !(flags & HX_MAP_ATOMIC);
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (ordered) HX_TRY(bord.alloc(sizeof(unsigned) * n));
    {
        ProfScope ps("ang2pix");
        HX_TRY(launch_ang2pix(nside, n, vlon.as<double>(), vlat.as<double>(), bpix.as<long long>(),
                              ordered ? bord.as<unsigned>() : nullptr, "hx_map_values"));
    }
    if (ordered) {
        HX_TRY(bpix2.alloc(sizeof(long long) * n));
        HX_TRY(bord2.alloc(sizeof(unsigned) * n));
        unsigned end_bit = 1;
        while ((1ll << end_bit) < npix) ++end_bit;
        ProfScope ps("map_sort");
        // HX_SORT_WIDE=1 (tests): the 64-bit path that nside > 16384 takes, at any size
        static const bool force_wide = getenv("HX_SORT_WIDE") && getenv("HX_SORT_WIDE")[0] == '1';
        if (end_bit <= 32 && !force_wide) {
            // pixel indices fit 32 bits (nside <= 16384): the first pass narrows the keys, bpix2 holds the two 32-bit key buffers of the later passes
            HX_TRY(rsort::radix_sort_pairs_narrow(bpix.as<long long>(), bord.as<unsigned>(), bpix2.as<unsigned>(), bpix2.as<unsigned>() + n, bord2.as<unsigned>(),
                                                  (unsigned long long)n, (int)end_bit, btmp, st, &spix, &sord));
        } else {
            HX_TRY(rsort::radix_sort_pairs<long long>(bpix.as<long long>(), bord.as<unsigned>(), bpix2.as<long long>(), bord2.as<unsigned>(),
                                                      (unsigned long long)n, (int)end_bit, btmp, st, &spix64, &sord));
        }
    }
    {
        ProfScope ps("map_add");
        if (ordered && spix64)
            hipLaunchKernelGGL(k_run_add<long long>, dim3(blocks), dim3(256), 0, st, (long long)n, spix64, sord, nval, vval.as<double>(), (long long)n, dmaps, npix);
        else if (ordered)
            hipLaunchKernelGGL(k_run_add<unsigned>, dim3(blocks), dim3(256), 0, st, (long long)n, spix, sord, nval, vval.as<double>(), (long long)n, dmaps, npix);
        else
            hipLaunchKernelGGL(k_scatter_atomic, dim3(blocks), dim3(256), 0, st, (long long)n, bpix.as<long long>(), nval,
                               vval.as<double>(), (long long)n, dmaps, npix);
        HX_HIP(hipGetLastError());
    }
    if (host_maps) {
        return copy_d2h(maps, dmaps, sizeof(double) * npix * nval);
    }
    // temporaries die with this scope: the stream must have drained them
    HX_HIP(hipStreamSynchronize(st));
    return HX_OK;
}

extern "C" int hx_ud_grade(int nside_in, int nside_out, int nmaps, const double *in, double *out)
{
    HX_TRY(ensure_ready());
    if (!nside_pow2(nside_in) || !nside_pow2(nside_out))
        return fail(HX_ERR_ARG, "hx_ud_grade: %d -> %d: nside must be a power of 2 (<= 8192)", nside_in, nside_out);
    if (nmaps < 0 || (nmaps > 0 && (!in || !out))) return fail(HX_ERR_ARG, "hx_ud_grade: bad arguments");
    if (nmaps == 0) return HX_OK;
    const long long npix_in = 12ll * nside_in * nside_in, npix_out = 12ll * nside_out * nside_out;
    InView vin;
    OutView vout;
    HX_TRY(vin.bind(in, sizeof(double) * npix_in * nmaps));
    HX_TRY(vout.bind(out, sizeof(double) * npix_out * nmaps));
    int oi = 0, oo = 0;
    while ((1 << oi) < nside_in) ++oi;
    while ((1 << oo) < nside_out) ++oo;
    {
        ProfScope ps("ud_grade");
        hipLaunchKernelGGL(k_ud_grade, dim3((unsigned)((npix_out + 255) / 256), nmaps), dim3(256), 0, rt().stream, oi, oo,
                           vin.as<double>(), vout.as<double>());
        HX_HIP(hipGetLastError());
    }
    HX_TRY(vout.finish());
    return finish_call();
}

// NESTED <-> RING reordering of full-sky maps (healpy.read_map converts a NESTED file to RING: heracles/io.py:365 reads visibility maps
// through it).  One thread per output pixel: out[p] = in[other index of p].
__global__ __launch_bounds__(256) void k_reorder(int order, int to_ring, const double *__restrict__ in, double *__restrict__ out)
{
    const long long npix = 12ll << (2 * order);
    const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= npix) return;
    const long long q = to_ring ? ring2nest_dev(order, p) : nest2ring_dev(order, p);
    out[(long long)blockIdx.y * npix + p] = in[(long long)blockIdx.y * npix + q];
}

extern "C" int hx_reorder(int nside, int to_ring, int nmaps, const double *in, double *out)
{
    HX_TRY(ensure_ready());
    if (!nside_pow2(nside)) return fail(HX_ERR_ARG, "hx_reorder: nside %d must be a power of 2 (<= 8192)", nside);
    if (nmaps < 0 || (nmaps > 0 && (!in || !out)) || in == out) return fail(HX_ERR_ARG, "hx_reorder: bad arguments (in place is not supported)");
    if (nmaps == 0) return HX_OK;
    const long long npix = 12ll * nside * nside;
    InView vin;
    OutView vout;
    HX_TRY(vin.bind(in, sizeof(double) * npix * nmaps));
    HX_TRY(vout.bind(out, sizeof(double) * npix * nmaps));
    int o = 0;
    while ((1 << o) < nside) ++o;
    hipLaunchKernelGGL(k_reorder, dim3((unsigned)((npix + 255) / 256), nmaps), dim3(256), 0, rt().stream, o, to_ring ? 1 : 0, vin.as<double>(), vout.as<double>());
    HX_HIP(hipGetLastError());
    HX_TRY(vout.finish());
    return finish_call();
}

// =====================================================================================
// healpy pixel weights: compressed octant -> full-sky array (heracles/healpy.py:183-189, use_pixel_weights=True)
// =====================================================================================
// healpy's `healpix_full_weights_nside_NNNN.fits` holds (nside + 1)(3 nside + 1) / 4 values: for every ring of the northern
// hemisphere (equator included) the weights of the pixels of half a quadrant -- the quadrature weights have the 8-fold symmetry
// of the pixelisation (4 quadrants x mirror inside a quadrant) plus north/south.  The expansion is the published algorithm of
// healpix_cxx's apply_fullweights (third-party, not in /root/reference; restated):  ring i < 2 nside has q = min(nside, i + 1)
// pixels per quadrant; it is "shifted" (pixel centres off the quadrant boundary) for i < nside - 1 or i + nside odd; it stores
// w = (q + 1) / 2 values, + 1 if q is even and the ring is not shifted; pixel j of the ring uses value min(j4, q - shifted - j4),
// j4 = j mod q; the map value is multiplied by 1 + value.  One work-group per ring pair.
namespace hx {
__global__ __launch_bounds__(256) void k_expand_pixel_weights(int nside, const long long *__restrict__ voff, const double *__restrict__ wgt,
                                                              double *__restrict__ out)
{
    const int i = blockIdx.x;  // ring of the northern hemisphere, 0 = the ring next to the pole
    const long long ns = nside, npix = 12 * ns * ns;
    const int q = i + 1 < nside ? i + 1 : nside;
    const bool shifted = (i < nside - 1) || ((i + nside) & 1);
    const long long pix = i < nside ? 2ll * i * (i + 1) : 2 * ns * (ns - 1) + (long long)(i - nside + 1) * 4 * ns;  // first pixel of the ring
    const long long psouth = npix - pix - 4ll * q;
    const double *w = wgt + voff[i];
    for (int j = threadIdx.x; j < 4 * q; j += blockDim.x) {
        const int j4 = j % q, mir = q - (shifted ? 1 : 0) - j4;
        const double v = 1.0 + w[j4 < mir ? j4 : mir];
        out[pix + j] = v;
        if (i != 2 * nside - 1) out[psouth + j] = v;
    }
}
}  // namespace hx

extern "C" int64_t hx_pixel_weights_size(int nside) { return nside < 1 ? 0 : ((int64_t)(nside + 1) * (3 * (int64_t)nside + 1)) / 4; }

extern "C" int hx_pixel_weights_expand(int nside, int64_t ncompressed, const double *compressed, double *weights)
{
    HX_TRY(ensure_ready());
    if (nside < 1 || nside > 8192 || !compressed || !weights) return fail(HX_ERR_ARG, "hx_pixel_weights_expand: bad arguments");
    if (ncompressed != hx_pixel_weights_size(nside))
        return fail(HX_ERR_ARG, "hx_pixel_weights_expand: %lld compressed weights, NSIDE=%d needs %lld", (long long)ncompressed, nside,
                    (long long)hx_pixel_weights_size(nside));
    std::vector<long long> voff(2 * nside);
    long long v = 0;
    for (int i = 0; i < 2 * nside; ++i) {
        voff[i] = v;
        const int q = std::min(nside, i + 1);
        const bool shifted = (i < nside - 1) || ((i + nside) & 1);
        v += ((q + 1) >> 1) + (((q & 1) || shifted) ? 0 : 1);
    }
    if (v != ncompressed) return fail(HX_ERR_ARG, "hx_pixel_weights_expand: internal size mismatch");
    DevBuf d_off;
    HX_TRY(d_off.alloc(sizeof(long long) * voff.size()));
    HX_HIP(hipMemcpy(d_off.p, voff.data(), sizeof(long long) * voff.size(), hipMemcpyHostToDevice));
    InView vin;
    OutView vout;
    HX_TRY(vin.bind(compressed, sizeof(double) * ncompressed));
    HX_TRY(vout.bind(weights, sizeof(double) * 12ll * nside * nside));
    hipLaunchKernelGGL(hx::k_expand_pixel_weights, dim3(2 * nside), dim3(256), 0, rt().stream, nside, d_off.as<long long>(), vin.as<double>(),
                       vout.as<double>());
    HX_HIP(hipGetLastError());
    HX_TRY(vout.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));  // d_off dies with this scope
    return HX_OK;
}
