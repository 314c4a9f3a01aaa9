// hx_fits.hip -- payload conversion of the FITS binary tables Heracles keeps its maps and alms in
// (heracles/io.py:128-218: maps as HEALPix RING IMPLICIT FULLSKY tables, one column per map component; alms as the two
// columns "real" / "imag").  A FITS table is ROW-major and BIG-endian: row r holds its value of every column.  The arrays
// of the path are component-major and native-endian.  The byte swap and the (de)interleave are one HBM-bound pass over the
// payload on the GPU, straight from / into the device arrays the transforms work on, instead of numpy's
// astype / moveaxis / ascontiguousarray copies on one host core.
//
//   table element (row, c1, c2), c = c1 * nc2 + c2 the position within the row  <->  array[c1 * s1 + c2 * s2 + row * srow]
//     maps (ncols columns 'D'):            nc1 = ncols, nc2 = 1,  s1 = nrows,  srow = 1
//     alms (columns real, imag 'rD'):      nc1 = 2 (re, im), nc2 = r,  s1 = 1,  s2 = 2 * nrows,  srow = 2   (complex128, shape (r, nrows))
#include "hx_common.h"

namespace hx {
namespace {

__device__ inline unsigned long long bswap64(unsigned long long v) { return __builtin_bswap64(v); }

struct FitsLayout {
    long long nrows;
    int nc1, nc2;
    long long s1, s2, srow;
};

// lanes run along the table row (coalesced on the table side); the array side is strided by construction of the format
template <bool UNPACK>
__global__ __launch_bounds__(256) void k_fits_f64(FitsLayout L, const unsigned long long *__restrict__ src,
                                                  unsigned long long *__restrict__ dst)
{
    const long long nc = (long long)L.nc1 * L.nc2, total = L.nrows * nc;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long st = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += st) {
        const long long row = i / nc;
        const int c = (int)(i - row * nc), c1 = c / L.nc2, c2 = c - c1 * L.nc2;
        const long long a = c1 * L.s1 + c2 * L.s2 + row * L.srow;
        if (UNPACK) dst[a] = bswap64(src[i]);
        else dst[i] = bswap64(src[a]);
    }
}

int fits_convert(bool unpack, int64_t nrows, int nc1, int nc2, int64_t s1, int64_t s2, int64_t srow, const void *src, void *dst)
{
    HX_TRY(ensure_ready());
    if (nrows < 0 || nc1 < 1 || nc2 < 1 || !src || !dst) return fail(HX_ERR_ARG, "hx_fits_%s_f64: bad arguments", unpack ? "unpack" : "pack");
    if (nrows == 0) return HX_OK;
    const size_t n = (size_t)nrows * nc1 * nc2;
    // extent of the array side (strides are non-negative by contract)
    if (s1 < 0 || s2 < 0 || srow < 0) return fail(HX_ERR_ARG, "hx_fits: negative stride");
    const size_t extent = (size_t)((nc1 - 1) * s1 + (nc2 - 1) * s2 + (nrows - 1) * srow + 1);
    InView vin;
    OutView vout;
    HX_TRY(vin.bind(src, sizeof(double) * (unpack ? n : extent)));
    HX_TRY(vout.bind(dst, sizeof(double) * (unpack ? extent : n)));
    FitsLayout L = {nrows, nc1, nc2, s1, s2, srow};
    ProfScope ps(unpack ? "fits_unpack" : "fits_pack");
    if (unpack)
        hipLaunchKernelGGL(k_fits_f64<true>, dim3(4096), dim3(256), 0, rt().stream, L, vin.as<unsigned long long>(), vout.as<unsigned long long>());
    else
        hipLaunchKernelGGL(k_fits_f64<false>, dim3(4096), dim3(256), 0, rt().stream, L, vin.as<unsigned long long>(), vout.as<unsigned long long>());
    HX_HIP(hipGetLastError());
    HX_TRY(vout.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

}  // namespace
}  // namespace hx

extern "C" int hx_fits_unpack_f64(int64_t nrows, int nc1, int nc2, int64_t s1, int64_t s2, int64_t srow, const void *table, double *array)
{
    return hx::fits_convert(true, nrows, nc1, nc2, s1, s2, srow, table, array);
}

extern "C" int hx_fits_pack_f64(int64_t nrows, int nc1, int nc2, int64_t s1, int64_t s2, int64_t srow, const double *array, void *table)
{
    return hx::fits_convert(false, nrows, nc1, nc2, s1, s2, srow, array, table);
}
