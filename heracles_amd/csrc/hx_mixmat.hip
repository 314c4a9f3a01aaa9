// hx_mixmat.hip -- Gauss-Legendre nodes, Wigner-d tables and mode-coupling (mixing) matrices.
//
// Replaces convolvecl.mixmat / mixmat_eb as called at heracles/twopoint.py:378-388:
//   M_{l1 l2} = (2 l2 + 1)/(4 pi) sum_{l3} (2 l3 + 1) W_{l3} (l1 l2 l3; s1 -s1 0)(l1 l2 l3; s2 -s2 0)
// evaluated in its dense quadrature form:
//   xi(x) = sum_{l3} (2 l3 + 1)/(4 pi) W_{l3} P_{l3}(x)
//   G^{(ab)} = D^{(ab)T} diag(w xi) D^{(ab)} diag((2 l2 + 1)/2),   D^{(ab)}[k][l] = d^l_{ab}(x_k)
// on N >= (l1max + l2max + l3max)/2 + 1 Gauss-Legendre nodes, which is exact.  The
// (l, l') contraction is a symmetric FP64 GEMM on v_mfma_f64_16x16x4_f64; this is the only
// place in the engine that is GEMM-shaped, so the only place MFMA is used as a GEMM.
#include <algorithm>
#include <cmath>

#include "hx_common.h"

namespace hx {

typedef double double4_t __attribute__((ext_vector_type(4)));

constexpr int GB = 128;   // block tile edge (l, l')
constexpr int GK = 32;    // nodes per k tile

// ---- Gauss-Legendre nodes: Newton on P_n, one thread per node pair ---------------------
// xlo (may be null): the part of every node that a double cannot hold -- node = x + xlo, xlo = -P_n(x) / P_n'(x) at the converged
// double, accurate to ~1 % (P_n(x) ~ P_n' * 1e-16 stands well above the rounding noise of its recurrence).  The mixing-matrix tables
// need it: next to the poles d^l(x)' ~ l^2 / 2, so rounding a node to a double shifts d^l there by l^2 / 2 * 5e-17 ~ 1e-9 for l = 6000,
// coherently over all l of a table -- 2.7e-10 on the diagonal of the L = 4096 matrix of a mask whose correlation function peaks at
// theta = 0 (found by tests/test_gpu_mixmat.py::test_mixmat_blocks_at_high_l_vs_3j; the weights are good to 1e-16 as they are).
__global__ void k_gauss_legendre(int n, double *__restrict__ x, double *__restrict__ w, double *__restrict__ xlo)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (n + 1) / 2) return;
    double t = cospi((i + 0.75) / (n + 0.5));
    double dp = 1.0, tl = 0.0, pn = 0.0, om = 1.0;
    for (int it = 0; it < 10; ++it) {
        double p0 = 1.0, p1 = t;
        for (int k = 2; k <= n; ++k) {
            double p2 = ((2.0 * k - 1.0) * t * p1 - (k - 1.0) * p0) / k;
            p0 = p1;
            p1 = p2;
        }
        dp = n * (t * p1 - p0) / (t * t - 1.0);
        double dt = p1 / dp;
        t -= dt;
        // (|t| <= 1: an absolute threshold.  Relative to |t| the nodes next to 0 never met it and every thread ran all ten O(n) sweeps:
        // 7.7 ms of the L = 6144 build; converged Newton steps do not move the node any more)
        if (fabs(dt) <= 4e-16) break;
    }
    {
        // P_n at the converged node in double-double: in plain doubles the value (~ P_n' x 1e-16) drowns in the rounding noise of the
        // recurrence (~ n x 1e-16) everywhere but next to the poles, and a correction made of noise is worse than none
#pragma clang fp contract(off)  // (error-free transformations: every product and sum below rounds on its own)
        double ph0 = 1.0, pl0 = 0.0, ph1 = t, pl1 = 0.0;
        for (int k = 2; k <= n; ++k) {
            const double a = 2.0 * k - 1.0, b = k - 1.0;
            // u = t * p1 (dd x double), v = a * u, w = b * p0, d = v - w, p2 = d / k
            double uh = ph1 * t, ul = fma(ph1, t, -uh) + pl1 * t;
            double vh = uh * a, vl = fma(uh, a, -vh) + ul * a;
            double wh = ph0 * b, wl = fma(ph0, b, -wh) + pl0 * b;
            double sh = vh - wh, bb = sh - vh, sl = (vh - (sh - bb)) + (-wh - bb);  // two_sum(vh, -wh)
            sl += vl - wl;
            double dh = sh + sl, dl = sl - (dh - sh);                              // quick_two_sum
            const double q1 = dh / k, r = fma(-q1, (double)k, dh), q2 = (r + dl) / k;
            ph0 = ph1; pl0 = pl1;
            ph1 = q1 + q2; pl1 = q2 - (ph1 - q1);
        }
        // (1 - t^2 as (1 - t)(1 + t): t * t - 1 loses 7e-10 of its value next to the poles, and the weights with it)
        om = (1.0 - t) * (1.0 + t);
        dp = -n * (t * ph1 - ph0) / om;
        pn = ph1 + pl1;
        tl = -pn / dp;
    }
    // the weight belongs to the node t + tl: evaluated at the rounded t it is off by up to 6e-10 next to the poles
    // (d ln w / dx = -2 x / (1 - x^2)); P_n'(t + tl) = P_n' + tl P_n'' with (1 - t^2) P_n'' = 2 t P_n' - n (n + 1) P_n
    const double dps = dp + tl * (2.0 * t * dp - (double)n * (n + 1.0) * pn) / om;
    const double ww = 2.0 / ((om - 2.0 * t * tl) * dps * dps);
    if ((n & 1) && i == n / 2) { t = 0.0; tl = 0.0; }
    x[i] = -t; x[n - 1 - i] = t;
    w[i] = ww; w[n - 1 - i] = ww;
    if (xlo) { xlo[i] = -tl; xlo[n - 1 - i] = tl; }
}

// ---- Wigner-d tables: out[l * sl + k * sk] = d^l_{ab}(x_k), l = 0 .. lmax, in double-double arithmetic -------------------------------
// coef[l]: d^{l+1} = (c1x x + c1c) d^l - c2 d^{l-1}
// At l ~ 4000-6000 a table value next to the poles carries ~l^1.5 x 1e-16 of rounding noise from the upward recurrence (every
// step's rounding error is carried on by a solution that grows like l there), and the noise is alike for the ~100 polar nodes,
// which carry the whole integrand when the mask's correlation function peaks at theta = 0: 5e-11 on a diagonal element of 10.8 at
// L = 4096 (with exact nodes and weights; tests/test_gpu_mixmat.py::test_mixmat_blocks_at_high_l_vs_3j).  In double-double the noise is
// ~1e-32 l^1.5; the coefficients come as (hi, lo) pairs from the host's long doubles, the node as x + xlo.  One thread per node, a
// dependent chain of ~16 operations per step: ~0.4 ms per table at L = 6144, once per context (measured: 2.1 ms per table under rocprofv3, profiles/r06_kernel_stats.csv).
struct DD {
    double h, l;
};
__device__ __forceinline__ DD dd_quick(double a, double b)
{
#pragma clang fp contract(off)
    const double s = a + b;
    return DD{s, b - (s - a)};
}
__device__ __forceinline__ DD dd_add(DD a, DD b)
{
#pragma clang fp contract(off)
    const double s = a.h + b.h, bb = s - a.h;
    const double e = ((a.h - (s - bb)) + (b.h - bb)) + (a.l + b.l);
    return dd_quick(s, e);
}
__device__ __forceinline__ DD dd_mul(DD a, DD b)
{
#pragma clang fp contract(off)
    const double p = a.h * b.h;
    const double e = fma(a.h, b.h, -p) + (a.h * b.l + a.l * b.h);
    return dd_quick(p, e);
}
__device__ __forceinline__ DD dd_neg(DD a) { return DD{-a.h, -a.l}; }

struct WigCoefDD {
    double c1xh, c1xl, c1ch, c1cl, c2h, c2l;
};
// out[l * sl + k * sk] = d^l_{ab}(x_k + xlo_k), l = 0..lmax; s6: sqrt(6) / 4 as (hi, lo)
__global__ void k_wigner_table_dd(int lmax, int a, int b, int n, const double *__restrict__ x, const double *__restrict__ xlo,
                                  const WigCoefDD *__restrict__ coef, double *__restrict__ out, long long sl, long long sk, double s6h,
                                  double s6l)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const DD xx = dd_quick(x[k], xlo[k]);
    const DD one = DD{1.0, 0.0};
    const DD om = dd_add(one, dd_neg(xx)), op = dd_add(one, xx);
    const int l0 = max(abs(a), abs(b));
    DD d0;
    if (a == 0 && b == 0) d0 = one;
    else if (a == 2 && b == 0) d0 = dd_mul(DD{s6h, s6l}, dd_mul(om, op));
    else if (a == 2 && b == 2) { d0 = dd_mul(op, op); d0.h *= 0.25; d0.l *= 0.25; }
    else if (a == 1 && b == 1) { d0 = op; d0.h *= 0.5; d0.l *= 0.5; }     // d^1_{11}   (heracles/transforms.py:68-73)
    else if (a * b == -1) { d0 = om; d0.h *= 0.5; d0.l *= 0.5; }          // d^1_{-1,1} = d^1_{1,-1}
    else { d0 = dd_mul(om, om); d0.h *= 0.25; d0.l *= 0.25; }  // (2,-2)
    for (int l = 0; l < l0 && l <= lmax; ++l) out[l * sl + k * sk] = 0.0;
    if (l0 > lmax) return;
    DD dp = DD{0.0, 0.0}, dc = d0;
    out[l0 * sl + k * sk] = dc.h + dc.l;
    for (int l = l0; l < lmax; ++l) {
        const WigCoefDD c = coef[l];
        const DD t = dd_add(dd_mul(DD{c.c1xh, c.c1xl}, xx), DD{c.c1ch, c.c1cl});
        const DD dn = dd_add(dd_mul(t, dc), dd_neg(dd_mul(DD{c.c2h, c.c2l}, dp)));
        dp = dc;
        dc = dn;
        out[(l + 1) * sl + k * sk] = dc.h + dc.l;
    }
}

// ---- node weights s_k = w_k xi(x_k + xlo_k), xi = sum_l (2 l + 1) / (4 pi) W_l P_l --------------------------------------------------
// (Round 5: k_weight_xi_dd, one thread per node running the P_l recurrence and the sum in double-double: 1.6 ms per mask at L = 6144.)
// As a matrix-vector product over the (0,0) table (round 6): T0[l][k] = P_l(x_k + xlo_k) is in HBM anyway (built in
// double-double, rounded once), so  s_k = w_k sum_l (2 l + 1) / (4 pi) W_l T0[l][k]  needs no recurrence -- that kernel ran ONE
// dependent chain of 6144 double-double steps per node, 1.6 ms per mask at L = 6144 whatever the GPU could do beside it (9217 nodes:
// 145 waves), three quarters of a binned mixing-matrix key.  Here a block takes 64 nodes x a slice of the multipoles (4 sub-slices
// of it across its waves), every partial sum is compensated (two_sum) and the slices are added in fixed order: the rounding of the table
// entries is random from l to l (1e-16 of each term), unlike the node errors the double-double tables exist for.
constexpr int XI_SLICES = 8;
__global__ __launch_bounds__(256) void k_xi_partial(int l3max, int kpad, const double *__restrict__ cl, const double *__restrict__ T0,
                                                    double2 *__restrict__ part)
{
#pragma clang fp contract(off)
    __shared__ double2 red[4][64];
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const int nl = l3max + 1, per = (nl + XI_SLICES * 4 - 1) / (XI_SLICES * 4);
    const int l0 = (blockIdx.y * 4 + sub) * per, l1 = min(l0 + per, nl);
    double sum = 0.0, comp = 0.0;
    if (k < kpad) {
        for (int l = l0; l < l1; ++l) {
            const double v = ((2.0 * l + 1.0) / (4.0 * M_PI) * cl[l]) * T0[(long long)l * kpad + k];
            const double t = sum + v, bb = t - sum;
            comp += (sum - (t - bb)) + (v - bb);
            sum = t;
        }
    }
    red[sub][threadIdx.x & 63] = make_double2(sum, comp);
    __syncthreads();
    if (sub == 0 && k < kpad) {
        DD a = dd_quick(red[0][threadIdx.x].x, red[0][threadIdx.x].y);
#pragma unroll
        for (int q = 1; q < 4; ++q) a = dd_add(a, dd_quick(red[q][threadIdx.x].x, red[q][threadIdx.x].y));
        part[(long long)blockIdx.y * kpad + k] = make_double2(a.h, a.l);
    }
}
__global__ void k_xi_finish(int n, int kpad, const double *__restrict__ w, const double2 *__restrict__ part, double *__restrict__ s)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    DD a = DD{part[k].x, part[k].y};
    for (int q = 1; q < XI_SLICES; ++q) a = dd_add(a, DD{part[(long long)q * kpad + k].x, part[(long long)q * kpad + k].y});
    s[k] = w[k] * (a.h + a.l);
}

// ---- symmetric GEMM: G[i][j] = colscale[j] * sum_k T[i][k] s[k] T[j][k] -----------------
// T: [rows_pad][kpad] (zero padded), block tiles (bi <= bj) from `tiles` ((-1, -1) = padding entry); writes both halves.
// Round 4: the round-1 kernel held 350 registers (one work-group per CU) and loaded, stored and multiplied in
// sequence: 41 TFLOP/s executed.  Now
//   * two work-groups per CU (<= 256 registers, 72 KiB of LDS each): one group's loads, LDS stores and barrier sit under the other's
//     matrix instructions, and two waves per SIMD keep the matrix pipe at 97 % of its rate (tools/ubench_mblock.hip);
//   * the k tiles (16 nodes) are double-buffered in LDS: the global loads of tile t + 1 are requested before the matrix instructions of
//     tile t and stored behind them -- one barrier per tile;
//   * every LDS operand read is 128 bits wide: the two nodes of a lane's double2 feed two consecutive matrix instructions (the
//     contraction over the nodes may take them in any order, as long as A and B agree: instruction h of a pair contracts the nodes
//     8 q + 2 (lane >> 4) + h); rows are padded to 18 doubles, which spreads the 16 rows of a read over all 64 banks.
constexpr int GKT = 16;          // nodes per k tile of the GEMM (GK = 32 stays the padding unit of the tables)
constexpr int GLT = GKT + 2;     // LDS row stride in doubles
// Where this kernel stands (round 4, work-group timelines of a stamped build, tools/analyse_gemm_stamps.py; L = 6144: 1225 tiles
// on 512 resident groups): a tile takes 2.41 ms when two groups share a CU (1.9-2.9; 2.20 would be the matrix pipe's rate at the 2.15 GHz
// the part holds here: 0.92) and 2.24 ms ALONE on a CU (a build with ONE group per CU: 10.0 ms per product, 0.55 of the pipe -- its loads run ONE k
// tile = 4096 cycles ahead, less than a trip to memory under load, its LDS reads wait behind every barrier with no other wave to cover
// them, and 243 registers leave no room for a second tile in flight); the launch is two full rounds, to 4.3-4.8 ms, and a last round of
// 20-28 tiles per XCD, most of them alone on their CU, to 6.75 ms: matrix pipe busy 0.78.  k_mixmat_gemm_dma below is the answer to the
// lone group (1.44 ms); this kernel stays for products whose A side is scaled on the way (hx_pinv).  Measured and not kept on either kernel: the tiles of the last round as two 128 x 64 halves (6.8 ms) and as the two
// halves of the NODES added into a zeroed G (two addends commute: repeatable bit for bit; 6.55 ms here, 13.0-13.2 against 12.9-13.0 ms per
// two products on the other kernel: the rounds are blurred by the spread of the tile times, the memset and the atomics cost what the
// halves gain).  Equal ranges of k tiles dealt to persistent groups (stream-K) would end every group together, but groups that stand
// at different nodes of their tiles no longer share the rows of T through their XCD's L2, which is what the tile order is for.
// G[i][j] = sum_k T[i][k] s[k] T2[j][k] for every listed tile, colscale may be null (the last product of hx_pinv: V diag(1 / sigma^2) W^T).
// (Until round 6 a symmetric instantiation of this kernel was the A/B reference of the mixing-matrix product: tools/patches/r05_switches.patch.)
__global__ __launch_bounds__(256, 2) void k_mixmat_gemm(const double *__restrict__ T, const double *__restrict__ T2, int kpad,
                                                        const double *__restrict__ s,
                                                        const int2 *__restrict__ tiles, int n1, int n2,
                                                        const double *__restrict__ colscale,
                                                        double *__restrict__ G, long long ldg)
{
    __shared__ double As[2][GB][GLT], Bs[2][GB][GLT];
    const int2 tl = tiles[blockIdx.x];
    const int bi = tl.x, bj = tl.y;
    if (bi < 0) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int wr = w >> 1, wc = w & 1;
    double4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

    // staging: element pair e = t + 256 u of the 128 x 16 tile: row e >> 3, nodes 2 (e & 7), + 1 (8 threads read 128 contiguous bytes of a row)
    const int srow = t >> 3, sc2 = (t & 7) * 2;
    // buffer loads: the row block of a tile is a buffer resource (scalar registers), a thread brings one 32-bit byte offset per row
    // quarter and the k tile is the scalar offset -- no address arithmetic on the vector unit (as 64-bit per-thread addresses every load
    // cost a vector instruction in the matrix stream: ~12 cycles of pipe time each)
    const __amdgpu_buffer_rsrc_t Ra = __builtin_amdgcn_make_buffer_rsrc((void *)(T + (long long)bi * GB * kpad), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t Rb = __builtin_amdgcn_make_buffer_rsrc((void *)(T2 + (long long)bj * GB * kpad), 0, 0x7fffffff, 0x00020000);
    const int so = (int)(((long long)srow * kpad + sc2) * sizeof(double)), sq = (int)((long long)32 * kpad * sizeof(double));
    const int so1 = so + sq, so2 = so + 2 * sq, so3 = so + 3 * sq;
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    auto bload = [](const __amdgpu_buffer_rsrc_t r, int voff, int soff) __attribute__((always_inline)) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        double2 d;
        __builtin_memcpy(&d, &v, sizeof(d));
        return d;
    };
    // (no lambdas around the staging registers, no arithmetic on them before the store: either makes hipcc keep them in scratch and wait
    // for the loads at the top of the loop)
    double2 va0, va1, va2, va3, vb0, vb1, vb2, vb3, sa;
#define HX_GLOAD(k0)                                                                      \
    do {                                                                                  \
        sa = *reinterpret_cast<const double2 *>(s + (k0) + sc2);                          \
        va0 = bload(Ra, so, (k0) * 8);                                                    \
        va1 = bload(Ra, so1, (k0) * 8);                                                   \
        va2 = bload(Ra, so2, (k0) * 8);                                                   \
        va3 = bload(Ra, so3, (k0) * 8);                                                   \
        vb0 = bload(Rb, so, (k0) * 8);                                                    \
        vb1 = bload(Rb, so1, (k0) * 8);                                                   \
        vb2 = bload(Rb, so2, (k0) * 8);                                                   \
        vb3 = bload(Rb, so3, (k0) * 8);                                                   \
    } while (0)
#define HX_LSTORE(buf)                                                                                               \
    do {                                                                                                             \
        *reinterpret_cast<double2 *>(&As[buf][srow][sc2]) = make_double2(va0.x * sa.x, va0.y * sa.y);                \
        *reinterpret_cast<double2 *>(&As[buf][srow + 32][sc2]) = make_double2(va1.x * sa.x, va1.y * sa.y);           \
        *reinterpret_cast<double2 *>(&As[buf][srow + 64][sc2]) = make_double2(va2.x * sa.x, va2.y * sa.y);           \
        *reinterpret_cast<double2 *>(&As[buf][srow + 96][sc2]) = make_double2(va3.x * sa.x, va3.y * sa.y);           \
        *reinterpret_cast<double2 *>(&Bs[buf][srow][sc2]) = vb0;                                                     \
        *reinterpret_cast<double2 *>(&Bs[buf][srow + 32][sc2]) = vb1;                                                \
        *reinterpret_cast<double2 *>(&Bs[buf][srow + 64][sc2]) = vb2;                                                \
        *reinterpret_cast<double2 *>(&Bs[buf][srow + 96][sc2]) = vb3;                                                \
    } while (0)
    HX_GLOAD(0);
    HX_LSTORE(0);
    __syncthreads();
    const int nk = kpad / GKT;
    const int arow = wr * 64 + (lane & 15), brow = wc * 64 + (lane & 15), kcol = 2 * (lane >> 4);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const int knext = (kt + 1 < nk ? kt + 1 : kt) * GKT;  // (unconditional: the last iteration re-requests its own tile)
        HX_GLOAD(knext);
#pragma unroll
        for (int q = 0; q < GKT / 8; ++q) {
            double2 a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *reinterpret_cast<const double2 *>(&As[buf][arow + 16 * i][8 * q + kcol]);
                b[i] = *reinterpret_cast<const double2 *>(&Bs[buf][brow + 16 * i][8 * q + kcol]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(h ? a[i].y : a[i].x, h ? b[j].y : b[j].x, acc[i][j], 0, 0, 0);
        }
        HX_LSTORE(buf ^ 1);  // (buffer buf ^ 1 was last read in iteration kt - 1, closed by its barrier)
        __syncthreads();
    }
#undef HX_GLOAD
#undef HX_LSTORE
    // D layout: row = (lane>>4) + 4*reg, col = lane&15
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = bi * GB + wr * 64 + i * 16 + (lane >> 4) + 4 * r;
                const int gj = bj * GB + wc * 64 + j * 16 + (lane & 15);
                const double v = acc[i][j][r];
                if (gi < n1 && gj < n2) G[(long long)gi * ldg + gj] = colscale ? v * colscale[gj] : v;
            }
}

// ---- the same product with the k tiles brought in by loads that write LDS directly (round 4, second session) ---------------------
// Ts = T diag(s) is formed once per mask (k_scale_table: the loads cannot multiply on the way), then  G = Ts T2^T.  A stage is 8
// nodes of the 128 + 128 rows of a tile: 64-byte rows, 16 KiB per stage, FOUR stages per work-group (64 KiB: two groups per CU as
// before), three of them in flight -- 6144 cycles of matrix instructions for a group that is alone on its CU, where the register-
// staged kernel has 4096 and reaches 0.55 of the pipe (see above).  One `global_load_lds_dwordx4` of a wave fills 16 rows: lane l
// writes row l >> 2, 16-byte position l & 3 (the LDS image of such a load is lane-linear), and FETCHES chunk (l & 3) ^ ((row >> 2) & 3)
// of that row: the 16 rows of an operand read (fixed chunk, rows r .. r + 15) then hit 16 different bank groups.  Per stage a wave
// issues 2 + 2 loads; `s_waitcnt vmcnt(8)` (two younger stages) retires its own pieces of the oldest stage, the barrier behind it
// tells every wave that all pieces have landed AND that stage kt - 1 has been read by everyone, so its buffer is re-filled with
// stage kt + 3 right behind the barrier.
constexpr int DK = 8;  // nodes per stage (four stages)
// one 16-byte load per lane straight into LDS: lane l's data lands at lds_dst + 16 l (lds_dst through M0, which the compiler owns:
// saved and restored); address = scalar base + 32-bit lane offset -- hipcc's builtin takes a per-lane 64-bit pointer and pays a vector
// add per load for it
__device__ __forceinline__ void glds16(const double *sbase, int voff_bytes, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void *)p;
}
// Shader clock UNDER this kernel: every 64th tile adds its (shader ticks, 100 MHz ticks) from first load to last store; hx_mixmat_gemm_clock
// reads and clears the sums (two scalar clock reads and two atomics per sampled tile: 1225 tiles -> ~20 samples per product)
__device__ unsigned long long g_gemm_clk[2];
template <bool SYM>
__global__ __launch_bounds__(256, 2) void k_mixmat_gemm_dma(const double *__restrict__ Ts, const double *__restrict__ T2, int kpad,
                                                            const int2 *__restrict__ tiles, int n1, int n2,
                                                            const double *__restrict__ colscale, double *__restrict__ G, long long ldg)
{
    // one array per stage and operand: hipcc orders an LDS read behind EVERY load-to-LDS in flight that it cannot tell apart from it
    // (s_waitcnt vmcnt(0) in front of the reads: nothing left in flight); distinct LDS variables it can
    // (one array per STAGE, both operands in it: with eight arrays the compiler runs out of slots to track them and one stage gets its vmcnt(0) back)
    __shared__ double S0[2][GB][DK], S1[2][GB][DK], S2[2][GB][DK], S3[2][GB][DK];
    const int2 tl = tiles[blockIdx.x];
    const int bi = tl.x, bj = tl.y;
    if (bi < 0) return;
    const bool sample_clk = (blockIdx.x & 63) == 33;
    unsigned long long clk_t0 = 0, clk_r0 = 0;
    if (sample_clk) { clk_t0 = __builtin_amdgcn_s_memtime(); clk_r0 = __builtin_amdgcn_s_memrealtime(); }
    const int t = threadIdx.x, lane = t & 63;
    const int w = __builtin_amdgcn_readfirstlane(t >> 6);  // (a scalar: the LDS destinations of a wave's loads go through M0)
    const int wr = w >> 1, wc = w & 1;
    double4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // this lane's pieces of a stage: rows w 32 + i 16 + (lane >> 2), i = 0, 1.  Scalar base + 32-bit lane offset: the base advances on the
    // scalar unit (with per-lane 64-bit pointers every load cost a vector add and a v_readfirstlane inside the matrix stream: ~12
    // cycles of pipe time each, 8 per stage)
    const int lrow = lane >> 2;
    const double *pa = Ts + (long long)bi * GB * kpad, *pb = T2 + (long long)bj * GB * kpad;
    int off[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = w * 32 + i * 16 + lrow;
        const int chunk = (lane & 3) ^ ((row >> 2) & 3);
        off[i] = (row * kpad + 2 * chunk) * (int)sizeof(double);  // bytes
    }
#define HX_DMA_ISSUE(S, kt)                                                                                                               \
    do {                                                                                                                                  \
        const double *qa_ = pa + (long long)(kt) * DK, *qb_ = pb + (long long)(kt) * DK;                                                  \
        _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                                                                \
            glds16(qa_, off[i_], lds_addr(&S[0][w * 32 + i_ * 16][0]));                                                                   \
            glds16(qb_, off[i_], lds_addr(&S[1][w * 32 + i_ * 16][0]));                                                                   \
        }                                                                                                                                 \
    } while (0)
    const int nk = kpad / DK;  // (a multiple of 4: kpad is a multiple of 32)
    HX_DMA_ISSUE(S0, 0);
    HX_DMA_ISSUE(S1, 1);
    HX_DMA_ISSUE(S2, 2);
    // operand reads: row (lane & 15) of a 16-row block, nodes 2 c, 2 c + 1 with c = lane >> 4, at position c ^ ((row >> 2) & 3)
    const int arow = wr * 64 + (lane & 15), brow = wc * 64 + (lane & 15);
    const int apos = 2 * ((lane >> 4) ^ ((arow >> 2) & 3)), bpos = 2 * ((lane >> 4) ^ ((brow >> 2) & 3));  // (+ 16 i keeps (row >> 2) & 3)
    // The operands of stage kt + 1 are read while the matrix instructions of stage kt run (two register sets): a group that is alone
    // on its CU has no other wave to cover the LDS latency behind every barrier.  The reads are inline assembly: behind the loop's
    // back edge hipcc cannot count the loads to LDS in flight and drains them all (vmcnt(0)) in front of ordinary LDS reads.
    typedef double v2d __attribute__((ext_vector_type(2)));
    v2d ra[2][4], rb[2][4];
#define HX_DMA_READ(S, q)                                                                                                                 \
    do {                                                                                                                                  \
        const unsigned la_ = lds_addr(&S[0][arow][apos]), lb_ = lds_addr(&S[1][brow][bpos]);                                              \
        asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\tds_read_b128 %3, %8 offset:3072\n\t" \
                     "ds_read_b128 %4, %9\n\tds_read_b128 %5, %9 offset:1024\n\tds_read_b128 %6, %9 offset:2048\n\tds_read_b128 %7, %9 offset:3072"     \
                     : "=&v"(ra[q][0]), "=&v"(ra[q][1]), "=&v"(ra[q][2]), "=&v"(ra[q][3]), "=&v"(rb[q][0]), "=&v"(rb[q][1]), "=&v"(rb[q][2]), "=&v"(rb[q][3]) \
                     : "v"(la_), "v"(lb_)                                                                                                 \
                     : "memory");                                                                                                         \
    } while (0)
    // (the registers are operands of the wait, so that no matrix instruction that reads them is scheduled above it)
#define HX_DMA_LANDED(q)                                                                                                                  \
    asm volatile("s_waitcnt lgkmcnt(0)"                                                                                                   \
                 : "+v"(ra[q][0]), "+v"(ra[q][1]), "+v"(ra[q][2]), "+v"(ra[q][3]), "+v"(rb[q][0]), "+v"(rb[q][1]), "+v"(rb[q][2]), "+v"(rb[q][3]) \
                 :: "memory")
    // iteration kt: stage kt is in register set q; (Sr) = the buffer of stage kt + 1 (read now), (Sn) = that of stage kt + 3 = kt - 1 (refilled now)
#define HX_DMA_STAGE(Sr, Sn, q, kt)                                                                                                       \
    do {                                                                                                                                  \
        const int left_ = nk - 1 - (kt); /* younger stages issued so far: min(left_, 2); stage kt + 1 must have landed */                   \
        if (left_ >= 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                                                  \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                                             \
        __builtin_amdgcn_s_barrier();                                                                                                     \
        if ((kt) + 3 < nk) HX_DMA_ISSUE(Sn, (kt) + 3);                                                                                    \
        if ((kt) + 1 < nk) HX_DMA_READ(Sr, (q) ^ 1);                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                                                \
        _Pragma("unroll") for (int h_ = 0; h_ < 2; ++h_)                                                                                  \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                                                              \
                _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_)                                                                          \
                    acc[i_][j_] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[q][i_][h_], rb[q][j_][h_], acc[i_][j_], 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0); /* (the wait stays BEHIND the matrix instructions) */                                          \
        HX_DMA_LANDED((q) ^ 1);                                                                                                           \
    } while (0)
    // stage 0 into set 0
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    HX_DMA_READ(S0, 0);
    HX_DMA_LANDED(0);
    for (int kt = 0; kt < nk; kt += 4) {
        HX_DMA_STAGE(S1, S3, 0, kt);
        HX_DMA_STAGE(S2, S0, 1, kt + 1);
        HX_DMA_STAGE(S3, S1, 0, kt + 2);
        HX_DMA_STAGE(S0, S2, 1, kt + 3);
    }
#undef HX_DMA_READ
#undef HX_DMA_LANDED
#undef HX_DMA_STAGE
#undef HX_DMA_ISSUE
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = bi * GB + wr * 64 + i * 16 + (lane >> 4) + 4 * r;
                const int gj = bj * GB + wc * 64 + j * 16 + (lane & 15);
                const double v = acc[i][j][r];
                if (gi < n1 && gj < n2) G[(long long)gi * ldg + gj] = colscale ? v * colscale[gj] : v;
                if (SYM && bi != bj && gj < n1 && gi < n2) G[(long long)gj * ldg + gi] = v * colscale[gi];
            }
    if (sample_clk && threadIdx.x == 0) {
        atomicAdd(&g_gemm_clk[0], (unsigned long long)__builtin_amdgcn_s_memtime() - clk_t0);
        atomicAdd(&g_gemm_clk[1], (unsigned long long)__builtin_amdgcn_s_memrealtime() - clk_r0);
    }
}

// Ts[r][k] = T[r][k] s[k]
__global__ __launch_bounds__(256) void k_scale_table(long long n2, int kpad2, const double2 *__restrict__ T, const double2 *__restrict__ s, double2 *__restrict__ Ts)
{
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long stp = (long long)gridDim.x * blockDim.x;
    for (; i < n2; i += stp) {
        const double2 v = T[i], f = s[i % kpad2];
        Ts[i] = make_double2(v.x * f.x, v.y * f.y);
    }
}

// G (n1 x n2, leading dimension ldg) = T diag(s) T2^T for two zero-padded tables T [rows1_pad][kpad], T2 [rows2_pad][kpad] (rows padded to
// multiples of 128, kpad to a multiple of 32) on the stream of the library (hx_svd.hip)
int launch_gemm_tst(const double *T, int rows1_pad, const double *T2, int rows2_pad, int kpad, const double *s, int n1, int n2, double *G, long long ldg)
{
    std::vector<int2> tiles;
    for (int i = 0; i < rows1_pad / GB; ++i)
        for (int j = 0; j < rows2_pad / GB; ++j) tiles.push_back(make_int2(i, j));
    DevBuf d_tiles;
    HX_TRY(d_tiles.alloc(sizeof(int2) * tiles.size()));
    HX_HIP(hipMemcpyAsync(d_tiles.p, tiles.data(), sizeof(int2) * tiles.size(), hipMemcpyHostToDevice, rt().stream));
    hipLaunchKernelGGL(k_mixmat_gemm, dim3((unsigned)tiles.size()), dim3(256), 0, rt().stream, T, T2, kpad, s, d_tiles.as<int2>(), n1, n2,
                       (const double *)nullptr, G, ldg);
    HX_HIP(hipGetLastError());
    HX_HIP(hipStreamSynchronize(rt().stream));  // the tile list dies with this scope
    return HX_OK;
}

// out0 = (a + b)/2, out1 = (a - b)/2, out2 = b   (a = G22 in out0, b = G2-2 in out2)
__global__ void k_eb_combine(long long n, double *__restrict__ o0, double *__restrict__ o1,
                             const double *__restrict__ o2)
{
    long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x;
    const long long st = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += st) {
        const double a = o0[i], b = o2[i];
        o0[i] = 0.5 * (a + b);
        o1[i] = 0.5 * (a - b);
    }
}

// ---- binned rows (heracles.twopoint.mixing_matrices(bins=, weights=): heracles/twopoint.py:391-397 -> heracles/result.py:124-248) ----
// The reference bins the OUTPUT multipole of every matrix on the host, column by column (5 s per (3, 6145, 6145) matrix).  Binning is
// linear in the rows, so with the mask-independent binned tables
//     Tb[b][k] = sum_{l in bin b} w_l d^l(x_k)
// the numerators of the binned matrix are the SKINNY product  sum_k Tb[b][k] s_k T[l2][k] (2 l2 + 1) / 2  --  (nbins x N)(N x L2):
// ~200x fewer flops than the full matrix at 32 bins, one read of the table from HBM (0.46 GB at L = 6144) and nbins x (l2max + 1)
// doubles to the host instead of 0.9 GB.
//
// k_bin_table: one thread per node, the rows of a bin in ascending l with a compensated sum (the terms oscillate in l and cancel).
__global__ __launch_bounds__(256) void k_bin_table(int kpad, const int *__restrict__ start, const int *__restrict__ rows, const double *__restrict__ rw,
                                                   const double *__restrict__ T, double *__restrict__ Tb)
{
#pragma clang fp contract(off)
    const int k = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (k >= kpad) return;
    double sum = 0.0, comp = 0.0;
    for (int e = start[b]; e < start[b + 1]; ++e) {
        const double v = rw[e] * T[(long long)rows[e] * kpad + k];
        const double t = sum + v, bb = t - sum;
        comp += (sum - (t - bb)) + (v - bb);  // two_sum
        sum = t;
    }
    Tb[(long long)b * kpad + k] = sum + comp;
}

// partial[ks][b][j] = sum_{k in share ks} A[b][k] B[j][k]:  A = Tb diag(s) [nbpad][kpad], B = T [rows_pad][kpad].
// A wave owns 16 columns j and MT tiles of 16 bins; a lane brings 32 contiguous bytes (4 nodes) of one row of each operand per
// step, so the four matrix instructions of a step contract the nodes k + 4 (lane >> 4) + {0, 1, 2, 3} -- any order is fine as long as
// A and B agree.  B (the big table) streams from HBM once, 128 contiguous bytes per row and step; A stays in L2.  The node range is cut
// into gridDim.y shares (enough work-groups to fill the chip); k_binned_finish adds them in fixed order.
template <int MT>
__global__ __launch_bounds__(256) void k_binned_gemm(const double *__restrict__ A, const double *__restrict__ B, int kpad, int kchunk, int nbpad, int l2pad,
                                                     double *__restrict__ partial)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r = lane & 15, kg = lane >> 4;
    const int c0 = (blockIdx.x * 4 + w) * 16, b0 = blockIdx.z * (MT * 16);
    const int k0 = blockIdx.y * kchunk, k1 = min(k0 + kchunk, kpad);
    const double *pb = B + (long long)(c0 + r) * kpad + kg * 4;
    const double *pa = A + (long long)(b0 + r) * kpad + kg * 4;
    double4_t acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = (double4_t){0.0, 0.0, 0.0, 0.0};
    int k = k0;
    for (; k + 64 <= k1; k += 64) {  // four steps of B in flight
        double4_t b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b[u] = *reinterpret_cast<const double4_t *>(pb + k + 16 * u);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            double4_t a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const double4_t *>(pa + (long long)t * 16 * kpad + k + 16 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int h = 0; h < 4; ++h) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u][h], b[u][h], acc[t], 0, 0, 0);
        }
    }
    for (; k < k1; k += 16) {
        const double4_t b = *reinterpret_cast<const double4_t *>(pb + k);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const double4_t a = *reinterpret_cast<const double4_t *>(pa + (long long)t * 16 * kpad + k);
#pragma unroll
            for (int h = 0; h < 4; ++h) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h], b[h], acc[t], 0, 0, 0);
        }
    }
    // D layout: row = (lane >> 4) + 4 * reg, col = lane & 15
    double *po = partial + ((long long)blockIdx.y * nbpad + b0) * l2pad + c0 + r;
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) po[(long long)(t * 16 + kg + 4 * q) * l2pad] = acc[t][q];
}

// out[b][j] = ratio(colscale[j] * sum_ks partial[ks][b][j], norm[b]), ratio(n, d) = n / d where n != 0, else 0 (the reference's rule:
// heracles/result.py:132-135).  EB: pa = the (2,2) product, pb = the (2,-2) product -> [0] = (a + b) / 2, [1] = (a - b) / 2, [2] = b.
template <bool EB>
__global__ __launch_bounds__(256) void k_binned_finish(int nbins, int n2, int nbpad, int l2pad, int ksplit, const double *__restrict__ pa,
                                                       const double *__restrict__ pb, const double *__restrict__ colscale,
                                                       const double *__restrict__ norm, double *__restrict__ out)
{
    const int j = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (j >= n2) return;
    const long long slab = (long long)nbpad * l2pad, at = (long long)b * l2pad + j;
    double a = 0.0, c = 0.0;
    for (int ks = 0; ks < ksplit; ++ks) {
        a += pa[ks * slab + at];
        if (EB) c += pb[ks * slab + at];
    }
    const double cs = colscale[j], nb = norm[b];
    auto ratio = [nb](double v) { return v != 0.0 ? v / nb : 0.0; };
    const long long o = (long long)b * n2 + j, sz = (long long)nbins * n2;
    if (EB) {
        a *= cs;
        c *= cs;
        out[o] = ratio(0.5 * (a + c));
        out[o + sz] = ratio(0.5 * (a - c));
        out[o + 2 * sz] = ratio(c);
    } else {
        out[o] = ratio(a * cs);
    }
}

// ---- host helpers --------------------------------------------------------------------
static void wigner_coefs_dd(int lmax, int a, int b, std::vector<WigCoefDD> &c)
{
    c.assign(lmax + 1, WigCoefDD{0, 0, 0, 0, 0, 0});
    auto split = [](long double v, double &h, double &l) {
        h = (double)v;
        l = (double)(v - (long double)h);
    };
    for (int l = 0; l <= lmax; ++l) {
        if (l == 0) { c[l].c1xh = 1.0; continue; }
        long double dl = l, lp = l + 1.0L;
        long double den = dl * sqrtl((lp * lp - (long double)a * a) * (lp * lp - (long double)b * b));
        if (den == 0.0L) continue;
        split((2 * dl + 1) * dl * lp / den, c[l].c1xh, c[l].c1xl);
        split(-(2 * dl + 1) * (long double)a * b / den, c[l].c1ch, c[l].c1cl);
        split(lp * sqrtl((dl * dl - (long double)a * a) * (dl * dl - (long double)b * b)) / den, c[l].c2h, c[l].c2l);
    }
}

int launch_gauss_legendre(int n, double *d_x, double *d_w, double *d_xlo)
{
    const int half = (n + 1) / 2;
    hipLaunchKernelGGL(k_gauss_legendre, dim3((half + 63) / 64), dim3(64), 0, rt().stream, n, d_x, d_w, d_xlo);
    HX_HIP(hipGetLastError());
    return HX_OK;
}

template <class T>
static int upload_vec(DevBuf &b, const std::vector<T> &v)
{
    HX_TRY(b.alloc(sizeof(T) * (v.empty() ? 1 : v.size())));
    if (!v.empty()) HX_HIP(hipMemcpy(b.p, v.data(), sizeof(T) * v.size(), hipMemcpyHostToDevice));
    return HX_OK;
}

struct GLCache {
    int n = 0;
    DevBuf x, w, xlo;  // node k = x[k] + xlo[k]
};

static int gl_nodes_device(int n, GLCache &c)
{
    if (c.n == n && c.x.p) return HX_OK;
    HX_TRY(c.x.alloc(sizeof(double) * n));
    HX_TRY(c.w.alloc(sizeof(double) * n));
    HX_TRY(c.xlo.alloc(sizeof(double) * n));
    HX_TRY(launch_gauss_legendre(n, c.x.as<double>(), c.w.as<double>(), c.xlo.as<double>()));
    c.n = n;
    return HX_OK;
}

// Everything of a mixing-matrix build that does not depend on the mask: Gauss-Legendre nodes, the Wigner-d tables
// T_ab[l][k] = d^l_ab(x_k) of the products asked for, the tile list and the column scaling (2 l2 + 1) / 2.  One context
// serves any number of masks: per mask only the node weights s_k = w_k xi(x_k) and one GEMM per product remain
// (the reference calls convolvecl once per (field pair, bin pair), heracles/twopoint.py:354-397, rebuilding all of it).
struct MixCtx {
    int l1max = -1, l2max = -1, l3max = -1, L = 0, n = 0, kpad = 0, rows_pad = 0;
    GLCache gl;
    DevBuf s, d_tiles, d_cs;
    size_t ntiles = 0;
    DevBuf T[4];            // (0,0), (2,0), (2,2), (2,-2)
    DevBuf Ts;              // T diag(s) of the product in hand (k_mixmat_gemm_dma)
    DevBuf xi_part;         // slices of the node weights of the mask in hand (k_xi_partial)
    int rows0_pad = 0;      // rows of T[0]: it also serves the node weights, whose sum runs to l3max
    bool have[4] = {false, false, false, false};
    // binned rows (hx_mixctx_set_bins): bin lists of the output multipole, binned tables Tb_t[nbpad][kpad] per product
    int nbins = 0, nbpad = 0, l2pad = 0, ksplit = 1, kchunk = 0;
    DevBuf b_start, b_rows, b_w, b_norm;
    DevBuf Tb[4], Tbs, part[2], b_out;  // (b_out: device image of a host destination, kept between calls)
    bool have_b[4] = {false, false, false, false};
    // page-locked landing buffer of a binned result on its way to a pageable destination (the GPU writes it by DMA; the bytes are then
    // copied on by the host): straight into a fresh numpy array, the runtime pins the caller's pages call by call -- 3 ms per 1.5 MB key in
    // a process that holds a plan's scratch, against 0.15 ms this way (tools/exp_mixmat_list_env.py)
    void *b_pin = nullptr;
    size_t b_pin_bytes = 0;
    MixCtx() = default;
    MixCtx(const MixCtx &) = delete;
    MixCtx &operator=(const MixCtx &) = delete;
    ~MixCtx()
    {
        if (b_pin) (void)hipHostFree(b_pin);
    }
};
static const int kAB[4][2] = {{0, 0}, {2, 0}, {2, 2}, {2, -2}};

static int mix_ctx_init(MixCtx &c, int l1max, int l2max, int l3max)
{
    hipStream_t st = rt().stream;
    c.l1max = l1max; c.l2max = l2max; c.l3max = l3max;
    c.L = std::max(l1max, l2max);
    c.n = (l1max + l2max + l3max) / 2 + 1;
    c.kpad = (c.n + GK - 1) / GK * GK;
    c.rows_pad = (c.L + 1 + GB - 1) / GB * GB;
    HX_TRY(gl_nodes_device(c.n, c.gl));
    HX_TRY(c.s.alloc(sizeof(double) * c.kpad));
    // Tile order: work-groups go to the 8 XCDs round-robin (blockIdx % 8), and an XCD runs 64 of them at a time (32 CUs x 2).  The
    // upper triangle is cut into super-tiles of 8 x 4 block tiles (12 row blocks of T for 32 tiles), the super-tiles are dealt to the
    // XCDs (largest first, to the least loaded) and every XCD walks its own list: the tiles that run side by side on an XCD share their
    // rows of T through its L2 instead of each streaming 2 x 9.5 MB from memory.  (-1, -1) pads the shorter lists.
    const int nb = c.rows_pad / GB;
    constexpr int SR = 8, SC = 4, NX = 8;
    std::vector<std::vector<int2>> supers;
    for (int I = 0; I < nb; I += SR)
        for (int J = I / SC * SC; J < nb; J += SC) {
            std::vector<int2> m;
            for (int i = I; i < std::min(I + SR, nb); ++i)
                for (int j = std::max(J, i); j < std::min(J + SC, nb); ++j) m.push_back(make_int2(i, j));
            if (!m.empty()) supers.push_back(m);
        }
    std::stable_sort(supers.begin(), supers.end(), [](const std::vector<int2> &a, const std::vector<int2> &b) { return a.size() > b.size(); });
    std::vector<std::vector<int2>> per(NX);
    for (const auto &m : supers) {
        int x = 0;
        for (int q = 1; q < NX; ++q)
            if (per[q].size() < per[x].size()) x = q;
        per[x].insert(per[x].end(), m.begin(), m.end());
    }
    size_t longest = 0;
    for (const auto &l : per) longest = std::max(longest, l.size());
    std::vector<int2> tiles(longest * NX, make_int2(-1, -1));
    for (int x = 0; x < NX; ++x)
        for (size_t k = 0; k < per[x].size(); ++k) tiles[k * NX + x] = per[x][k];
    c.ntiles = tiles.size();
    HX_TRY(upload_vec(c.d_tiles, tiles));
    std::vector<double> cs(c.rows_pad, 0.0);
    for (int l = 0; l <= c.L; ++l) cs[l] = (2.0 * l + 1.0) / 2.0;
    HX_TRY(upload_vec(c.d_cs, cs));
    (void)st;
    return HX_OK;
}

static int mix_ctx_table(MixCtx &c, int t)
{
    if (c.have[t]) return HX_OK;
    hipStream_t st = rt().stream;
    // (the (0,0) table = P_l(x_k) also feeds the node weights xi(x_k) of every mask, a sum over l <= l3max: its rows run that far)
    const int ltop = t == 0 ? std::max(c.L, c.l3max) : c.L;
    const int rows = t == 0 ? (ltop + 1 + GB - 1) / GB * GB : c.rows_pad;
    if (t == 0) c.rows0_pad = rows;
    std::vector<WigCoefDD> coef;
    wigner_coefs_dd(ltop, kAB[t][0], kAB[t][1], coef);
    DevBuf d_coef;
    HX_TRY(upload_vec(d_coef, coef));
    HX_TRY(c.T[t].alloc(sizeof(double) * (size_t)rows * c.kpad));
    HX_HIP(hipMemsetAsync(c.T[t].p, 0, sizeof(double) * (size_t)rows * c.kpad, st));
    {
        ProfScope ps("wigner_tables");
        const long double s6 = sqrtl(6.0L) / 4.0L;
        const double s6h = (double)s6, s6l = (double)(s6 - (long double)s6h);
        hipLaunchKernelGGL(k_wigner_table_dd, dim3((c.n + 63) / 64), dim3(64), 0, st, ltop, kAB[t][0], kAB[t][1], c.n, c.gl.x.as<double>(),
                           c.gl.xlo.as<double>(), d_coef.as<WigCoefDD>(), c.T[t].as<double>(), (long long)c.kpad, 1LL, s6h, s6l);
    }
    HX_HIP(hipStreamSynchronize(st));  // d_coef dies with this scope
    c.have[t] = true;
    return HX_OK;
}

// node weights s_k = w_k xi(x_k) of one mask spectrum (device, l3max + 1 values): a matrix-vector product over the (0,0) table
static int mix_ctx_mask(MixCtx &c, const double *d_cl)
{
    hipStream_t st = rt().stream;
    HX_TRY(mix_ctx_table(c, 0));
    HX_TRY(c.xi_part.alloc(sizeof(double2) * (size_t)XI_SLICES * c.kpad));
    HX_HIP(hipMemsetAsync(c.s.p, 0, sizeof(double) * c.kpad, st));
    ProfScope ps("weight_xi");
    hipLaunchKernelGGL(k_xi_partial, dim3((c.kpad + 63) / 64, XI_SLICES), dim3(256), 0, st, c.l3max, c.kpad, d_cl, c.T[0].as<double>(), c.xi_part.as<double2>());
    hipLaunchKernelGGL(k_xi_finish, dim3((c.n + 255) / 256), dim3(256), 0, st, c.n, c.kpad, c.gl.w.as<double>(), c.xi_part.as<double2>(), c.s.as<double>());
    HX_HIP(hipGetLastError());
    return HX_OK;
}

// G^{(ab)} of the current mask into d_out (device, (l1max + 1) x (l2max + 1), ld = l2max + 1)
static int mix_ctx_product(MixCtx &c, int t, double *d_out)
{
    HX_TRY(mix_ctx_table(c, t));
    ProfScope ps("mixmat_gemm");
    const size_t nel = (size_t)c.rows_pad * c.kpad;
    if (!c.Ts.p) HX_TRY(c.Ts.alloc(sizeof(double) * nel));
    hipLaunchKernelGGL(k_scale_table, dim3(2048), dim3(256), 0, rt().stream, (long long)(nel / 2), c.kpad / 2, c.T[t].as<double2>(), c.s.as<double2>(),
                       c.Ts.as<double2>());
    ProfScope pk("mixmat_gemm_kernel");  // (the matrix kernel alone; "mixmat_gemm" includes the scaling pass)
    hipLaunchKernelGGL(k_mixmat_gemm_dma<true>, dim3((unsigned)c.ntiles), dim3(256), 0, rt().stream, c.Ts.as<double>(), c.T[t].as<double>(), c.kpad,
                       c.d_tiles.as<int2>(), c.l1max + 1, c.l2max + 1, c.d_cs.as<double>(), d_out, (long long)(c.l2max + 1));
    HX_HIP(hipGetLastError());
    return HX_OK;
}

static int stage_cl(const double *cl, int ncl, int l3max, DevBuf &buf)
{
    std::vector<double> h(l3max + 1, 0.0);
    HX_TRY(buf.alloc(sizeof(double) * (l3max + 1)));
    const int ncopy = std::min(ncl, l3max + 1);
    if (is_device_ptr(cl)) {
        HX_HIP(hipMemsetAsync(buf.p, 0, sizeof(double) * (l3max + 1), rt().stream));
        HX_HIP(hipMemcpyAsync(buf.p, cl, sizeof(double) * ncopy, hipMemcpyDeviceToDevice, rt().stream));
    } else {
        for (int i = 0; i < ncopy; ++i) h[i] = cl[i];
        HX_HIP(hipMemcpy(buf.p, h.data(), sizeof(double) * (l3max + 1), hipMemcpyHostToDevice));
    }
    return HX_OK;
}

// ---- the context and the staging buffer of the one-shot entry points, kept between calls ------------------------------------------
// hx_mixmat / hx_mixmat_eb / hx_mixmat_batch used to build and free a context per call: nodes, two or three 0.46 GB tables, the 0.46 GB
// scaled table and -- for a host destination -- a 0.9 GB staging buffer at L = 6144, i.e. ~3 GB of hipMalloc / hipFree per build.  Every
// second or third build of a row then took 0.12-0.18 s instead of 0.031 (BENCH_r04 seconds_all; round 5's first run: builds 2, 5 and 8 of a
// row, whatever the destination's kind, never with a device destination): the driver's unmapping of the freed blocks catching up.  The
// tables do not depend on the mask, so the last context is kept (one (l1max, l2max, l3max) at a time, as hx_mixctx_* keeps its own), and
// so is the staging buffer: a build allocates nothing.  hx_mixmat_release() frees both; hx_init on another device drops them.
struct MixCache {
    MixCtx *ctx = nullptr;
    DevBuf out_tmp, cl;
    int device = -1;
};
static MixCache &mix_cache()
{
    static MixCache *c = new MixCache;  // (never destroyed: its buffers must not be freed behind the HIP runtime's back at process exit)
    return *c;
}
static void mix_cache_drop()
{
    MixCache &mc = mix_cache();
    delete mc.ctx;
    mc.ctx = nullptr;
    mc.out_tmp.release();
    mc.cl.release();
}
static int mix_cached_ctx(int l1max, int l2max, int l3max, MixCtx **out)
{
    MixCache &mc = mix_cache();
    if (mc.device != rt().device) {
        mix_cache_drop();
        mc.device = rt().device;
    }
    if (!mc.ctx || mc.ctx->l1max != l1max || mc.ctx->l2max != l2max || mc.ctx->l3max != l3max) {
        if (mc.ctx) HX_HIP(hipStreamSynchronize(rt().stream));
        delete mc.ctx;
        mc.ctx = new MixCtx;
        const int rc = mix_ctx_init(*mc.ctx, l1max, l2max, l3max);
        if (rc != HX_OK) {
            delete mc.ctx;
            mc.ctx = nullptr;
            return rc;
        }
    }
    *out = mc.ctx;
    return HX_OK;
}
}  // namespace hx
void hx::mixmat_drop_cache() { hx::mix_cache_drop(); }
namespace hx {
// destination of a one-shot build: a device pointer as it is, a host pointer through the cached staging buffer
static int mix_bind_out(OutView &vo, double *out, size_t bytes)
{
    vo.bytes = bytes;
    if (is_device_ptr(out)) {
        vo.dev = out;
        vo.host = nullptr;
        return HX_OK;
    }
    HX_TRY(mix_cache().out_tmp.alloc(bytes));
    vo.dev = mix_cache().out_tmp.p;
    vo.host = out;
    return HX_OK;
}
// The three spin-2 x spin-2 matrices of one mask (context c, node weights set) into vo = [3][n1][n2].
// b = G^{(2,-2)} first: it IS the third matrix, so a host destination receives it (second stream, ~5 ms at L = 6144) while the
// product a = G^{(2,2)} is computed; then [0] = (a + b) / 2, [1] = (a - b) / 2.  Complete on return for a host destination.
static int mix_eb_into(MixCtx &c, OutView &vo)
{
    const size_t sz = (size_t)(c.l1max + 1) * (c.l2max + 1);
    double *o0 = vo.as<double>(), *o1 = o0 + sz, *o2 = o0 + 2 * sz;
    hipStream_t st = rt().stream;
    HX_TRY(mix_ctx_product(c, 3, o2));
    HX_TRY(mix_ctx_table(c, 2));
    hipStream_t cs = vo.host ? copy_stream() : nullptr;
    if (cs) {
        Runtime &r = rt();
        if (!r.order_ev) HX_HIP(hipEventCreateWithFlags(&r.order_ev, hipEventDisableTiming));
        HX_HIP(hipEventRecord(r.order_ev, st));
        HX_HIP(hipStreamWaitEvent(cs, r.order_ev, 0));
    }
    HX_TRY(mix_ctx_product(c, 2, o0));
    if (cs) HX_TRY(copy_d2h((double *)vo.host + 2 * sz, o2, sizeof(double) * sz, cs));  // (complete on return; the product above runs meanwhile)
    hipLaunchKernelGGL(k_eb_combine, dim3(1024), dim3(256), 0, st, (long long)sz, o0, o1, o2);
    HX_HIP(hipGetLastError());
    if (cs) {
        HX_TRY(copy_d2h(vo.host, o0, sizeof(double) * 2 * sz));
    } else {
        HX_TRY(vo.finish());
    }
    return HX_OK;
}

}  // namespace hx

using namespace hx;

extern "C" int hx_gauss_legendre(int n, double *x, double *w)
{
    HX_TRY(ensure_ready());
    if (n < 1 || !x || !w) return fail(HX_ERR_ARG, "hx_gauss_legendre: bad argument");
    GLCache c;
    HX_TRY(gl_nodes_device(n, c));
    OutView vx, vw;
    HX_TRY(vx.bind(x, sizeof(double) * n));
    HX_TRY(vw.bind(w, sizeof(double) * n));
    HX_HIP(hipMemcpyAsync(vx.dev, c.x.p, sizeof(double) * n, hipMemcpyDeviceToDevice, rt().stream));
    HX_HIP(hipMemcpyAsync(vw.dev, c.w.p, sizeof(double) * n, hipMemcpyDeviceToDevice, rt().stream));
    HX_TRY(vx.finish());
    HX_TRY(vw.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

// The same with the part of every node that a double cannot hold: node k = x[k] + xlo[k] (|xlo| <~ 1e-16; see k_gauss_legendre).
extern "C" int hx_gauss_legendre_dd(int n, double *x, double *w, double *xlo)
{
    HX_TRY(ensure_ready());
    if (n < 1 || !x || !w || !xlo) return fail(HX_ERR_ARG, "hx_gauss_legendre_dd: bad argument");
    GLCache c;
    HX_TRY(gl_nodes_device(n, c));
    OutView vx, vw, vl;
    HX_TRY(vx.bind(x, sizeof(double) * n));
    HX_TRY(vw.bind(w, sizeof(double) * n));
    HX_TRY(vl.bind(xlo, sizeof(double) * n));
    HX_HIP(hipMemcpyAsync(vx.dev, c.x.p, sizeof(double) * n, hipMemcpyDeviceToDevice, rt().stream));
    HX_HIP(hipMemcpyAsync(vw.dev, c.w.p, sizeof(double) * n, hipMemcpyDeviceToDevice, rt().stream));
    HX_HIP(hipMemcpyAsync(vl.dev, c.xlo.p, sizeof(double) * n, hipMemcpyDeviceToDevice, rt().stream));
    HX_TRY(vx.finish());
    HX_TRY(vw.finish());
    HX_TRY(vl.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

extern "C" int hx_wigner_d_table(int lmax, int a, int b, int n, const double *x, double *out)
{
    HX_TRY(ensure_ready());
    if (lmax < 0 || n < 1 || !x || !out) return fail(HX_ERR_ARG, "hx_wigner_d_table: bad argument");
    if (!((a == 0 && b == 0) || (a == 2 && b == 0) || (a == 2 && b == 2) || (a == 2 && b == -2) || (a == 1 && b == 1) || (a * b == -1)))
        return fail(HX_ERR_UNSUPPORTED, "hx_wigner_d_table: (a,b)=(%d,%d) not supported", a, b);
    InView vx;
    OutView vo;
    HX_TRY(vx.bind(x, sizeof(double) * n));
    HX_TRY(vo.bind(out, sizeof(double) * (size_t)n * (lmax + 1)));
    // the double-double kernel of the mixing-matrix tables at the caller's nodes (no low part: xlo = 0)
    std::vector<WigCoefDD> coef;
    wigner_coefs_dd(lmax, a, b, coef);
    DevBuf d_coef, d_zero;
    HX_TRY(upload_vec(d_coef, coef));
    HX_TRY(d_zero.alloc(sizeof(double) * n));
    HX_HIP(hipMemsetAsync(d_zero.p, 0, sizeof(double) * n, rt().stream));
    {
        ProfScope ps("wigner_tables");
        const long double s6 = sqrtl(6.0L) / 4.0L;
        const double s6h = (double)s6, s6l = (double)(s6 - (long double)s6h);
        hipLaunchKernelGGL(k_wigner_table_dd, dim3((n + 63) / 64), dim3(64), 0, rt().stream, lmax, a, b, n, vx.as<double>(), d_zero.as<double>(),
                           d_coef.as<WigCoefDD>(), vo.as<double>(), 1LL, (long long)(lmax + 1), s6h, s6l);
    }
    HX_HIP(hipGetLastError());
    HX_TRY(vo.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

static int mixmat_args(const double *cl, int ncl, int l1max, int l2max, int l3max, double *out)
{
    if (!cl || !out || ncl < 1 || l1max < 0 || l2max < 0 || l3max < 0)
        return fail(HX_ERR_ARG, "mixmat: bad argument");
    return HX_OK;
}

extern "C" int hx_mixmat(const double *cl, int ncl, int l1max, int l2max, int l3max, int s1, int s2, double *out)
{
    HX_TRY(ensure_ready());
    HX_TRY(mixmat_args(cl, ncl, l1max, l2max, l3max, out));
    int ab[1][2];
    if (s1 == 0 && s2 == 0) { ab[0][0] = 0; ab[0][1] = 0; }
    else if ((abs(s1) == 2 && s2 == 0) || (s1 == 0 && abs(s2) == 2)) { ab[0][0] = 2; ab[0][1] = 0; }
    else return fail(HX_ERR_UNSUPPORTED, "hx_mixmat: spin (%d,%d) not supported (use hx_mixmat_eb for (2,2))", s1, s2);
    MixCtx *c = nullptr;
    HX_TRY(mix_cached_ctx(l1max, l2max, l3max, &c));
    DevBuf &d_cl = mix_cache().cl;
    HX_TRY(stage_cl(cl, ncl, l3max, d_cl));
    OutView vo;
    HX_TRY(mix_bind_out(vo, out, sizeof(double) * (size_t)(l1max + 1) * (l2max + 1)));
    HX_TRY(mix_ctx_mask(*c, d_cl.as<double>()));
    HX_TRY(mix_ctx_product(*c, ab[0][0] == 0 ? 0 : 1, vo.as<double>()));
    HX_TRY(vo.finish());
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

extern "C" int hx_mixmat_eb(const double *cl, int ncl, int l1max, int l2max, int l3max, double *out)
{
    HX_TRY(ensure_ready());
    HX_TRY(mixmat_args(cl, ncl, l1max, l2max, l3max, out));
    const size_t sz = (size_t)(l1max + 1) * (l2max + 1);
    MixCtx *c = nullptr;
    HX_TRY(mix_cached_ctx(l1max, l2max, l3max, &c));
    DevBuf &d_cl = mix_cache().cl;
    HX_TRY(stage_cl(cl, ncl, l3max, d_cl));
    OutView vo;
    HX_TRY(mix_bind_out(vo, out, sizeof(double) * 3 * sz));
    HX_TRY(mix_ctx_mask(*c, d_cl.as<double>()));
    HX_TRY(mix_eb_into(*c, vo));
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

// Shader clock in GHz under k_mixmat_gemm_dma since the last call (sampled tiles: shader ticks / 100 MHz ticks); 0 if none ran.
extern "C" double hx_mixmat_gemm_clock(void)
{
    if (ensure_ready() != HX_OK) return 0.0;
    unsigned long long h[2] = {0, 0}, z[2] = {0, 0};
    if (hipStreamSynchronize(rt().stream) != hipSuccess || hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gemm_clk), sizeof(h)) != hipSuccess ||
        hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_clk), z, sizeof(z)) != hipSuccess) {
        (void)hipGetLastError();
        return 0.0;
    }
    return h[1] ? (double)h[0] / (double)h[1] * 0.1 : 0.0;
}

// Everything the library keeps in HBM between calls outside a plan or a context: the one-shot mixing-matrix cache above (tables of the
// last (l1max, l2max, l3max), the staging buffer of a host destination -- which hx_mixctx_apply shares --, the staged mask spectrum:
// ~3 GB at L = 6144) and the buffers of hx_alm2cl_pairs (tables, partial sums, staging: up to 512 MB).  The next call allocates again.
extern "C" int hx_release_caches(void)
{
    if (rt().ready) (void)hipStreamSynchronize(rt().stream);
    mix_cache_drop();
    alm2cl_drop_cache();
    return HX_OK;
}

// Frees what hx_mixmat / hx_mixmat_eb / hx_mixmat_batch keep between calls (the tables of the last (l1max, l2max, l3max) and the
// staging buffer of a host destination: ~3 GB at L = 6144).  The next build allocates them again.
extern "C" int hx_mixmat_release(void)
{
    if (rt().ready) (void)hipStreamSynchronize(rt().stream);
    mix_cache_drop();
    return HX_OK;
}

// y[v][i] = sum_j M[i][j] x[v][j]: the products of heracles.twopoint.apply_mixing_matrix (heracles/twopoint.py:497-524: `_M @ cl` per
// component spectrum).  One wave per row, the row read once for all nvec <= 4 vectors (HBM-bound: 8 B per matrix element).
__global__ __launch_bounds__(256) void k_matvec(int n, int m, long long ld, const double *__restrict__ M, int nvec, const double *__restrict__ x,
                                                double *__restrict__ y)
{
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= n) return;
    const double *r = M + (long long)row * ld;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int j = lane; j < m; j += 64) {
        const double a = r[j];
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (v < nvec) acc[v] = fma(a, x[(long long)v * m + j], acc[v]);
    }
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        double t = acc[v];
        for (int o = 32; o > 0; o >>= 1) t += __shfl_down(t, o, 64);  // fixed order: bitwise repeatable
        if (lane == 0 && v < nvec) y[(long long)v * n + row] = t;
    }
}

extern "C" int hx_matvec(int n, int m, const double *M, int nvec, const double *x, double *y)
{
    HX_TRY(ensure_ready());
    if (n < 0 || m < 0 || nvec < 0 || ((n > 0 && m > 0 && nvec > 0) && (!M || !x || !y))) return fail(HX_ERR_ARG, "hx_matvec: bad arguments");
    if (n == 0 || nvec == 0) return HX_OK;
    InView vm, vx;
    OutView vy;
    HX_TRY(vm.bind(M, sizeof(double) * (size_t)n * std::max(m, 1)));
    HX_TRY(vx.bind(x, sizeof(double) * (size_t)nvec * std::max(m, 1)));
    HX_TRY(vy.bind(y, sizeof(double) * (size_t)nvec * n));
    for (int v0 = 0; v0 < nvec; v0 += 4) {
        const int nv = std::min(4, nvec - v0);
        hipLaunchKernelGGL(k_matvec, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, rt().stream, n, m, (long long)m, vm.as<double>(), nv,
                           vx.as<double>() + (size_t)v0 * m, vy.as<double>() + (size_t)v0 * n);
        HX_HIP(hipGetLastError());
    }
    HX_TRY(vy.finish());
    return finish_call();
}

// Mixing matrices of nmask mask spectra for one (l1max, l2max, l3max): nodes, Wigner-d tables and tile list are built once,
// then every mask costs its node weights and one GEMM per product.  Replaces the serial loop of convolvecl calls in
// heracles.twopoint.mixing_matrices (heracles/twopoint.py:354-397).
//   cls   [nmask][ncl] (host or device), zero-padded / truncated to l3max + 1 like hx_mixmat
//   kinds [nmask] bit mask: 1 spin (0,0) -> out00[k], 2 spin (0,2) / (2,0) -> out02[k], 4 spin (2,2) -> outeb[k] (3 matrices)
//   out*  [nmask] pointers (host or device; entries of kinds not asked for are ignored and may be NULL)
extern "C" int hx_mixmat_batch(int nmask, const double *cls, int ncl, int l1max, int l2max, int l3max, const int *kinds,
                               double *const *out00, double *const *out02, double *const *outeb)
{
    HX_TRY(ensure_ready());
    if (nmask < 0 || (nmask > 0 && (!cls || !kinds)) || ncl < 1 || l1max < 0 || l2max < 0 || l3max < 0)
        return fail(HX_ERR_ARG, "hx_mixmat_batch: bad argument");
    if (nmask == 0) return HX_OK;
    for (int k = 0; k < nmask; ++k)
        if ((kinds[k] & ~7) || ((kinds[k] & 1) && (!out00 || !out00[k])) || ((kinds[k] & 2) && (!out02 || !out02[k])) ||
            ((kinds[k] & 4) && (!outeb || !outeb[k])))
            return fail(HX_ERR_ARG, "hx_mixmat_batch: mask %d: kinds=%d without an output buffer", k, kinds[k]);
    const size_t sz = (size_t)(l1max + 1) * (l2max + 1);
    MixCtx *cp = nullptr;
    HX_TRY(mix_cached_ctx(l1max, l2max, l3max, &cp));
    MixCtx &c = *cp;
    const bool cls_dev = is_device_ptr(cls);
    DevBuf &d_cl = mix_cache().cl;
    for (int k = 0; k < nmask; ++k) {
        if (!kinds[k]) continue;
        HX_TRY(stage_cl(cls + (size_t)k * ncl, ncl, l3max, d_cl));
        (void)cls_dev;
        HX_TRY(mix_ctx_mask(c, d_cl.as<double>()));
        if (kinds[k] & 1) {
            OutView vo;
            HX_TRY(mix_bind_out(vo, out00[k], sizeof(double) * sz));
            HX_TRY(mix_ctx_product(c, 0, vo.as<double>()));
            HX_TRY(vo.finish());
            HX_HIP(hipStreamSynchronize(rt().stream));
        }
        if (kinds[k] & 2) {
            OutView vo;
            HX_TRY(mix_bind_out(vo, out02[k], sizeof(double) * sz));
            HX_TRY(mix_ctx_product(c, 1, vo.as<double>()));
            HX_TRY(vo.finish());
            HX_HIP(hipStreamSynchronize(rt().stream));
        }
        if (kinds[k] & 4) {
            OutView vo;
            HX_TRY(mix_bind_out(vo, outeb[k], sizeof(double) * 3 * sz));
            HX_TRY(mix_eb_into(c, vo));
            HX_HIP(hipStreamSynchronize(rt().stream));
        }
    }
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

// ---- the same as an object: tables built once, masks streamed through (the outputs of a 13-bin job do not fit in memory
// at once; the reference's `out` mapping may write each matrix to disk as it arrives, heracles/twopoint.py:393-397) ----
struct hx_mixctx {
    hx::MixCtx c;
};

extern "C" hx_mixctx *hx_mixctx_create(int l1max, int l2max, int l3max)
{
    if (ensure_ready() != HX_OK) return nullptr;
    if (l1max < 0 || l2max < 0 || l3max < 0) {
        set_error("hx_mixctx_create: bad argument");
        return nullptr;
    }
    hx_mixctx *x = new hx_mixctx;
    if (mix_ctx_init(x->c, l1max, l2max, l3max) != HX_OK || hipStreamSynchronize(rt().stream) != hipSuccess) {
        delete x;
        return nullptr;
    }
    return x;
}

extern "C" void hx_mixctx_destroy(hx_mixctx *x)
{
    if (!x) return;
    if (rt().ready) (void)hipStreamSynchronize(rt().stream);
    delete x;
}

// kind 1: spin (0,0) -> out (l1max+1, l2max+1); 2: spin (0,2)/(2,0) -> the same shape; 4: spin (2,2) -> out (3, l1max+1, l2max+1)
extern "C" int hx_mixctx_apply(hx_mixctx *x, const double *cl, int ncl, int kind, double *out)
{
    HX_TRY(ensure_ready());
    if (!x || !cl || !out || ncl < 1 || (kind != 1 && kind != 2 && kind != 4)) return fail(HX_ERR_ARG, "hx_mixctx_apply: bad argument");
    MixCtx &c = x->c;
    const size_t sz = (size_t)(c.l1max + 1) * (c.l2max + 1);
    DevBuf d_cl;
    HX_TRY(stage_cl(cl, ncl, c.l3max, d_cl));
    HX_TRY(mix_ctx_mask(c, d_cl.as<double>()));
    OutView vo;
    HX_TRY(mix_bind_out(vo, out, sizeof(double) * sz * (kind == 4 ? 3 : 1)));  // (the staging buffer of a host destination is kept between calls)
    if (kind == 4) {
        HX_TRY(mix_eb_into(c, vo));
    } else {
        HX_TRY(mix_ctx_product(c, kind == 1 ? 0 : 1, vo.as<double>()));
        HX_TRY(vo.finish());
    }
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

// ---- binned rows: heracles.twopoint.mixing_matrices(..., bins, weights) (heracles/twopoint.py:391-397) ----------------------------
// which [l1max + 1]: bin of output multipole l (0 .. nbins - 1) or -1 (in no bin); w [l1max + 1]: its weight; norm [nbins]: the
// divisor of bin b (the reference: the summed weight).  Host arrays.  The binned tables are built on the first product that needs them.
extern "C" int hx_mixctx_set_bins(hx_mixctx *x, int nbins, const int *which, const double *w, const double *norm)
{
    HX_TRY(ensure_ready());
    if (!x || nbins < 1 || !which || !w || !norm) return fail(HX_ERR_ARG, "hx_mixctx_set_bins: bad argument");
    MixCtx &c = x->c;
    const int n1 = c.l1max + 1;
    for (int l = 0; l < n1; ++l)
        if (which[l] < -1 || which[l] >= nbins) return fail(HX_ERR_ARG, "hx_mixctx_set_bins: which[%d] = %d outside [-1, %d)", l, which[l], nbins);
    HX_HIP(hipStreamSynchronize(rt().stream));  // (a product of the previous bins may still be reading the lists)
    std::vector<int> start(nbins + 1, 0), rows;
    std::vector<double> rw;
    for (int b = 0; b < nbins; ++b) {
        for (int l = 0; l < n1; ++l)
            if (which[l] == b) {
                rows.push_back(l);
                rw.push_back(w[l]);
            }
        start[b + 1] = (int)rows.size();
    }
    c.nbins = nbins;
    c.nbpad = nbins <= 16 ? 16 : nbins <= 32 ? 32 : (nbins + 63) / 64 * 64;
    c.l2pad = (c.l2max + 1 + 63) / 64 * 64;  // (<= rows_pad: the tables are zero-padded to multiples of 128 rows)
    // enough work-groups for every CU to hold a few: the node range in shares of a multiple of 16
    const int mt = std::min(c.nbpad / 16, 4), groups = (c.l2pad / 64) * (c.nbpad / (16 * mt));
    int want = std::max(1, (4 * rt().cus + groups - 1) / groups);
    c.kchunk = std::max(64, ((c.kpad + want - 1) / want + 15) / 16 * 16);
    c.ksplit = (c.kpad + c.kchunk - 1) / c.kchunk;
    HX_TRY(upload_vec(c.b_start, start));
    HX_TRY(upload_vec(c.b_rows, rows));
    HX_TRY(upload_vec(c.b_w, rw));
    std::vector<double> nv(norm, norm + nbins);
    HX_TRY(upload_vec(c.b_norm, nv));
    for (bool &h : c.have_b) h = false;
    return HX_OK;
}

namespace hx {
static int mix_ctx_binned_table(MixCtx &c, int t)
{
    if (c.have_b[t]) return HX_OK;
    HX_TRY(mix_ctx_table(c, t));
    const size_t nel = (size_t)c.nbpad * c.kpad;
    HX_TRY(c.Tb[t].alloc(sizeof(double) * nel));
    HX_HIP(hipMemsetAsync(c.Tb[t].p, 0, sizeof(double) * nel, rt().stream));
    ProfScope ps("mixmat_bin_table");
    hipLaunchKernelGGL(k_bin_table, dim3((c.kpad + 255) / 256, c.nbins), dim3(256), 0, rt().stream, c.kpad, c.b_start.as<int>(), c.b_rows.as<int>(),
                       c.b_w.as<double>(), c.T[t].as<double>(), c.Tb[t].as<double>());
    HX_HIP(hipGetLastError());
    c.have_b[t] = true;
    return HX_OK;
}

// shares of the numerators of product t for the current mask into c.part[slot]
static int mix_ctx_binned_product(MixCtx &c, int t, int slot)
{
    HX_TRY(mix_ctx_binned_table(c, t));
    hipStream_t st = rt().stream;
    const size_t nel = (size_t)c.nbpad * c.kpad;
    HX_TRY(c.Tbs.alloc(sizeof(double) * nel));
    HX_TRY(c.part[slot].alloc(sizeof(double) * (size_t)c.ksplit * c.nbpad * c.l2pad));
    ProfScope ps("mixmat_binned");
    hipLaunchKernelGGL(k_scale_table, dim3(256), dim3(256), 0, st, (long long)(nel / 2), c.kpad / 2, c.Tb[t].as<double2>(), c.s.as<double2>(), c.Tbs.as<double2>());
    const int mt = std::min(c.nbpad / 16, 4);
    const dim3 grid(c.l2pad / 64, c.ksplit, c.nbpad / (16 * mt));
    if (mt == 1)
        hipLaunchKernelGGL(k_binned_gemm<1>, grid, dim3(256), 0, st, c.Tbs.as<double>(), c.T[t].as<double>(), c.kpad, c.kchunk, c.nbpad, c.l2pad, c.part[slot].as<double>());
    else if (mt == 2)
        hipLaunchKernelGGL(k_binned_gemm<2>, grid, dim3(256), 0, st, c.Tbs.as<double>(), c.T[t].as<double>(), c.kpad, c.kchunk, c.nbpad, c.l2pad, c.part[slot].as<double>());
    else
        hipLaunchKernelGGL(k_binned_gemm<4>, grid, dim3(256), 0, st, c.Tbs.as<double>(), c.T[t].as<double>(), c.kpad, c.kchunk, c.nbpad, c.l2pad, c.part[slot].as<double>());
    HX_HIP(hipGetLastError());
    return HX_OK;
}
}  // namespace hx

// kind as hx_mixctx_apply; out (nbins, l2max + 1) or (3, nbins, l2max + 1), host or device: the rows of the matrices binned as
// heracles.result.binned does along axis -2 (weighted mean per bin; exactly 0 where the weighted sum is exactly 0).
extern "C" int hx_mixctx_apply_binned(hx_mixctx *x, const double *cl, int ncl, int kind, double *out)
{
    HX_TRY(ensure_ready());
    if (!x || !cl || !out || ncl < 1 || (kind != 1 && kind != 2 && kind != 4)) return fail(HX_ERR_ARG, "hx_mixctx_apply_binned: bad argument");
    MixCtx &c = x->c;
    if (c.nbins < 1) return fail(HX_ERR_ARG, "hx_mixctx_apply_binned: no bins set (hx_mixctx_set_bins)");
    const int n2 = c.l2max + 1;
    const size_t sz = (size_t)c.nbins * n2;
    DevBuf &d_cl = mix_cache().cl;
    HX_TRY(stage_cl(cl, ncl, c.l3max, d_cl));
    HX_TRY(mix_ctx_mask(c, d_cl.as<double>()));
    const size_t bytes = sizeof(double) * sz * (kind == 4 ? 3 : 1);
    const bool to_host = !is_device_ptr(out);
    if (to_host) HX_TRY(c.b_out.alloc(bytes));
    double *d_out = to_host ? c.b_out.as<double>() : out;
    hipStream_t st = rt().stream;
    const dim3 grid((n2 + 255) / 256, c.nbins);
    if (kind == 4) {
        HX_TRY(mix_ctx_binned_product(c, 2, 0));
        HX_TRY(mix_ctx_binned_product(c, 3, 1));
        hipLaunchKernelGGL(k_binned_finish<true>, grid, dim3(256), 0, st, c.nbins, n2, c.nbpad, c.l2pad, c.ksplit, c.part[0].as<double>(), c.part[1].as<double>(),
                           c.d_cs.as<double>(), c.b_norm.as<double>(), d_out);
    } else {
        HX_TRY(mix_ctx_binned_product(c, kind == 1 ? 0 : 1, 0));
        hipLaunchKernelGGL(k_binned_finish<false>, grid, dim3(256), 0, st, c.nbins, n2, c.nbpad, c.l2pad, c.ksplit, c.part[0].as<double>(), (const double *)nullptr,
                           c.d_cs.as<double>(), c.b_norm.as<double>(), d_out);
    }
    HX_HIP(hipGetLastError());
    if (to_host) {  // (complete on return)
        if (is_pinned_host(out)) {
            HX_TRY(copy_d2h(out, d_out, bytes));
        } else {
            if (c.b_pin_bytes < bytes) {
                if (c.b_pin) (void)hipHostFree(c.b_pin);
                c.b_pin = nullptr;
                c.b_pin_bytes = 0;
                HX_HIP(hipHostMalloc(&c.b_pin, bytes, hipHostMallocDefault));
                c.b_pin_bytes = bytes;
            }
            HX_HIP(hipMemcpyAsync(c.b_pin, d_out, bytes, hipMemcpyDeviceToHost, st));
            HX_HIP(hipStreamSynchronize(st));
            memcpy(out, c.b_pin, bytes);
        }
    }
    HX_HIP(hipStreamSynchronize(rt().stream));
    return HX_OK;
}

// hx_cl2corr / hx_corr2cl live in hx_transforms.hip
