// hx_svd.hip -- truncated pseudo-inverse of a dense matrix: np.linalg.pinv(M, rcond) as heracles.twopoint.invert_mixing_matrix calls
// it on the mixing matrices (heracles/twopoint.py:447-460), i.e. pinv(M) = V diag(1 / sigma_j, sigma_j > rcond sigma_max) U^T.
//
// Blocked one-sided Jacobi (Hestenes) on the taller orientation W (n x m, n >= m) of the matrix:
//   columns in blocks of 32; a round-robin tournament pairs the blocks (nb / 2 independent pairs per step, nb - 1 steps per sweep);
//   per pair (I, J):  G = [W_I W_J]^T [W_I W_J]  (64 x 64, k_svd_gram: row chunks, partial sums added with f64 atomics)
//                     G = Q diag Q^T              (k_svd_eig: parallel-order cyclic Jacobi of the 64 x 64 matrix in LDS, one work-group per pair)
//                     [W_I W_J] <- [W_I W_J] Q,  [V_I V_J] <- [V_I V_J] Q   (k_svd_rotate, row chunks)
//   (one sweep of it per visit: see hx_pinv) until no pair had an off-diagonal Gram element above 1e-14 sqrt(G_ii G_jj).  Then the columns of W are u_j sigma_j, those of V are v_j:
//       pinv(M) = sum_j v_j w_j^T / sigma_j^2  (kept j)  =  V diag(mask / sigma^2) W^T     -- one GEMM on the matrix unit (launch_gemm_tst).
// One-sided Jacobi computes small singular values to high RELATIVE accuracy, which is what a relative cut-off (rcond) asks for.
// Everything runs on the GPU; the host drives the tournament and reads one convergence word per sweep and the m singular values once.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hx_common.h"

namespace hx {

constexpr int SB = 32;        // columns per block
constexpr int SP = 2 * SB;    // columns of a block pair
constexpr int SR = 64;        // rows per LDS tile of the Gram / rotation kernels
constexpr int SLD = SP + 2;   // LDS row stride of the 64 x 64 matrices of k_svd_eig (even: rows stay 16-byte aligned)

// W: [rows_pad][ld] row-major (ld = nb SB columns).  G[pair] += P^T P over the rows of this chunk, P = the 64 columns of the pair.
__global__ __launch_bounds__(256) void k_svd_gram(const double *__restrict__ W, long long ld, int nrows, const int2 *__restrict__ pairs,
                                                  int rows_per_chunk, double *__restrict__ G)
{
    __shared__ double Ps[SR][SP + 2];
    const int2 pr = pairs[blockIdx.x];
    const int t = threadIdx.x, ti = t >> 4, tj = t & 15;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, nrows);
    double acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
    for (int rb = r0; rb < r1; rb += SR) {
        // 64 rows x 64 columns: thread t loads row t >> 2, 16 columns from (t & 3) 16 (two column blocks of 32)
        {
            const int lr = t >> 2, c0 = (t & 3) * 16;
            const int row = rb + lr;
            const int gcol = (c0 < SB ? pr.x * SB + c0 : pr.y * SB + (c0 - SB));
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                double2 v = make_double2(0.0, 0.0);
                if (row < r1) v = *reinterpret_cast<const double2 *>(W + (long long)row * ld + gcol + c);
                Ps[lr][c0 + c] = v.x;
                Ps[lr][c0 + c + 1] = v.y;
            }
        }
        __syncthreads();
#pragma unroll 8
        for (int r = 0; r < SR; ++r) {
            double a[4], b[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { a[q] = Ps[r][4 * ti + q]; b[q] = Ps[r][4 * tj + q]; }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = fma(a[x], b[y], acc[x][y]);
        }
        __syncthreads();
    }
    double *g = G + (long long)blockIdx.x * SP * SP;
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
            __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)(g + (4 * ti + x) * SP + 4 * tj + y), acc[x][y]);
}

// Symmetric eigen-decomposition of every pair's 64 x 64 Gram matrix: G = Q diag Q^T, Q written over G's slot in Qout ([pair][64][64],
// Qout[c][j] = component c of eigenvector j).  Parallel-order cyclic Jacobi: 63 rounds of 32 disjoint rotations per sweep.  A round is
// A <- J^T A J with J = the 32 rotations together, applied 2 x 2 block by 2 x 2 block: block (a, b) = rows (p_a, q_a) x columns (p_b, q_b)
// becomes J_a^T [block] J_b, 1024 blocks over 256 threads, one pass and one barrier (separate column and row passes were two, with twice
// the LDS traffic); the eigenvectors are kept as ROWS (QT <- J^T QT), whose update runs over neighbouring columns in 128-bit accesses.
// (1.42 -> 1.2 ms per call of 8-9 sweeps at n = 4097: a round is latency, not traffic -- three dependent trips to LDS and the
// square roots of the rotation between two barriers, one wave per SIMD.)
// offmax (one double, as ordered integer bits): max over all pairs of |G_ij| / sqrt(G_ii G_jj) BEFORE the diagonalisation -- the
// convergence measure of the outer (one-sided) iteration.
// null2: squared norm below which a column counts as numerically zero for that measure (1e-30 of the matrix' squared Frobenius norm:
// sigma / sigma_max < 1e-15 sqrt(n)) -- the null columns of a rank-deficient matrix are rounding noise, any two of them "parallel" at
// O(1), and the measure never fell below the tolerance: such inputs ran all 60 sweeps (and, since round 5, would be refused)
__global__ __launch_bounds__(256) void k_svd_eig(const double *__restrict__ G, double *__restrict__ Qout, unsigned long long *__restrict__ offmax,
                                                 int max_sweeps, double stop_below, double null2)
{
    __shared__ __attribute__((aligned(16))) double A[SP][SLD], QT[SP][SLD];
    __shared__ double2 rot[SB];   // (c, s) of the round's rotations
    __shared__ int2 pq[SB];       // their (p, q), p < q
    __shared__ unsigned long long lmax_bits;
    const int t = threadIdx.x;
    const double *g = G + (long long)blockIdx.x * SP * SP;
    for (int e = t; e < SP * SP; e += 256) {
        A[e >> 6][e & 63] = g[e];
        QT[e >> 6][e & 63] = ((e >> 6) == (e & 63)) ? 1.0 : 0.0;
    }
    if (t == 0) lmax_bits = 0ull;
    __syncthreads();
    {   // off-diagonal measure of the incoming Gram matrix (zero columns -- padding -- are skipped)
        double mx = 0.0;
        for (int e = t; e < SP * SP; e += 256) {
            const int i = e >> 6, j = e & 63;
            if (i < j) {
                const double d = A[i][i] * A[j][j];
                if (d > 0.0 && A[i][i] > null2 && A[j][j] > null2) mx = fmax(mx, fabs(A[i][j]) / sqrt(d));
            }
        }
        atomicMax(&lmax_bits, (unsigned long long)__double_as_longlong(mx));
    }
    __syncthreads();
    if (t == 0) atomicMax(offmax, lmax_bits);
    const double first = __longlong_as_double((long long)lmax_bits);
    if (first > 1e-15) {
        const int a = t >> 3, b0 = t & 7;  // this thread's row pair; its column pairs are b0 + 8 u
        for (int sweep = 0; sweep < max_sweeps; ++sweep) {
            __syncthreads();
            if (t == 0) lmax_bits = 0ull;
            for (int round = 0; round < SP - 1; ++round) {
                __syncthreads();
                if (t < SB) {
                    // circle method: player 63 fixed, the others rotate
                    int p, q;
                    if (t == 0) { p = SP - 1; q = round; }
                    else { p = (round + t) % (SP - 1); q = (round - t + (SP - 1)) % (SP - 1); }
                    if (p > q) { const int z = p; p = q; q = z; }
                    const double app = A[p][p], aqq = A[q][q], apq = A[p][q];
                    double c = 1.0, s = 0.0;
                    const double dd = app * aqq;
                    const double rel = dd > 0.0 ? fabs(apq) / sqrt(dd) : (apq != 0.0 ? 1.0 : 0.0);
                    if (rel > 1e-17 && apq != 0.0) {
                        const double tau = (aqq - app) / (2.0 * apq);
                        const double tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                        c = 1.0 / sqrt(1.0 + tt * tt);
                        s = tt * c;
                    }
                    atomicMax(&lmax_bits, (unsigned long long)__double_as_longlong(rel));
                    pq[t] = make_int2(p, q);
                    rot[t] = make_double2(c, s);
                }
                __syncthreads();
                const int2 ra = pq[a];
                const double2 ja = rot[a];
                // eigenvector rows p_a, q_a: columns 2 (b0 + 8 u), + 1
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = 2 * (b0 + 8 * u);
                    double2 *xp = reinterpret_cast<double2 *>(&QT[ra.x][k]), *xq = reinterpret_cast<double2 *>(&QT[ra.y][k]);
                    const double2 vp = *xp, vq = *xq;
                    *xp = make_double2(ja.x * vp.x - ja.y * vq.x, ja.x * vp.y - ja.y * vq.y);
                    *xq = make_double2(ja.y * vp.x + ja.x * vq.x, ja.y * vp.y + ja.x * vq.y);
                }
                // blocks (a, b) of A
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int2 rb = pq[b0 + 8 * u];
                    const double2 jb = rot[b0 + 8 * u];
                    const double app = A[ra.x][rb.x], apq = A[ra.x][rb.y], aqp = A[ra.y][rb.x], aqq = A[ra.y][rb.y];
                    // rows: J_a^T
                    const double r0p = ja.x * app - ja.y * aqp, r0q = ja.x * apq - ja.y * aqq;
                    const double r1p = ja.y * app + ja.x * aqp, r1q = ja.y * apq + ja.x * aqq;
                    // columns: J_b
                    A[ra.x][rb.x] = jb.x * r0p - jb.y * r0q;
                    A[ra.x][rb.y] = jb.y * r0p + jb.x * r0q;
                    A[ra.y][rb.x] = jb.x * r1p - jb.y * r1q;
                    A[ra.y][rb.y] = jb.y * r1p + jb.x * r1q;
                }
            }
            __syncthreads();
            if (__longlong_as_double((long long)lmax_bits) < stop_below) break;
        }
    }
    __syncthreads();
    // eigenvectors in descending order of their eigenvalue = squared norm of the rotated column: the larger columns of a pair move to its
    // first block (de Rijk's ordering, block-wise).  Without it the one-sided iteration idled for ~16 sweeps at an off-diagonal measure of
    // 0.2-0.7 before its quadratic phase (21 sweeps at n = 2049)
    int *rank_of = reinterpret_cast<int *>(rot);  // (32 double2 = 128 ints, free after the last round)
    if (t < SP) {
        const double lam = A[t][t];
        int rank = 0;
        for (int i = 0; i < SP; ++i) {
            const double li = A[i][i];
            rank += (li > lam || (li == lam && i < t)) ? 1 : 0;
        }
        rank_of[t] = rank;
    }
    __syncthreads();
    double *qo = Qout + (long long)blockIdx.x * SP * SP;
    for (int e = t; e < SP * SP; e += 256) {
        const int j = e >> 6, c = e & 63;   // eigenvector j, component c
        qo[c * SP + rank_of[j]] = QT[j][c];
    }
}

// [X_I X_J] <- [X_I X_J] Q for the rows of this chunk (X = W, then X = V: blockIdx.z)
__global__ __launch_bounds__(256) void k_svd_rotate(double *__restrict__ W, int nrows_w, double *__restrict__ V, int nrows_v, long long ld,
                                                    const int2 *__restrict__ pairs, int rows_per_chunk, const double *__restrict__ Qall)
{
    __shared__ double Ps[SR][SP + 2], Qs[SP][SP + 2];
    double *X = blockIdx.z ? V : W;
    const int nrows = blockIdx.z ? nrows_v : nrows_w;
    const int2 pr = pairs[blockIdx.x];
    const int t = threadIdx.x, ti = t >> 4, tj = t & 15;
    const int r0 = blockIdx.y * rows_per_chunk, r1 = min(r0 + rows_per_chunk, nrows);
    if (r0 >= r1) return;
    const double *q = Qall + (long long)blockIdx.x * SP * SP;
    for (int e = t; e < SP * SP; e += 256) Qs[e >> 6][e & 63] = q[e];
    for (int rb = r0; rb < r1; rb += SR) {
        const int lr = t >> 2, c0 = (t & 3) * 16;
        const int row = rb + lr;
        const int gcol = (c0 < SB ? pr.x * SB + c0 : pr.y * SB + (c0 - SB));
        __syncthreads();
#pragma unroll
        for (int c = 0; c < 16; c += 2) {
            double2 v = make_double2(0.0, 0.0);
            if (row < r1) v = *reinterpret_cast<const double2 *>(X + (long long)row * ld + gcol + c);
            Ps[lr][c0 + c] = v.x;
            Ps[lr][c0 + c + 1] = v.y;
        }
        __syncthreads();
        // thread (ti, tj): rows 4 ti .. + 3, columns 4 tj .. + 3 of the product
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
#pragma unroll 8
        for (int c = 0; c < SP; ++c) {
            double a[4], b[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) { a[x] = Ps[4 * ti + x][c]; b[x] = Qs[c][4 * tj + x]; }
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y) acc[x][y] = fma(a[x], b[y], acc[x][y]);
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int orow = rb + 4 * ti + x;
            if (orow < r1) {
                const int oc = 4 * tj;
                const int ocol = (oc < SB ? pr.x * SB + oc : pr.y * SB + (oc - SB));
                *reinterpret_cast<double2 *>(X + (long long)orow * ld + ocol) = make_double2(acc[x][0], acc[x][1]);
                *reinterpret_cast<double2 *>(X + (long long)orow * ld + ocol + 2) = make_double2(acc[x][2], acc[x][3]);
            }
        }
    }
}

// out[c] = sum_r W[r][c]^2
__global__ __launch_bounds__(256) void k_svd_colnorm2(const double *__restrict__ W, long long ld, int nrows, int ncols, double *__restrict__ out)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= ncols) return;
    double s = 0.0;
    for (int r = 0; r < nrows; ++r) {
        const double v = W[(long long)r * ld + c];
        s = fma(v, v, s);
    }
    out[c] = s;
}

// dst [rows_pad][ld] (zero padded) <- src (n x m, row-major), transposed if tr
__global__ __launch_bounds__(256) void k_svd_load(const double *__restrict__ src, int n, int m, int tr, double *__restrict__ dst, long long ld, int rows)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * ld) return;
    const int r = (int)(i / ld), c = (int)(i % ld);
    double v = 0.0;
    if (!tr) { if (r < n && c < m) v = src[(long long)r * m + c]; }
    else { if (r < m && c < n) v = src[(long long)c * m + r]; }
    dst[i] = v;
}
__global__ __launch_bounds__(256) void k_svd_identity(double *__restrict__ V, long long ld, int rows, int m)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)rows * ld) return;
    const int r = (int)(i / ld), c = (int)(i % ld);
    V[i] = (r == c && r < m) ? 1.0 : 0.0;
}
// out (a x b) <- in (b x a) transposed
__global__ __launch_bounds__(256) void k_svd_transpose(const double *__restrict__ in, int a, int b, double *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a * b) return;
    const int r = (int)(i / b), c = (int)(i % b);
    out[i] = in[(long long)c * a + r];
}

}  // namespace hx

using namespace hx;

// out (m x n) = pinv(M) for M (n x m, row-major), singular values <= rcond * largest dropped (numpy.linalg.pinv).  M / out host or device.
// info (nullable, host): [0] sweeps of the one-sided iteration, [1] singular values kept, [2] largest, [3] smallest kept singular value.
extern "C" int hx_pinv(int n, int m, const double *M, double rcond, double *out, double *info)
{
    HX_TRY(ensure_ready());
    if (n < 1 || m < 1 || !M || !out || !(rcond >= 0.0)) return fail(HX_ERR_ARG, "hx_pinv: bad arguments");
    hipStream_t st = rt().stream;
    InView vin;
    OutView vout;
    HX_TRY(vin.bind(M, sizeof(double) * (size_t)n * m));
    HX_TRY(vout.bind(out, sizeof(double) * (size_t)n * m));
    // work on the taller orientation: W (nw x mw), nw >= mw
    const bool tr = n < m;
    const int nw = tr ? m : n, mw = tr ? n : m;
    int nb = (mw + SB - 1) / SB;
    if (nb & 1) ++nb;                       // the tournament wants an even number of blocks (an all-zero block is harmless)
    if (nb < 2) nb = 2;
    const long long ld = (long long)nb * SB;
    const int wrows = (nw + 127) / 128 * 128, vrows = (int)((ld + 127) / 128 * 128);
    DevBuf W, V, G, Q, d_pairs, d_off, d_s;
    HX_TRY(W.alloc(sizeof(double) * (size_t)wrows * ld));
    HX_TRY(V.alloc(sizeof(double) * (size_t)vrows * ld));
    const int npair = nb / 2, nsteps = nb - 1;
    HX_TRY(G.alloc(sizeof(double) * (size_t)npair * SP * SP));
    HX_TRY(Q.alloc(sizeof(double) * (size_t)npair * SP * SP));
    HX_TRY(d_off.alloc(sizeof(unsigned long long)));
    HX_TRY(d_s.alloc(sizeof(double) * ld));
    hipLaunchKernelGGL(k_svd_load, dim3((unsigned)(((long long)wrows * ld + 255) / 256)), dim3(256), 0, st, vin.as<double>(), n, m, tr ? 1 : 0, W.as<double>(), ld, wrows);
    hipLaunchKernelGGL(k_svd_identity, dim3((unsigned)(((long long)vrows * ld + 255) / 256)), dim3(256), 0, st, V.as<double>(), ld, vrows, mw);
    // round-robin tournament of the blocks (circle method)
    std::vector<int2> pairs((size_t)nsteps * npair);
    for (int r = 0; r < nsteps; ++r)
        for (int k = 0; k < npair; ++k) {
            int a, b;
            if (k == 0) { a = nb - 1; b = r; }
            else { a = (r + k) % (nb - 1); b = (r - k + (nb - 1)) % (nb - 1); }
            pairs[(size_t)r * npair + k] = make_int2(std::min(a, b), std::max(a, b));
        }
    HX_TRY(d_pairs.alloc(sizeof(int2) * pairs.size()));
    HX_HIP(hipMemcpyAsync(d_pairs.p, pairs.data(), sizeof(int2) * pairs.size(), hipMemcpyHostToDevice, st));
    const int rpc = 256;  // rows per chunk
    const int wchunks = (nw + rpc - 1) / rpc, vchunks = (int)((ld + rpc - 1) / rpc);
    // converged when no Gram element is above the rounding noise of its own sum (nw terms added in an unordered way)
    const double tol = std::max(1e-14, 16.0 * 1.1e-16 * std::sqrt((double)nw));
    // ONE Jacobi sweep per Gram matrix: diagonalising every 64 x 64 problem to rounding (8-9 sweeps on the first visits) bought nothing --
    // n = 4097: 18 outer sweeps / 3.49 s with up to 12 inner sweeps, 18 / 1.59 s with 2, 19 / 1.29 s with 1 (tools/pinv_grid.sh);
    // HX_SVD_INNER overrides (experiments)
    const char *e_in = getenv("HX_SVD_INNER");
    const int inner_sweeps = e_in ? std::max(1, atoi(e_in)) : 1;
    const double inner_stop = 1e-8;   // (a sweep that met nothing above 1e-8 leaves nothing above 1e-16: quadratic convergence)
    // squared Frobenius norm (invariant under the rotations): the scale of "numerically zero" columns in the convergence measure
    hipLaunchKernelGGL(k_svd_colnorm2, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, st, W.as<double>(), ld, nw, (int)ld, d_s.as<double>());
    double fro2 = 0.0;
    {
        std::vector<double> c2(ld);
        HX_HIP(hipMemcpyAsync(c2.data(), d_s.p, sizeof(double) * ld, hipMemcpyDeviceToHost, st));
        HX_HIP(hipStreamSynchronize(st));
        for (double v : c2) fro2 += v;
    }
    const double null2 = 1e-30 * fro2;
    int sweeps = 0;
    bool converged = false;
    double last_off = 0.0;
    for (; sweeps < 60; ++sweeps) {
        HX_HIP(hipMemsetAsync(d_off.p, 0, sizeof(unsigned long long), st));
        for (int r = 0; r < nsteps; ++r) {
            const int2 *pr = d_pairs.as<int2>() + (size_t)r * npair;
            HX_HIP(hipMemsetAsync(G.p, 0, sizeof(double) * (size_t)npair * SP * SP, st));
            hipLaunchKernelGGL(k_svd_gram, dim3(npair, wchunks), dim3(256), 0, st, W.as<double>(), ld, nw, pr, rpc, G.as<double>());
            hipLaunchKernelGGL(k_svd_eig, dim3(npair), dim3(256), 0, st, G.as<double>(), Q.as<double>(), d_off.as<unsigned long long>(), inner_sweeps, inner_stop, null2);
            hipLaunchKernelGGL(k_svd_rotate, dim3(npair, std::max(wchunks, vchunks), 2), dim3(256), 0, st, W.as<double>(), nw, V.as<double>(), (int)ld, ld, pr, rpc,
                               Q.as<double>());
        }
        HX_HIP(hipGetLastError());
        unsigned long long bits = 0;
        HX_HIP(hipMemcpyAsync(&bits, d_off.p, sizeof(bits), hipMemcpyDeviceToHost, st));
        HX_HIP(hipStreamSynchronize(st));
        double off;
        memcpy(&off, &bits, sizeof(off));
        if (getenv("HX_TRACE")) fprintf(stderr, "[hx] pinv: sweep %d, largest |G_ij| / sqrt(G_ii G_jj) = %.3e\n", sweeps, off);
        last_off = off;
        if (off < tol) { ++sweeps; converged = true; break; }
    }
    // (a pseudo-inverse from columns that are not orthogonal yet is silently wrong -- 60 sweeps are three times what the
    // hardest spectrum tried needs, tools/time_pinv.py; running out of them is an error, not a result)
    if (!converged)
        return fail(HX_ERR_UNSUPPORTED, "hx_pinv: the Jacobi sweeps did not converge (%d sweeps, largest |G_ij| / sqrt(G_ii G_jj) = %.3e, tolerance %.3e)", sweeps, last_off, tol);
    // singular values, cut-off, pinv = V diag(mask / sigma^2) W^T
    hipLaunchKernelGGL(k_svd_colnorm2, dim3((unsigned)((ld + 255) / 256)), dim3(256), 0, st, W.as<double>(), ld, nw, (int)ld, d_s.as<double>());
    std::vector<double> s2(ld);
    HX_HIP(hipMemcpyAsync(s2.data(), d_s.p, sizeof(double) * ld, hipMemcpyDeviceToHost, st));
    HX_HIP(hipStreamSynchronize(st));
    double smax = 0.0;
    for (double v : s2) smax = std::max(smax, std::sqrt(v));
    int kept = 0;
    double smin = smax;
    for (double &v : s2) {
        const double sg = std::sqrt(v);
        if (sg > rcond * smax && sg > 0.0) { ++kept; smin = std::min(smin, sg); v = 1.0 / v; }
        else v = 0.0;
    }
    HX_HIP(hipMemcpyAsync(d_s.p, s2.data(), sizeof(double) * ld, hipMemcpyHostToDevice, st));
    // pinv(Wm) (mw x nw) = V S W^T; the caller's pinv(M) is that (M tall) or its transpose (M wide: pinv(M) = pinv(M^T)^T)
    if (!tr) {
        HX_TRY(launch_gemm_tst(V.as<double>(), vrows, W.as<double>(), wrows, (int)ld, d_s.as<double>(), mw, nw, vout.as<double>(), nw));
    } else {
        DevBuf tmp;
        HX_TRY(tmp.alloc(sizeof(double) * (size_t)mw * nw));
        HX_TRY(launch_gemm_tst(V.as<double>(), vrows, W.as<double>(), wrows, (int)ld, d_s.as<double>(), mw, nw, tmp.as<double>(), nw));
        // tmp = pinv(M^T) (n x m) -> out (m x n)
        hipLaunchKernelGGL(k_svd_transpose, dim3((unsigned)(((long long)n * m + 255) / 256)), dim3(256), 0, st, tmp.as<double>(), m, n, vout.as<double>());
        HX_HIP(hipStreamSynchronize(st));
    }
    HX_HIP(hipGetLastError());
    HX_TRY(vout.finish());
    HX_HIP(hipStreamSynchronize(st));
    if (info) { info[0] = sweeps; info[1] = kept; info[2] = smax; info[3] = kept ? smin : 0.0; }
    return HX_OK;
}
