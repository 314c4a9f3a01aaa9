// hx_analysis.hip -- Legendre / Wigner-d analysis stage of hx_map2alm on FP64 MFMA.
//
// Replaces the Legendre half of healpy.map2alm (heracles/healpy.py:183-189):
//     a_lm = sum_rings lambda_lm(theta_r) F_m(r)               (spin 0)
//     E_lm, B_lm from  sum_rings  (+2)lambda, (-2)lambda ...    (spin 2)
//
// The sum over rings is the reduction the three-term recursion wants on the LANES, so it is
// given to v_mfma_f64_16x16x4_f64: per m,  a[l][col] = Lambda_m[l][ring] x F_m[ring][col]  is a
// GEMM whose A operand is generated on the fly and whose K dimension is the ring index.
//
//   workgroup  = one m  x  NW waves x 32 ring pairs
//   lane       = (h = lane>>5, ring = lane&31):   spin 0: h = parity chain of the two-step
//                recursion (l-m even / odd);  spin 2: h = which function (d_{m,-2} / d_{m,+2}).
//                Every lane runs ONE normalised recursion chain (2 FMAs per value).
//   tile       = 16 rows (l) x 64 lanes in LDS, XOR-swizzled so that the row-wise stores and
//                the [16 l x 4 rings] A-operand reads are both bank-conflict free.
//   MFMA q     contracts rings {2q, 2q+1, 2q+16, 2q+17} of the wave; its B operand (F of 8 maps =
//                16 real columns, x NG column groups) stays in registers for the whole l sweep.
//   flush      per 32-l block the waves' D tiles are summed through LDS in fixed order
//                (bit-reproducible) and scaled by alpha_l into `partial`.
// The kernel here is k_legendre_duo (batches of 5 or more spin-0 maps / 3 or more spin-2 fields: one work-group per m that walks its
// ring groups and adds them in place with f64 atomics in a fixed order -- `partial` holds ONE span of rows per m and k_alm_reduce only
// changes the layout).  Smaller batches run on the vector unit (hx_legendre_valu.hip: one span of rows per ring group, summed by
// k_alm_reduce).  Retired: round 1's barrier-phased kernel on the 4x4x4 instruction (round 3) and the one-group-per-CU software-
// pipelined k_legendre_pipe of rounds 2-3 (round 6; HISTORY.md section 4.1a has its timelines, tools/patches/r05_switches.patch its
// last source together with the diagnostic build switches of rounds 2-5).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "hx_sht_common.h"

namespace hx {
using namespace hxfft;

// HALF_F: spin-2 operand rows of the odd parity are not stored.  With x = B+(P+_N), y = B-(P-_S) (operand of lambda+; lambda-
// alike) the two rows are x + y and x - y, and x - y is x + y with the four columns of a field reversed and the signs (-, +, +, -)
// under lambda+, (+, -, -, +) under lambda-.  The Legendre kernel loads the even row a second time at column j ^ 3 (plain loads:
// anything else in its prologue -- a DPP move, a sign flip -- moves its 256 registers of operands around and costs the stages 140
// register copies each: 496 instead of 400 ms); the signs go where they are free: the lambda- chains run with alternating
// sign (-p', +q' in their coefficient table, the seed negated for m = 1), which leaves (-, +, +, -) on the odd-parity rows under
// BOTH functions, and k_alm_reduce<2> flips those when it changes the layout.  Half the F rows: half of what k_fourier_combine<2>
// writes and the Legendre prologue fetches from HBM.
constexpr int HALF_F = 1;
template <int SPIN>
struct LegCfg {
    static constexpr int NW = SPIN == 0 ? 16 : 8;   // waves per workgroup
    static constexpr int NT = SPIN == 0 ? 1 : 2;    // 16x64 tiles per wave
    static constexpr int NOP = SPIN == 0 ? 1 : 2;   // A-operand functions per ring
    static constexpr int NPAR = (SPIN == 2 && HALF_F) ? 1 : 2;  // parity rows of F per (m, ring pair)
};
int analysis_f_rows(int spin) { return spin == 0 ? LegCfg<0>::NPAR * LegCfg<0>::NOP : LegCfg<2>::NPAR * LegCfg<2>::NOP; }

struct LegParams {
    PlanDev P;
    const LegTask *__restrict__ tasks;
    const double *__restrict__ F;      // [m - m0][rp][par][op][16*NG]
    double *__restrict__ partial;      // [row - row0][16*NG]
    int m0, ms;                        // the chunk holds the orders m0 + k ms; F row block k
    long long row0;                    // first partial row of the chunk
    int ng;                            // active column groups (<= NG, + 1 if there are extra 4-column blocks)
    int ncol;                          // doubles per F / partial row: 16 per full group + 4 per extra block
    int pcol;                          // pipelined kernel: doubles per accumulation row = ncol rounded up to whole 128-byte lines
    // pipelined kernel only: a work-group owns one m and walks its ring groups in order (tasks and of_m are then the whole
    // lists, indexed by m), adding into ONE span of rows per m: arow[m] - arow0
    const MTasks *__restrict__ of_m;
    const long long *__restrict__ arow;
    long long arow0;
    int add_all;                       // k_legendre_duo: every ring group ADDS to its (zeroed) rows -- a launch that holds only some ring groups of an m (StreamSweep)
};

// =====================================================================================
// Y -> F operands:  un-pack N/S, ring phase, quadrature weight, parity combinations
// =====================================================================================
// grid: x = m - m0, y = tiles of 32 ring pairs; block 256 = 32 ring pairs x 8 slots.
// Component c lives in column group c/8, slot c%8 (spin 2: field f = c/2 in group f/4).
template <int SPIN>
__global__ __launch_bounds__(256) void k_fourier_combine(PlanDev P, const double2 *__restrict__ Y, int ncomp, int ng,
                                                         int ncol, int m0, int ms, const double *__restrict__ rw,
                                                         const LegTask *__restrict__ tasks, const MTasks *__restrict__ of_m,
                                                         double *__restrict__ F, int tile0)
{
    constexpr int NOP = LegCfg<SPIN>::NOP;
    __shared__ RingAtM ring_at_m[32];
    const int m = m0 + blockIdx.x * ms;
    const int ty = (int)blockIdx.y + tile0;  // tile of 32 ring pairs (tile0: first tile of the slab of a StreamSweep)
    // ring blocks in front of the first task of this m are pruned (m beyond what their rings resolve): their rows of F are
    // never read
    const MTasks mt = of_m[m];
    if (mt.count == 0 || ty < tasks[mt.first].rb0) return;
    if (threadIdx.x < 32) ring_at_m[threadIdx.x] = ring_at_m_of(P, ty * 32 + threadIdx.x, m, rw);
    __syncthreads();
    const int rp = ty * 32 + (threadIdx.x >> 3);
    const int slot = threadIdx.x & 7;
    const bool live = rp < P.nrp;
    const RingAtM ram = ring_at_m[threadIdx.x >> 3];
    constexpr int NPAR = LegCfg<SPIN>::NPAR;
    double *row = F + (((long long)blockIdx.x * P.nrp_pad + rp) * NPAR) * NOP * ncol;
    for (int g = 0; g < ng; ++g) {
        if (SPIN == 0) {
            const int c = g * 8 + slot;
            if (g * NCOL + 2 * slot >= ncol) continue;  // beyond the last (4-column) block of the row
            double2 s = make_double2(0.0, 0.0), d = s;
            if (live && c < ncomp) {
                double2 fn, fs;
                ring_modes_ns(P, Y, c, rp, m, ram, fn, fs);
                s = cadd(fn, fs);
                d = csub(fn, fs);
            }
            *reinterpret_cast<double2 *>(row + g * NCOL + 2 * slot) = s;
            *reinterpret_cast<double2 *>(row + ncol + g * NCOL + 2 * slot) = d;
        } else {
            const int f = g * 4 + (slot >> 1), op = slot & 1;
            if (g * NCOL + 4 * (slot >> 1) >= ncol) continue;  // beyond the last field of the row
            double4 o0 = make_double4(0.0, 0.0, 0.0, 0.0), o1 = o0;
            if (live && 2 * f + 1 < ncomp) {
                double2 qn, qs, un, us;
                ring_modes_ns(P, Y, 2 * f, rp, m, ram, qn, qs);
                ring_modes_ns(P, Y, 2 * f + 1, rp, m, ram, un, us);
                // P+ = -(Q + iU)/2, P- = -(Q - iU)/2
                const double2 ppn = cscale(cadd(qn, mul_pi(un)), -0.5), pmn = cscale(csub(qn, mul_pi(un)), -0.5);
                const double2 pps = cscale(cadd(qs, mul_pi(us)), -0.5), pms = cscale(csub(qs, mul_pi(us)), -0.5);
                // B+(P) = [Pr, Pi, Pi, -Pr]  (E_re, E_im, B_re, B_im columns; operand of lambda+)
                // B-(P) = [Pr, Pi, -Pi, Pr]  (operand of lambda-);  lambda+-_S = p lambda-+_N
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
            const int col = g * NCOL + 4 * (slot >> 1);
            *reinterpret_cast<double4 *>(row + (0 * NOP + op) * ncol + col) = o0;
            if (NPAR == 2) *reinterpret_cast<double4 *>(row + (1 * NOP + op) * ncol + col) = o1;
        }
    }
}

// =====================================================================================
// Legendre analysis
// =====================================================================================
// Tile addressing of the analysis kernel.  Element (row r, lane c) lives at column c ^ r
// (r < 16): row-wise stores stay contiguous per 16-lane group.  MFMA q contracts the rings
// ringsel(q, k), k = 0..3, chosen so that the [16 rows x 4 rings] operand read hits 32
// distinct 8-byte slots per 32-lane half (ds_read_b64) AND 16 distinct slots per 16-lane
// group (ds_read2_b64, which hipcc forms for the two parity halves).
__device__ __host__ inline int tile_swz(int r) { return r & 15; }
__device__ __host__ inline int ringsel(int q, int k) { return (k & 1) * 16 + 2 * q + (k >> 1); }

// Returns v through an empty asm statement that differs per K: the compiler cannot prove the 16
// swizzled tile-store addresses of an unrolled block loop-invariant and keeps ONE lane index in a
// register instead of 16 addresses (an xor + shift-add per store is free here; the registers are
// what the extra B operands of the spin-0 hybrid sweep need to stay out of scratch: 48 -> 16 spilled
// VGPRs).  Used for spin 0 only -- the spin-2 kernels have the registers and ran 8 % slower with it.
// Not volatile: it does not order anything.
template <int K>
__device__ inline int opaque(int v)
{
    asm("; opaque %1" : "+v"(v) : "n"(K));
    return v;
}

// Work the Legendre kernels EXECUTE, counted by the kernels themselves (one atomic per wave at its end; wave-uniform scalar
// counters): [0] FP64 flops of the matrix instructions actually issued (stages whose ring set is still dead (set_mode) skip theirs),
// [1] FP64 vector flops of the recursions (2 FMAs per generated value).  bench.py's roofline fraction is quoted on these; they agree
// with SQ_INSTS_VALU_MFMA_F64 of the PMC passes (profiles/).
__device__ unsigned long long g_exec_flops[2];

template <int SPIN>
struct PipeCfg {
    static constexpr int NW = 4;                              // waves per work-group
    static constexpr int RBS = SPIN == 0 ? 2 : 1;             // 32-ring-pair blocks per ring set (one set per wave)
    static constexpr int NCH = SPIN == 0 ? 2 : 1;             // recursion chains per lane
    static constexpr int NOP = SPIN == 0 ? 1 : 2;
};

// Tile of one ring set: element (lane-column c < 64, row r < 16, position p < 2) at double index
//     c * 32 + ((r ^ (c & 7)) * 2) + p .
// A lane-column is a recursion lane (spin 0: ring pair; spin 2: (function, ring pair)); row j / position p holds the value
// of l = lb + 2 j + p.  Every LDS access of the loop is 128 bits wide -- a wave that is alone on its SIMD pays ~30 cycles of
// issue for a 64-bit DS instruction and ~13 for a 128-bit one (measured: tools/stamp_pipe2.sh) --
//   store: one lane writes (p = 0, 1) of row j after two recursion steps: 8 lanes of a store group hit 8 different 16-byte
//          chunks of the 128-byte bank window because of the (c & 7) swizzle;
//   read:  lane (row i = lane & 15, k = lane >> 4) reads (p = 0, 1) of lane-column rho(q, k) = 4 q + k: the two lane-columns in
//          a 16-lane read group (k, k + 1) agree in bit 2, so their swizzled rows tile the 256-byte window.
__device__ __host__ inline int pipe_rho(int q, int k) { return 4 * q + k; }
__device__ __host__ inline int pipe_tile_idx(int c, int r) { return c * 32 + ((r ^ (c & 7)) * 2); }

// =====================================================================================
// Legendre analysis, TWO independent work-groups per CU ("duo", round 4)
// =====================================================================================
// The software-pipelined kernel of rounds 2-3 kept ONE work-group per CU: whatever one of its waves cannot overlap inside its own instruction stream -- the
// flush (D tiles through LDS, two barriers, atomics), waits on operands -- is idle time of that SIMD's FP64 pipe (busy 0.51-0.58 by
// counters).  Here a work-group is 4 waves of <= 256 registers with <= 80 KiB of LDS, so that TWO of them share a CU, each wave
// beside a wave of the OTHER group on its SIMD.  The two groups work on different orders m and never synchronise with each other.
// What the second wave can and cannot hide was measured first (tools/ubench_duo.hip, profiles/r04_ubench_duo.txt): a wave that streams
// FP64 matrix instructions back to back STARVES the FP64 vector instructions of the other wave of its SIMD completely (not one FMA
// gets through, whatever s_setprio says), so recursion and matrix work of the two waves can only alternate; LDS traffic, barrier
// waits and memory latency of one wave do sit under the other's matrix block.  The recipe is therefore: as much matrix work per
// recursion step and per flush as the registers allow, and nothing but matrix instructions + LDS reads inside a matrix block.
//   per 32-l block:  recursion of the wave's ring set (32 steps, values into its 16 KiB tile)  ->  16 slot pairs of matrix
//   instructions out of that tile  ->  flush: D tiles of the 4 waves through their own tiles (consumed by then), fixed order,
//   added in place into the rows of this m (one work-group per m, ring groups in order).
// One ring set per wave (spin 0: 64 ring pairs, both parity chains in a lane; spin 2: 32 ring pairs x the two functions): a task
// is 8 / 4 ring blocks.
// HALFB (spin 2, HALF_F): the operand of the odd-parity position is the operand of the even one with the four columns of every
// field reversed (see HALF_F): B_odd[k][j] = B_even[k][j ^ 3].  A matrix instruction that takes B_even for an odd-parity row block
// therefore yields that block's rows with the columns of every field reversed -- D_odd[i][j] = (A B_even)[i][j ^ 3] -- which
// k_alm_reduce undoes when it changes the layout.  ONE operand per (slot pair, column group) for both positions: 40 columns (ten
// fields) need 128 operand registers instead of 256 and fit a 256-register wave, with no instruction spent on the permutation.
// (Measured on the way, tools/ubench_mblock.hip: the permutation as two v_mov_b32_dpp per operand costs ~12 cycles of matrix-pipe
// time per VALU instruction -- 417 instead of 357 cycles per slot pair of 40 columns; as two ds_swizzle_b32 it is free.)
// Gaps in the matrix stream (round 4, second session; tools/ubench_duo.hip, profiles/r04_ubench_duo_gaps.txt).  A wave that streams FP64
// matrix instructions back to back starves the FP64 vector work of the other wave on its SIMD completely; with `s_nop 3` behind every
// (16 x 16 x 4, 4 x 4 x 4) pair and s_setprio 3 on the other wave, exactly two of its FMAs get through per gap, at 7.8 cycles of
// matrix-pipe time each (s_nop 7: 2.8 per gap at 11.8).  In the kernel the other group's recursion and reduction then advance under this
// group's matrix block (cycle accounting per wave-block, ten fields: recursion 4180 -> 3260, reduction 1965 -> 740, second barrier 547 ->
// 208) -- and the matrix block pays for every instruction let in (5390 -> 7850 with s_nop 7): 344 -> 337 ms for ten fields, a wash or a
// loss elsewhere (five fields 196 -> 203, ten spin-0 maps 100 -> 102); s_nop 3 is a gain of ~1 % everywhere.  DUO_GAP = n: s_nop n - 1
// behind every such pair (0: none; -1: 5 for the 40-column shape -- ten fields 344 ms without, 340 / 326 / 328 / 333 / 337 / 347 with n = 4 / 5 / 6 / 7 / 8 / 10 --, 6 for the 36-column one -- nine fields 304 ms without, 310 / 300.5 / 299.6 with n = 4 / 5 / 6 --, none for the 24-column one -- six fields 222 -> 225 ms with n = 3 ... 5 --, 4 otherwise (2 and 3 lose): five fields 196 -> 194, eight 275 -> 272, ten spin-0 maps 100.2 -> 99.2, sixteen 147.6 -> 139.7); DUO_PRIO: s_setprio 3 outside the matrix block.
constexpr int DUO_GAP = -1;
constexpr int DUO_PRIO = 1;
constexpr int DUO_ORDER = -1;  // -1: per shape (GORDER)
// Step of the scaled recursion chains of k_legendre_duo: value = v 2^(SB e), live (in the sums) from 2^-SB on.  Spin 2 runs with 100: the
// matrix work of 4 % of its blocks goes away (ten fields 321 -> 312 ms, results bit-identical: what is left out is below 2^-75 of a value
// of lambda).  Spin 0 stays at 300: the same rule saves 3.9 % of its matrix instructions and no time (97.3 -> 98.4 ms: two chains per lane
// to test in a kernel that already spills).
constexpr int DUO_SCALE_BITS2 = 100;
constexpr int DUO_SCALE_BITS0 = 300;

// lane K of every row of 16 lanes broadcast to its row: the only DPP control the FP64 ALU takes (one v_mov_b64_dpp)
template <int K>
__device__ __forceinline__ double row_bcast(double v)
{
    double d;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "n"(K));
    return d;
}

// t += p[lane K of the row] * x  (v_fmac_f64 with a row broadcast on its first source)
template <int K>
__device__ __forceinline__ double row_bcast_fmac(double t, double p, double x)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(p), "v"(x), "n"(K));
    return t;
}

// Rows of ONE order m (already summed over its ring groups: the pipelined kernels) -> alm layout, x alpha_l (k_legendre_duo leaves the
// output scaling of the normalised recursion to this pass: alphan != NULL) x fl.  r0 = the first row of the m in `partial`, ncol = doubles
// per row.  (Measured and not kept, round 5: the same pass at the end of the work-group of the m in k_legendre_duo instead of a launch of its
// own -- bit-identical, and the Legendre kernel grows by what the launch cost: ten fields 318.2 + 3.9 ms against 322.0 fused.)
template <int SPIN>
__device__ __forceinline__ void alm_rows_to_layout(const PlanDev &P, int m, bool any_task, const double *__restrict__ partial, long long r0, int ncol,
                                                   int ncomp, int ng, const double *__restrict__ fl, int add, double2 *__restrict__ alm,
                                                   long long alm_stride, const double *__restrict__ alphan, int tid, int nt)
{
    const int lmax = P.lmax;
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const int nl = lmax - m + 1, nc = 8 * ng;
    // U elements per thread and round, their loads issued together (one element per round: 4.9 instead of 3.9 ms per sweep of ten fields)
    constexpr int U = 8;
    for (int i0 = tid; i0 < nl * nc; i0 += nt * U) {
        double2 v[U];
        double al[U], f[U];
        long long dsti[U];
        bool on[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u * nt;
            const int l = m + i / nc, c = i % nc;
            on[u] = i < nl * nc && c < ncomp;
            v[u] = make_double2(0.0, 0.0);
            al[u] = 1.0;
            f[u] = 1.0;
            dsti[u] = 0;
            if (!on[u]) continue;
            dsti[u] = (long long)c * alm_stride + almidx(lmax, l, m);
            if (l >= l0 && any_task) {
                const double *prow = partial + (r0 + (l - l0)) * ncol + (c >> 3) * NCOL;
                v[u] = *reinterpret_cast<const double2 *>(prow + 2 * (c & 7));
                if (SPIN == 2 && alphan && ((l + m) & 1)) {
                    // k_legendre_duo: odd-parity rows were formed with the even-parity operand, i.e. with the four columns of every field reversed
                    const int j = 2 * (c & 7);
                    v[u] = make_double2(prow[j ^ 3], prow[(j + 1) ^ 3]);
                }
                if (SPIN == 2 && HALF_F && ((l + m) & 1)) {  // odd-parity rows carry the signs (-, +, +, -) on (E_re, E_im, B_re, B_im)
                    if (c & 1) v[u].y = -v[u].y;
                    else v[u].x = -v[u].x;
                }
                if (alphan) al[u] = alphan[almidx(lmax, l, m)];
                if (fl) f[u] = fl[l];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!on[u]) continue;
            double2 w = v[u];
            if (alphan) { w.x *= al[u]; w.y *= al[u]; }
            if (fl) { w.x *= f[u]; w.y *= f[u]; }
            double2 *dst = alm + dsti[u];
            if (add) { const double2 o = *dst; w.x += o.x; w.y += o.y; }
            *dst = w;
        }
    }
}

// NSUB: 32-l blocks per flush (their D tiles stay in registers; one staging round and one pair of barriers per NSUB blocks)
template <int SPIN, int NG, int NBX, int NSUB = 1>
__global__ __launch_bounds__(256, 2) void k_legendre_duo(LegParams A, const double2 *__restrict__ coefn)
{
    using C = PipeCfg<SPIN>;
    constexpr int NW = 4, NOP = C::NOP, NCH = C::NCH, NPAIR = 16;
    constexpr int NXA = NBX > 0 ? NBX : 1;
    constexpr int DQ0 = NG * 512, DSZ = NG * 512 + NBX * 128;
    constexpr bool HALFB = SPIN == 2;
    constexpr int SB = SPIN == 2 ? DUO_SCALE_BITS2 : DUO_SCALE_BITS0;  // scaled chains: value = v 2^(SB e), live from 2^-SB on (sval_rebase)
    constexpr int GAPN = DUO_GAP >= 0 ? DUO_GAP : ((SPIN == 2 && NG == 2 && NBX == 2) ? 5 : (SPIN == 2 && NG == 2 && NBX == 1) ? 6 : (SPIN == 2 && NG == 1 && NBX == 2) ? 0 : 4);
    // where the gaps stand: 0 behind every (16 x 16 x 4, 4 x 4 x 4) pair; 1 between the two instructions of a pair (ten spin-0 maps 99.4 ->
    // 98.1 ms; ten fields 344: not there); 2 behind the 16 x 16 x 4 and behind the 4 x 4 x 4 instructions of a position (ten fields 326.5 -> 324)
    constexpr int GORDER = DUO_ORDER >= 0 ? DUO_ORDER : ((SPIN == 2 && NG == 2 && NBX == 2) ? 2 : (SPIN == 0 && NG == 1 && NBX == 1) ? 1 : 0);
    constexpr int NPB = HALFB ? 1 : 2;  // operand positions kept in registers
    // doubles per wave of its tile: the 16 x 64 lambda tile (2048), or the D tiles of a flush if they need more (two blocks of 36 / 40
    // columns: 2304 / 2560 -- 80 KiB per work-group, still two per CU)
    constexpr int TW = NSUB * DSZ > 2048 ? NSUB * DSZ : 2048;
    static_assert(NG >= 1 && NG <= 2 && TW * NW * 8 <= 81920, "two work-groups per CU");
    static_assert(SPIN == 0 || HALF_F, "spin 2: the lambda- chain carries (-1)^(l + m) lambda- (one operand row, wave-uniform coefficients)");
    __shared__ double tile[NW][TW];            // 64 KiB (80 at most); doubles as the D staging area of the flush
    __shared__ int lead_in[2][NW];             // while no wave of the ring group has issued a matrix instruction yet: did wave w, in flush (slot) b?
    const PlanDev &P = A.P;
    const int m = A.m0 + blockIdx.x * A.ms, lmax = P.lmax;
    const MTasks mt = A.of_m[m];
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const int off = (l0 + m) & 1;
    const int nblk = (lmax - l0) / LBLK + 1;
    const long long orow = A.arow[m] - A.arow0;
    // recursion coefficients (p', q') of this m: WAVE-UNIFORM (spin 2: the lambda- chain, upper half of the wave, runs with (-p', +q'), i.e.
    // with the same coefficients and -x: HALF_F).  Lane j of every 16-lane row loads the pair of step j (one 16-byte load per 16 steps
    // and wave, requested a block ahead) and a step takes its pair from there with two row broadcasts (v_mov_b64_dpp row_newbcast) -- no
    // LDS hand-over, no LDS read per step (32 of the 48 LDS instructions of a block's recursion: 1760 -> cycles per block for a lone
    // wave), no scalar load (their latency cannot be covered: out-of-order return allows no load in flight across a wait: 5000 cycles)
    const double2 *__restrict__ cfm = coefn + almidx(lmax, 0, m) + l0 + (SPIN == 0 ? 0 : 1) + (threadIdx.x & 15);
    // the FIRST ring group of an m stores its rows, the others add to them: the rows need not be zeroed before the launch (3 GB per sweep of
    // ten fields) and a fortieth of the atomics is plain stores
    bool first_group = true;
    auto put = [&](double *p, double v) __attribute__((always_inline)) {
        if (first_group) *p = v;
        else __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)p, v);
    };
    auto lds_barrier = []() __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): global atomics in flight do not hold the barrier
        __builtin_amdgcn_s_barrier();
    };
    int n_mf = 0, n_rec = 0;
    if (DUO_PRIO) __builtin_amdgcn_s_setprio(3);
    for (int ti = 0; ti < mt.count; ++ti) {
        const LegTask task = A.tasks[mt.first + ti];
        first_group = ti == 0 && !A.add_all;
        int tid = threadIdx.x;
        asm volatile("; ring group" : "+v"(tid));
        const int w = tid >> 6, lane = tid & 63;
        const int ai = lane & 15, ak = lane >> 4;
        if (ti) __syncthreads();  // the previous ring group's last readers of the tiles

        // ring of this lane: wave w holds ring block w (spin 2) / blocks 2 w, 2 w + 1 (spin 0) of the task
        const int rbi = SPIN == 0 ? 2 * w + (lane >> 5) : w;
        const int rpl = (task.rb0 + rbi) * RBLK + (lane & 31);
        const bool valid = rbi < task.nrb && rpl < P.nrp;
        const double x = valid ? P.z[rpl] : 0.0;
        const double xx = SPIN == 0 ? x * x : ((lane >> 5) ? -x : x);

        // B operands: lane (k = lane >> 4, j = lane & 15) holds F[ring k of the slot][parity of the position][op][column]
        // (HALFB: position 0 only -- position 1 is its quad-reversed image)
        double fr[NPAIR][NPB][NG], frx[NPAIR][NPB][NXA];
#pragma unroll
        for (int sp = 0; sp < NPAIR; ++sp) {
            const int op = SPIN == 0 ? 0 : sp & 1, q = SPIN == 0 ? sp : sp >> 1;
            const int rbq = SPIN == 0 ? 2 * w + (q >> 3) : w;
            const bool on = rbq < task.nrb;
            const long long row = (long long)blockIdx.x * P.nrp_pad + (task.rb0 + rbq) * RBLK + pipe_rho(q & 7, ak);
            if (HALFB) {
                const double *f = A.F + (row * NOP + op) * A.ncol;  // the even-parity row, for both positions
#pragma unroll
                for (int g = 0; g < NG; ++g) fr[sp][0][g] = on ? f[g * NCOL + ai] : 0.0;
#pragma unroll
                for (int g = 0; g < NXA; ++g) frx[sp][0][g] = (NBX > 0 && on) ? f[NG * NCOL + 4 * g + (lane & 3)] : 0.0;
            } else {
#pragma unroll
                for (int pos = 0; pos < NPB; ++pos) {
                    const double *f = A.F + ((row * 2 + (pos ^ off)) * NOP + op) * A.ncol;
#pragma unroll
                    for (int g = 0; g < NG; ++g) fr[sp][pos][g] = on ? f[g * NCOL + ai] : 0.0;
#pragma unroll
                    for (int g = 0; g < NXA; ++g) frx[sp][pos][g] = (NBX > 0 && on) ? f[NG * NCOL + 4 * g + (lane & 3)] : 0.0;
                }
            }
        }

        // seeds
        double vc[NCH], vp[NCH];
        int sc[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) { vc[c] = 0.0; vp[c] = 0.0; sc[c] = -100; }
        if (valid) {
            if (SPIN == 0) {
                SVal a = spow(P.sth[rpl], m);
                a.v *= P.mfac[m];
                SVal b = a;
                b.v *= sqrt(2.0 * m + 3.0) * P.z[rpl];
                snorm_small(a);
                snorm_small(b);
                vc[0] = a.v; sc[0] = a.e;
                vc[NCH - 1] = b.v; sc[NCH - 1] = b.e;
            } else {
                SVal sp, sm;
                spin2_seeds(m, P.sth[rpl], P.omz[rpl], P.kfac2[m], sp, sm);
                vc[0] = (lane >> 5) ? (off ? -sm.v : sm.v) : sp.v;  // the lambda- chain alternates in sign, + at even l + m
                sc[0] = (lane >> 5) ? sm.e : sp.e;
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) sval_rebase<SB>(vc[c], sc[c]);
        }

        auto rec_step = [&](auto RMM, int c, int step, const double tq) __attribute__((always_inline)) {
            constexpr int RM = decltype(RMM)::value;
            if (RM != 3 && (step & 3) == 0) {
                const int hc = __double2hiint(vc[c]), hp = __double2hiint(vp[c]);
                const bool up = sc[c] < 0 && (hc & 0x7ff00000) >= 0x3ff00000;
                const int sub = up ? (SB << 20) : 0;
                const bool pz = up && (hp & 0x7ff00000) <= (SB << 20);
                vc[c] = __hiloint2double(hc - sub, __double2loint(vc[c]));
                vp[c] = pz ? 0.0 : __hiloint2double(hp - sub, __double2loint(vp[c]));
                sc[c] += up ? 1 : 0;
            }
            const double cur = (RM == 3 || sc[c] == 0) ? vc[c] : 0.0;
            const double vn = fma(tq, vc[c], -vp[c]);
            vp[c] = vc[c];
            vc[c] = vn;
            return cur;
        };
        constexpr int HB = 8;
        double *tw = &tile[w][0];
        // the 32 steps of a block; cl[i] = this lane's coefficient pair of steps 16 i + (lane & 15)
        auto recursion = [&](auto RMM, const double2 (&cl)[2]) __attribute__((always_inline)) {
            constexpr int RM = decltype(RMM)::value;
#pragma unroll
            for (int h = 0; h < LBLK / HB; ++h) {
                double cur[HB];
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    constexpr int dummy = 0; (void)dummy;
                    const int kk = HB * h + k;
                    double tq = 0.0;  // p' x + q' of this step: q' by a row broadcast, p' as the broadcast source of the multiply-add
                    // (the lane index of the broadcast is an immediate: the loop is fully unrolled)
                    switch (kk & 15) {
#define HX_BC(K) case K: tq = row_bcast_fmac<K>(row_bcast<K>(cl[kk >> 4].y), cl[kk >> 4].x, xx); break;
                        HX_BC(0) HX_BC(1) HX_BC(2) HX_BC(3) HX_BC(4) HX_BC(5) HX_BC(6) HX_BC(7)
                        HX_BC(8) HX_BC(9) HX_BC(10) HX_BC(11) HX_BC(12) HX_BC(13) HX_BC(14) HX_BC(15)
#undef HX_BC
                    }
                    cur[k] = rec_step(RMM, SPIN == 0 ? ((kk & 1) ? NCH - 1 : 0) : 0, SPIN == 0 ? kk >> 1 : kk, tq);
                }
                if (RM >= 2) {
#pragma unroll
                    for (int j = 0; j < HB / 2; ++j)
                        *reinterpret_cast<double2 *>(tw + pipe_tile_idx(lane, h * (HB / 2) + j)) = make_double2(cur[2 * j], cur[2 * j + 1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // Mode of block b for this wave: 1 = every chain stays out of the sums for the whole block (no tile, no matrix instructions), 3 = every
        // chain is live, 2 = mixed.  A chain is live from 2^-SB on; with SB < 300 "dead at the entry" is not enough for mode 1: near l = m a
        // step multiplies by up to sqrt(2m / (l - m)), and a chain that enters the first block at 2^-100 can leave it at 1e-10.  So a lane
        // counts as dead for the block only below 2^-(SB + E_b) at the entry, E_b = 60, 34, 26, 24, ... 8 for b = 0, 1, 2, 3 ... >= 11:
        // calibrated (tools/calibrate_dead_blocks.py: long-double emulation of both recursions on the plan's own rings) so that nothing
        // above 2^-75 of a value of lambda is left out.  The margin NEEDED saturates with m and lmax -- block 0 / 1 / 2: 51.8 / 25.9 / 17.6
        // bits at lmax 6144 (m 5944), 52.0 / 26.0 / 17.9 at lmax 8000 (m 7800), 52.2 / 26.3 / 18.1 at lmax 12288 (m 12088), nside 8192:
        // the chains that matter are those of the rings next to the pruning limit, whose growth is set by m / (l sin theta), not by m
        // (profiles/r06_dead_block_calibration.txt) -- so these constants hold for every plan the library accepts, with >= 7.7 bits
        // to spare (tests/test_host_logic.py holds them to >= 6 at lmax 6144, 8000 and 12288).  The matrix work of 4 % of the blocks
        // goes away against SB = 300.
        auto set_mode = [&](int b) __attribute__((always_inline)) {
            const int eb = b == 0 ? 60 : (b == 1 ? 34 : (30 - 2 * b > 8 ? 30 - 2 * b : 8));
            auto lane_dead = [&](int c) __attribute__((always_inline)) {
                if (SB == 300) return sc[c] < 0;
                const int ef = (__double2hiint(vc[c]) >> 20) & 0x7ff;  // |v| < 2^(ef - 1022)
                return sc[c] <= -2 || (sc[c] == -1 && ef <= 1022 - eb);
            };
            bool dead = !valid || lane_dead(0), live = !valid || sc[0] == 0;
            if (NCH == 2) {
                dead = dead && (!valid || lane_dead(NCH - 1));
                live = live && (!valid || sc[NCH - 1] == 0);
            }
            return __all(dead) ? 1 : (__all(live) ? 3 : 2);
        };
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;

        // flush roles: 16-byte chunk (col, c) of a group tile = rows (c >> 1) + 8 (c & 1) and + 4
        const int fcol = tid & 15, fch = (tid >> 4) & 7, fpos = (tid >> 7) & 1;
        const int frow = (fch >> 1) + 8 * (fch & 1);
        const int qrow = tid / (4 * NXA), qcol = tid % (4 * NXA);
        double *pgrp = A.partial + (orow + 2 * frow + fpos) * A.pcol + fcol;
        double *pquad = A.partial + (orow + 2 * qrow) * A.pcol + NG * NCOL + qcol;
        double2 cnext[2] = {cfm[0], cfm[16]};
        // Lead-in of a ring group: the blocks in which every ring of all four waves is still dead (below 2^-300: ~ 8 % of the blocks; below 2^-100, spin 2: ~ 12 %) have
        // nothing to flush.  Until the first wave issues matrix instructions the waves exchange one flag per flush through LDS (one barrier,
        // one 16-byte read) and skip staging, reduction, atomics and the second barrier; from then on no flag is read or written any more
        // (read in EVERY flush the flags cost more than they saved: 357 vs 345 ms).  The first ring group of an m writes all its rows.
        bool started = first_group;
        int nflush = 0;
        for (int b0 = 0; b0 < nblk; b0 += NSUB) {
            double4_t acc[NSUB][NG][2];
            double accx[NSUB][NXA][2];
            bool mine = false;
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                const int b = b0 + sub;
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[sub][g][0] = acc[sub][g][1] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int g = 0; g < NXA; ++g) accx[sub][g][0] = accx[sub][g][1] = 0.0;
                if (b >= nblk) break;  // (rows are padded to whole blocks, the row span of an m to nblk blocks: nothing is stored beyond it)
                int rm = set_mode(b);
                const double2 cl[2] = {cnext[0], cnext[1]};
                if (rm == 3) recursion(I3{}, cl);
                else if (rm == 2) {
                    recursion(I2{}, cl);
                    if (SB != 300) {  // a mixed block in which no chain came to life stored zeros only: nothing to multiply, nothing to flush
                        bool lv = valid && sc[0] == 0;
                        if (NCH == 2) lv = lv || (valid && sc[NCH - 1] == 0);
                        if (!__any(lv)) rm = 1;
                    }
                } else recursion(I1{}, cl);
                // coefficients of the next block: requested in front of this block's matrix work and of its flush (vmcnt retires in order: a load
                // behind the atomics of the flush could not be waited for without waiting for them)
                cnext[0] = cfm[(b + 1) * LBLK];
                cnext[1] = cfm[(b + 1) * LBLK + 16];
                n_rec = __builtin_amdgcn_readfirstlane(n_rec + 1);
                mine = mine || rm >= 2;
                if (rm >= 2) {
                    n_mf = __builtin_amdgcn_readfirstlane(n_mf + 1);
                    if (DUO_PRIO) __builtin_amdgcn_s_setprio(0);
                    constexpr int PF = 3;
                    auto a_fetch = [&](int sp) __attribute__((always_inline)) {
                        const int op = SPIN == 0 ? 0 : sp & 1, q = SPIN == 0 ? sp : sp >> 1;
                        const int c = SPIN == 0 ? (q >> 3) * 32 + pipe_rho(q & 7, ak) : op * 32 + pipe_rho(q, ak);
                        return *reinterpret_cast<const double2 *>(tw + pipe_tile_idx(c, ai));
                    };
                    double2 aq[PF + 1];
#pragma unroll
                    for (int j = 0; j < PF; ++j) aq[j] = a_fetch(j);
#pragma unroll
                    for (int sp = 0; sp < NPAIR; ++sp) {
                        if (sp + PF < NPAIR) aq[(sp + PF) % (PF + 1)] = a_fetch(sp + PF);
                        const double a0 = aq[sp % (PF + 1)].x, a1 = aq[sp % (PF + 1)].y;
                        // (HALFB: position 1 takes the operand of position 0; its rows come out with the columns of every field reversed)
#pragma unroll
                        for (int pos = 0; pos < 2; ++pos) {
                            const double a = pos ? a1 : a0;
                            if (GAPN > 0) {
                                // one 16 x 16 x 4 (+ one 4 x 4 x 4) instruction, then a gap of a few cycles: see DUO_GAP
                                if (GORDER == 1) {  // the gap between the 16 x 16 x 4 and the 4 x 4 x 4 instruction of a pair
#pragma unroll
                                for (int g = 0; g < (NG > NBX ? NG : NBX); ++g) {
                                    if (g < NG) acc[sub][g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[sp][HALFB ? 0 : pos][g], acc[sub][g][pos], 0, 0, 0);
                                    __builtin_amdgcn_sched_barrier(0);
                                    asm volatile("s_nop %0" ::"n"(GAPN > 0 ? GAPN - 1 : 0));
                                    __builtin_amdgcn_sched_barrier(0);
                                    if (g < NBX) accx[sub][g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, frx[sp][HALFB ? 0 : pos][g], accx[sub][g][pos], 0, 0, 0);
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                                } else if (GORDER == 2) {  // all 16 x 16 x 4 of the position, gap, all 4 x 4 x 4, gap
#pragma unroll
                                for (int g = 0; g < NG; ++g) acc[sub][g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[sp][HALFB ? 0 : pos][g], acc[sub][g][pos], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                                asm volatile("s_nop %0" ::"n"(GAPN > 0 ? GAPN - 1 : 0));
                                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                                for (int g = 0; g < NBX; ++g) accx[sub][g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, frx[sp][HALFB ? 0 : pos][g], accx[sub][g][pos], 0, 0, 0);
                                __builtin_amdgcn_sched_barrier(0);
                                if (NBX > 0) asm volatile("s_nop %0" ::"n"(GAPN > 0 ? GAPN - 1 : 0));
                                __builtin_amdgcn_sched_barrier(0);
                                } else {
#pragma unroll
                                for (int g = 0; g < (NG > NBX ? NG : NBX); ++g) {
                                    if (g < NG) acc[sub][g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[sp][HALFB ? 0 : pos][g], acc[sub][g][pos], 0, 0, 0);
                                    if (g < NBX) accx[sub][g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, frx[sp][HALFB ? 0 : pos][g], accx[sub][g][pos], 0, 0, 0);
                                    __builtin_amdgcn_sched_barrier(0);
                                    asm volatile("s_nop %0" ::"n"(GAPN > 0 ? GAPN - 1 : 0));
                                    __builtin_amdgcn_sched_barrier(0);
                                }
                                }
                            } else {
#pragma unroll
                                for (int g = 0; g < NG; ++g) acc[sub][g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[sp][HALFB ? 0 : pos][g], acc[sub][g][pos], 0, 0, 0);
#pragma unroll
                                for (int g = 0; g < NBX; ++g) accx[sub][g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, frx[sp][HALFB ? 0 : pos][g], accx[sub][g][pos], 0, 0, 0);
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (DUO_PRIO) __builtin_amdgcn_s_setprio(3);
                }
            }
            // ---- flush: D tiles of the 4 waves through their own tiles (consumed above), fixed order; the output scaling alpha_l is
            // applied by k_alm_reduce (the same factor for every ring group of the m).  (Measured and not kept: per-wave "issued matrix
            // instructions" flags in LDS so that dead waves stage nothing and all-dead blocks skip the flush -- the flag read behind the
            // barrier costs every flush more than the dead blocks save: 357 vs 345 ms for ten fields, same device.) ----
            // staging of (group g, position p): column-major, 16-byte chunk c = 2 (lane >> 4) + (reg >> 1) of column col at
            // (g 2 + p) 256 + col 16 + (c ^ (col & 7)) 2;  4-column blocks: lane (i = lane >> 4, blk = (lane >> 2) & 3, j = lane & 3) = row 4 blk + i,
            // column j, both positions in one 16-byte store at DQ0 + ((row 4 NBX + 4 x + j) 2); sub-block sub at + sub DSZ
            const bool two = NSUB > 1 && b0 + 1 < nblk;
            if (!started) {
                if (lane == 0) lead_in[nflush & 1][w] = mine ? 1 : 0;
                lds_barrier();
                const int4 fl = *reinterpret_cast<const int4 *>(&lead_in[nflush & 1][0]);
                ++nflush;
                started = __builtin_amdgcn_readfirstlane(fl.x | fl.y | fl.z | fl.w) != 0;
                if (!started) {  // nothing to add to the rows of these blocks
                    pgrp += (long long)NSUB * LBLK * A.pcol;
                    pquad += (long long)NSUB * LBLK * A.pcol;
                    continue;
                }
            }
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                double *dt = tw + sub * DSZ;
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int pos = 0; pos < 2; ++pos)
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            *reinterpret_cast<double2 *>(dt + (g * 2 + pos) * 256 + ai * 16 + (((2 * ak + h) ^ (ai & 7)) * 2)) = make_double2(acc[sub][g][pos][2 * h], acc[sub][g][pos][2 * h + 1]);
#pragma unroll
                for (int g = 0; g < NBX; ++g)
                    *reinterpret_cast<double2 *>(dt + DQ0 + ((4 * ((lane >> 2) & 3) + ak) * 4 * NBX + 4 * g + (lane & 3)) * 2) = make_double2(accx[sub][g][0], accx[sub][g][1]);
            }
            lds_barrier();
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
                if (sub > 0 && !two) break;
                const long long rsub = (long long)sub * LBLK * A.pcol;
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    const double *src = &tile[0][0] + sub * DSZ + (g * 2 + fpos) * 256 + fcol * 16 + ((fch ^ (fcol & 7)) * 2);
                    double2 s4[NW];
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) s4[ww] = *reinterpret_cast<const double2 *>(src + ww * TW);
                    put(pgrp + rsub + g * NCOL, (s4[0].x + s4[1].x) + (s4[2].x + s4[3].x));
                    put(pgrp + rsub + g * NCOL + 8 * (long long)A.pcol, (s4[0].y + s4[1].y) + (s4[2].y + s4[3].y));
                }
                if (NBX > 0 && tid < 64 * NBX) {
                    double2 s4[NW];
#pragma unroll
                    for (int ww = 0; ww < NW; ++ww) s4[ww] = *reinterpret_cast<const double2 *>(&tile[0][0] + ww * TW + sub * DSZ + DQ0 + tid * 2);
                    put(pquad + rsub, (s4[0].x + s4[1].x) + (s4[2].x + s4[3].x));
                    put(pquad + rsub + A.pcol, (s4[0].y + s4[1].y) + (s4[2].y + s4[3].y));
                }
            }
            pgrp += (long long)NSUB * LBLK * A.pcol;
            pquad += (long long)NSUB * LBLK * A.pcol;
            lds_barrier();  // D tiles consumed: the tiles may be overwritten by the next block's recursion
        }
    }  // ring groups of this m
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_exec_flops[0], (unsigned long long)n_mf * (32ull * (NG * 2048ull + NBX * 512ull)));
        atomicAdd(&g_exec_flops[1], (unsigned long long)n_rec * (64ull * LBLK * 4ull));
    }
}

// =====================================================================================
// partial sums -> alm (4x4x4 kernels: fixed order over the ring groups of each m; pipelined kernel: rows already summed)
// =====================================================================================
template <int SPIN>
__global__ __launch_bounds__(256) void k_alm_reduce(PlanDev P, const LegTask *__restrict__ tasks,
                                                    const MTasks *__restrict__ of_m,
                                                    const double *__restrict__ partial, long long row0, int m0, int ms,
                                                    int ncomp, int ng, int ncol, const double *__restrict__ fl, int add,
                                                    double2 *__restrict__ alm, long long alm_stride,
                                                    const long long *__restrict__ arow, const double *__restrict__ alphan)
{
    const int m = m0 + blockIdx.x * ms, lmax = P.lmax;
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const MTasks mt = of_m[m];
    const int nl = lmax - m + 1, nc = 8 * ng;
    if (arow) {
        // rows of the pipelined kernel: already summed over the ring groups, one span per m (row0 = first row of the chunk
        // in that numbering); what is left is the change of layout (x fl)
        alm_rows_to_layout<SPIN>(P, m, mt.count > 0, partial, arow[m] - row0, ncol, ncomp, ng, fl, add, alm, alm_stride, alphan, threadIdx.x, blockDim.x);
        return;
    }
    for (int i = threadIdx.x; i < nl * nc; i += blockDim.x) {
        const int l = m + i / nc, c = i % nc;
        if (c >= ncomp) continue;
        double2 v = make_double2(0.0, 0.0);
        if (l >= l0) {
            // component c: column group c/8, columns 2(c%8), +1 (spin 2: comp 2f+e of field f)
            const int col = (c >> 3) * NCOL + 2 * (c & 7);
            for (int t = 0; t < mt.count; ++t) {
                const double *p = partial + (tasks[mt.first + t].pout - row0 + (l - l0)) * ncol + col;
                v.x += p[0];
                v.y += p[1];
            }
            if (fl) { v.x *= fl[l]; v.y *= fl[l]; }
        }
        double2 *dst = alm + (long long)c * alm_stride + almidx(lmax, l, m);
        if (add) { const double2 o = *dst; v.x += o.x; v.y += o.y; }
        *dst = v;
    }
}

// =====================================================================================
// host side: task list, m-chunking, launch sequence
// =====================================================================================
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

// Column layout of one sweep over nb components (2 real columns per spin-0 map, 4 per spin-2 field):
//   cols <= 8 : no matrix instruction at all -- one sweep per map / field of the vector-unit kernel (valu);
//   else      : ng full 16-column groups + nbx extra 4-column blocks on the pipelined kernel.
// The pipelined kernel is matrix-bound, so a sweep costs what its (4-column padded) columns cost whatever the
// split; 32 columns is what the B operands of two ring sets leave of the register file.
struct SweepShape {
    int ng, nbx, valu, ncol, oneset, duo;
};
static SweepShape sweep_shape(int spin, int nb)
{
    const int cols = 2 * nb;
    SweepShape sh = {0, 0, 0, 0, 0, 0};
    // <= 4 spin-0 maps / <= 2 spin-2 fields: one sweep per map / field on the vector unit (hx_legendre_valu.hip)
    if (cols <= 8) { sh.valu = 1; return sh; }
    // k_legendre_duo: one ring set per wave; spin 2 keeps one operand of the two positions in registers (HALFB): up to two 16-column
    // groups + two 4-column blocks (ten fields); spin 0 one group + one block (ten maps) or two groups (sixteen)
    sh.duo = 1;
    sh.oneset = spin == 2;  // (task set of 4 ring blocks)
    sh.ng = std::max(cols / NCOL, 1);
    const int rem = cols > NCOL ? cols % NCOL : 0;
    const int maxbx = spin == 2 ? 2 : (sh.ng == 1 ? 1 : 0);
    if (rem > 4 * maxbx) sh.ng += 1;
    else sh.nbx = (rem + 3) / 4;
    sh.ncol = NCOL * sh.ng + 4 * sh.nbx;
    return sh;
}
int analysis_max_comp(int spin) { return spin == 2 ? 8 * NGMAX + 4 : 8 * NGMAX; }

// Components of the next sweep when `remaining` are left.  Resident inputs: the split that costs least by the measured sweep times
// (ms at nside 4096 / lmax 6144; only their ratios matter): a sweep pays for its PADDED columns and ~76 ms of recursion, flush and
// dead stages whatever it holds, so 13 spin-2 fields are 8 + 5 (two full shapes: 316 + 224) rather than 7 + 6 (two padded
// 32-column sweeps: 632), 26 spin-0 maps 16 + 10 rather than 13 + 13, ten fields one 40-column sweep, and one or two left-over
// maps / fields go to the vector-unit kernel.  Ties take the larger sweep first.
static double sweep_cost(int spin, int units)
{
    // k_legendre_duo with the gaps of its matrix stream (round 4: gpurun_out/ab_gap3.txt, ab_gap4.txt)
    if (spin == 0) return units <= 4 ? 21.6 * units : (units <= 8 ? 84.1 : (units <= 10 ? 99.2 : 139.7));
    return units <= 2 ? 61.0 * units : (units <= 4 ? 164.5 : (units == 5 ? 194.1 : (units == 6 ? 222.0 : (units <= 8 ? 272.0 : (units == 9 ? 299.6 : 326.0)))));
}
int analysis_next_batch(int spin, int remaining)
{
    const int unit = spin == 0 ? 1 : 2, maxu = analysis_max_comp(spin) / unit;
    const int units = remaining / unit;
    if (units <= 0) return 0;
    std::vector<double> best(units + 1, 0.0);
    std::vector<int> first(units + 1, 0);
    for (int r = 1; r <= units; ++r) {
        best[r] = 1e300;
        for (int b = std::min(maxu, r); b >= 1; --b) {  // (descending: a tie keeps the larger sweep)
            const double c = sweep_cost(spin, b) + best[r - b];
            if (c < best[r] - 1e-9) { best[r] = c; first[r] = b; }
        }
    }
    return unit * first[units];
}
// the largest sweep of a call over `ncomp` resident components (scratch of the Jacobi iterations)
int analysis_max_batch(int spin, int ncomp)
{
    int mx = 0;
    for (int left = ncomp; left > 0;) {
        const int nb = analysis_next_batch(spin, left);
        if (nb <= 0) break;
        mx = std::max(mx, nb);
        left -= nb;
    }
    return mx;
}

// ts[0], ts[1]: spin 0 / spin 2 tasks of NW ring blocks (ts[3]: one ring set per wave; ts[4], ts[5]: the vector-unit kernel)
static int build_task_set(hx_plan *pl, int spin, int nw, hx_plan::TaskSet &ts);
int build_tasks(hx_plan *pl, int spin)
{
    return build_task_set(pl, spin, spin == 0 ? LegCfg<0>::NW : LegCfg<2>::NW, pl->ts[spin ? 1 : 0]);
}
static int build_task_set(hx_plan *pl, int spin, int nw, hx_plan::TaskSet &ts)
{
    if (ts.built) return HX_OK;
    const int lmax = pl->lmax;
    const int nrb = (pl->nrp + RBLK - 1) / RBLK;
    ts.tasks.clear();
    ts.of_m.assign(lmax + 1, MTasks{0, 0});
    ts.rows_before_m.assign(lmax + 2, 0);
    ts.arow.assign(lmax + 2, 0);
    long long rows = 0, arows = 0;
    // mlim is monotone in the ring index (pole -> equator): first active ring by bisection
    std::vector<int> mlim(pl->nrp);
    for (int rp = 0; rp < pl->nrp; ++rp) mlim[rp] = ring_mlim(lmax, spin, pl->h_sth[rp], pl->h_z[rp]);
    for (int m = 0; m <= lmax; ++m) {
        ts.rows_before_m[m] = rows;
        ts.arow[m] = arows;
        const int l0 = spin == 0 ? m : std::max(m, 2);
        if (l0 <= lmax) arows += (long long)LBLK * ((lmax - l0) / LBLK + 1);
        ts.of_m[m].first = (int)ts.tasks.size();
        if (l0 <= lmax) {
            int first = (int)(std::lower_bound(mlim.begin(), mlim.end(), m) - mlim.begin());
            if (first >= pl->nrp) first = pl->nrp - 1;
            for (int rb = first / RBLK; rb < nrb; rb += nw) {
                LegTask t;
                t.m = m; t.rb0 = rb; t.nrb = std::min(nw, nrb - rb); t.pad = 0; t.pout = rows;
                rows += (long long)LBLK * ((lmax - l0) / LBLK + 1);  // padded to whole 32-l blocks (the pipelined kernel stores unconditionally)
                ts.tasks.push_back(t);
            }
        }
        ts.of_m[m].count = (int)ts.tasks.size() - ts.of_m[m].first;
    }
    ts.rows_before_m[lmax + 1] = rows;
    ts.arow[lmax + 1] = arows;
    HX_TRY(upload(ts.d_tasks, ts.tasks));
    HX_TRY(upload(ts.d_of_m, ts.of_m));
    HX_TRY(upload(ts.d_arow, ts.arow));
    ts.built = true;
    return HX_OK;
}

static bool duo_shape(const SweepShape &sh) { return sh.duo != 0; }
// doubles per accumulation row of a sweep: the rows of the 36- and 40-column shapes start on 128-byte lines (the atomics of a flush -- 16
// lanes x 8 B per row and column group -- then touch whole aligned lines instead of straddling two: -10 ms for ten fields)
static int sweep_pcol(const SweepShape &sh) { return sh.ncol > 2 * NCOL ? (sh.ncol + 15) / 16 * 16 : sh.ncol; }

// k_legendre_duo for a sweep shape (one work-group per order of the grid)
template <int SPIN>
static int launch_duo(const SweepShape &sh, dim3 pgrid, hipStream_t st, const LegParams &A, const double2 *cn)
{
    const dim3 db(256);
    const int key = sh.ng * 10 + sh.nbx;
    // two l-blocks per flush wherever the second accumulator set fits the 256 registers (ten spin-0 maps 105 -> 101 ms, five spin-2 fields
    // 206 -> 198, eight 289 -> 276, nine 320 -> 304; sixteen spin-0 maps spill: 147 -> 157; ten fields spill 43 registers: 345 -> 423);
    // one block per flush for the shapes named there (bit-identical results either way)
    if (key == 10) hipLaunchKernelGGL((k_legendre_duo<SPIN, 1, 0, 2>), pgrid, db, 0, st, A, cn);
    else if (key == 11) hipLaunchKernelGGL((k_legendre_duo<SPIN, 1, 1, 2>), pgrid, db, 0, st, A, cn);
    else if (key == 20 && SPIN == 2) hipLaunchKernelGGL((k_legendre_duo<2, 2, 0, 2>), pgrid, db, 0, st, A, cn);
    else if (key == 12 && SPIN == 2) hipLaunchKernelGGL((k_legendre_duo<2, 1, 2, 2>), pgrid, db, 0, st, A, cn);
    else if (key == 21 && SPIN == 2) hipLaunchKernelGGL((k_legendre_duo<2, 2, 1, 2>), pgrid, db, 0, st, A, cn);  // (its D tiles need 72 KiB of LDS)
    else if (key == 20) hipLaunchKernelGGL((k_legendre_duo<SPIN, 2, 0>), pgrid, db, 0, st, A, cn);
    else if (key == 22 && SPIN == 2) hipLaunchKernelGGL((k_legendre_duo<2, 2, 2>), pgrid, db, 0, st, A, cn);
    else return fail(HX_ERR_ARG, "legendre analysis: no two-group kernel for %d groups + %d blocks", sh.ng, sh.nbx);
    return HX_OK;
}

template <int SPIN>
static int launch_chunk(hx_plan *pl, hx_plan::TaskSet &ts, int m0, int m1, int nb, const SweepShape &sh, const double *d_rw,
                        const double *d_fl, int add, double2 *d_alms)
{
    // column groups of the F / partial rows: full groups (+ 1 holding the extra blocks)
    const int ng = sh.ng + (sh.nbx > 0 ? 1 : 0), ncol = sh.ncol;
    const int pcol = sweep_pcol(sh);
    hipStream_t st = rt().stream;
    PlanDev P = pl->dev();
    const int t0 = ts.of_m[m0].first;
    const int t1 = ts.of_m[m1 - 1].first + ts.of_m[m1 - 1].count;
    const int ms = std::max(pl->m_step, 1), nm = (m1 - m0 + ms - 1) / ms;  // the orders m0 + k ms < m1
    {
        ProfScope ps("fourier_combine");
        dim3 grid(nm, pl->nrp_pad / 32);
        hipLaunchKernelGGL(k_fourier_combine<SPIN>, grid, dim3(256), 0, st, P, pl->Y.as<double2>(), nb, ng, ncol, m0, ms, d_rw,
                           ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(), pl->F.as<double>(), 0);
    }
    if (t1 > t0) {
        ProfScope ps("legendre_analysis");
        ProfScope ps2(SPIN == 0 ? "legendre_analysis_s0" : "legendre_analysis_s2");
        LegParams A;
        A.P = P; A.tasks = ts.d_tasks.as<LegTask>() + t0; A.F = pl->F.as<double>(); A.partial = pl->partial.as<double>();
        A.m0 = m0; A.ms = ms; A.row0 = ts.rows_before_m[m0]; A.ng = ng; A.ncol = ncol; A.pcol = pcol;
        A.of_m = ts.d_of_m.as<MTasks>(); A.arow = ts.d_arow.as<long long>(); A.arow0 = ts.arow[m0]; A.add_all = 0;
        A.tasks = ts.d_tasks.as<LegTask>();  // (the kernel indexes the whole list through of_m)
        // (no memset of the rows: the first ring group of an m stores them)
        const double2 *cn = SPIN == 0 ? pl->cn0.as<double2>() : pl->cn2.as<double2>();
        HX_TRY(launch_duo<SPIN>(sh, dim3((unsigned)nm), st, A, cn));
    }
    {
        ProfScope ps("alm_reduce");
        hipLaunchKernelGGL(k_alm_reduce<SPIN>, dim3(nm), dim3(256), 0, st, P, ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(),
                           pl->partial.as<double>(), ts.arow[m0], m0, ms, nb, ng, pcol, d_fl, add, d_alms, pl->nlm, ts.d_arow.as<long long>(),
                           SPIN == 0 ? pl->al0.as<double>() : pl->al2.as<double>());
    }
    HX_HIP(hipGetLastError());
    return HX_OK;
}

// Batches of <= 4 spin-0 maps / <= 2 spin-2 fields: one sweep PER MAP / FIELD of the vector-unit kernel (hx_legendre_valu.hip).
// The ring Fourier stage runs once for the whole batch.
static bool valu_batch(int spin, int nb) { return sweep_shape(spin, nb).valu != 0; }

int valu_tasks(hx_plan *pl, int spin, hx_plan::TaskSet **out, int blocks)
{
    if (spin) HX_TRY(ensure_rec2(pl));
    if (blocks <= 0) blocks = valu_task_blocks(spin);
    if (blocks != valu_task_blocks(spin) && blocks != 8) return fail(HX_ERR_ARG, "valu_tasks: %d ring blocks per task", blocks);
    hx_plan::TaskSet &ts = blocks == 8 && blocks != valu_task_blocks(spin) ? (spin ? pl->ts[6] : pl->ts[7]) : (spin ? pl->ts[4] : pl->ts[5]);
    HX_TRY(build_task_set(pl, spin, blocks, ts));
    *out = &ts;
    return HX_OK;
}

int synth_duo_tasks(hx_plan *pl, int spin, hx_plan::TaskSet **out)
{
    if (spin) HX_TRY(ensure_rec2(pl));
    hx_plan::TaskSet &ts = spin ? pl->ts[3] : pl->ts[2];
    HX_TRY(build_task_set(pl, spin, PipeCfg<2>::NW * (spin ? PipeCfg<2>::RBS : PipeCfg<0>::RBS), ts));
    // the highest order every ring pair is synthesised for: the tasks of an order m start at the 32-ring-pair block that holds the first
    // ring with mlim >= m (build_task_set), so ring pair rp is covered for m <= the largest mlim of its block; the rows beyond are never
    // written and the spectrum pass does not read them (no 32 GB memset per sweep of ten fields)
    DevBuf &lim = spin ? pl->syn_mlim2 : pl->syn_mlim0;
    if (!lim.p) {
        std::vector<int> h(pl->nrp_pad, -1);
        for (int rp = 0; rp < pl->nrp; ++rp) {
            const int last = std::min(rp / RBLK * RBLK + RBLK - 1, pl->nrp - 1);
            h[rp] = std::min(pl->lmax, ring_mlim(pl->lmax, spin, pl->h_sth[last], pl->h_z[last]));
        }
        HX_TRY(upload(lim, h));
    }
    *out = &ts;
    return HX_OK;
}

static int analysis_batch_valu(hx_plan *pl, int spin, int nb, const double *d_maps, double2 *d_alms, const double *d_rw,
                               const double *d_pw, const double *d_fl, int add)
{
    hx_plan::TaskSet *tsp = nullptr;
    HX_TRY(valu_tasks(pl, spin, &tsp));
    hx_plan::TaskSet &ts = *tsp;
    if (pl->hsrc == nullptr && pl->nssrc == nullptr) {
        HX_TRY(pl->Y.alloc(sizeof(double2) * (size_t)pl->ny * nb));
        HX_TRY(launch_ring_subdft_maps(pl, nb, d_maps, d_pw, pl->Y.as<double2>()));
    }
    double budget = 80e9;
    if (scratch_budget_bytes() > 0.0) budget = scratch_budget_bytes();
    else {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget = std::min(budget, 0.5 * (double)(fr + pl->F.bytes + pl->partial.bytes));
        budget = std::max(budget, 2e9);
    }
    const int lmax = pl->lmax, pcol = valu_partial_cols(spin), unit = spin ? 2 : 1;
    const double f_per_m = (double)pl->nrp_pad * valu_operand_doubles(spin) * sizeof(double);
    const std::vector<long long> &prow = ts.rows_before_m;
    std::vector<std::pair<int, int>> chunks;
    size_t maxF = 16, maxP = 16;
    // a chunk [m0, m1) holds the orders m0, m0 + ms, ... < m1 (ms = 1 but on the m-sharded route); m1 - 1 is its last order
    const int m_end = pl->m_hi < 0 ? lmax + 1 : std::min(pl->m_hi, lmax + 1), ms = std::max(pl->m_step, 1);
    for (int m0 = std::max(pl->m_lo, 0); m0 < m_end;) {
        int last = m0;  // a chunk holds at least one m, whatever the budget
        while (last + ms < m_end) {
            const double bytes = f_per_m * ((last + ms - m0) / ms + 1) + (double)(prow[last + ms + 1] - prow[m0]) * pcol * sizeof(double);
            if (bytes > budget) break;
            last += ms;
        }
        const int m1 = last + 1;
        chunks.emplace_back(m0, m1);
        maxF = std::max(maxF, (size_t)(f_per_m * ((m1 - m0 + ms - 1) / ms)));
        maxP = std::max(maxP, (size_t)(prow[m1] - prow[m0]) * pcol * sizeof(double));
        m0 = last + ms;
    }
    HX_TRY(pl->F.alloc(maxF));
    HX_TRY(pl->partial.alloc(maxP));
    pl->last_chunks = (int)chunks.size();
    PlanDev P = pl->dev();
    for (int c0 = 0; c0 < nb; c0 += unit)
        for (auto &ch : chunks) {
            const int m0 = ch.first, m1 = ch.second, nm = (m1 - m0 + ms - 1) / ms;
            HX_TRY(launch_valu_chunk(pl, spin, ts, m0, m1, c0, d_rw));
            ProfScope ps("alm_reduce");
            if (spin == 0)
                hipLaunchKernelGGL(k_alm_reduce<0>, dim3(nm), dim3(256), 0, rt().stream, P, ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(),
                                   pl->partial.as<double>(), ts.rows_before_m[m0], m0, ms, 1, 1, pcol, d_fl, add, d_alms + (size_t)c0 * pl->nlm, pl->nlm, nullptr, nullptr);
            else
                hipLaunchKernelGGL(k_alm_reduce<2>, dim3(nm), dim3(256), 0, rt().stream, P, ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(),
                                   pl->partial.as<double>(), ts.rows_before_m[m0], m0, ms, 2, 1, pcol, d_fl, add, d_alms + (size_t)c0 * pl->nlm, pl->nlm, nullptr, nullptr);
            HX_HIP(hipGetLastError());
        }
    return HX_OK;
}

// One analysis pass over a batch of <= 8*NGMAX components (device pointers).  F and the
// partial sums are produced per m-chunk so that their footprint stays within a budget
// (HX_SCRATCH_GB, default min(80 GB, half the free HBM)); Y (ring spectra of the batch) persists across chunks.
int analysis_batch(hx_plan *pl, int spin, int nb, const double *d_maps, double2 *d_alms, const double *d_rw,
                   const double *d_pw, const double *d_fl, int add)
{
    if (valu_batch(spin, nb)) return analysis_batch_valu(pl, spin, nb, d_maps, d_alms, d_rw, d_pw, d_fl, add);
    const int sidx = spin ? 1 : 0;
    HX_TRY(build_tasks(pl, spin));
    if (spin) HX_TRY(ensure_rec2(pl));
    // doubles per F / partial row: only the columns in use are stored -- 4-column granularity on the
    // 4x4x4 path (<= 8 columns), 16 per full group + 4 per extra block on the pipelined kernel
    const SweepShape sh = sweep_shape(spin, nb);
    // one ring set per wave: 4 (spin 2) / 8 (spin 0) ring blocks per task
    const bool one_set = sh.oneset || duo_shape(sh);  // (spin 0: 8 ring blocks per task in ts[2])
    if (one_set) HX_TRY(build_task_set(pl, spin, PipeCfg<2>::NW * (spin ? PipeCfg<2>::RBS : PipeCfg<0>::RBS), spin ? pl->ts[3] : pl->ts[2]));
    hx_plan::TaskSet &ts = one_set ? (spin ? pl->ts[3] : pl->ts[2]) : pl->ts[sidx];
    const int ncol = sh.ncol;
    if (pl->hsrc == nullptr && pl->nssrc == nullptr) {
        HX_TRY(pl->Y.alloc(sizeof(double2) * (size_t)pl->ny * nb));
        HX_TRY(launch_ring_subdft_maps(pl, nb, d_maps, d_pw, pl->Y.as<double2>()));
    }

    // budget: hx_set_scratch_budget() / HX_SCRATCH_GB, else 80 GB but never more than half of what is free on
    // the device (what this plan already holds for F / partial counts as free)
    double budget = 80e9;
    if (scratch_budget_bytes() > 0.0) budget = scratch_budget_bytes();
    else {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess)
            budget = std::min(budget, 0.5 * (double)(fr + pl->F.bytes + pl->partial.bytes));
        budget = std::max(budget, 2e9);
    }
    const double f_per_m = (double)pl->nrp_pad * analysis_f_rows(spin) * ncol * sizeof(double);
    const int lmax = pl->lmax;
    // rows of the partial buffer: one span per m (the pipelined kernel adds its ring groups in place)
    const std::vector<long long> &prow = ts.arow;
    const int pcol = sweep_pcol(sh);  // doubles per row of the partial buffer (launch_chunk)
    std::vector<std::pair<int, int>> chunks;
    size_t maxF = 16, maxP = 16;
    // a chunk [m0, m1) holds the orders m0, m0 + ms, ... < m1 (ms = 1 but on the m-sharded route); m1 - 1 is its last order
    const int m_end = pl->m_hi < 0 ? lmax + 1 : std::min(pl->m_hi, lmax + 1), ms = std::max(pl->m_step, 1);
    for (int m0 = std::max(pl->m_lo, 0); m0 < m_end;) {
        int last = m0;  // a chunk holds at least one m, whatever the budget
        while (last + ms < m_end) {
            const double bytes = f_per_m * ((last + ms - m0) / ms + 1) + (double)(prow[last + ms + 1] - prow[m0]) * pcol * sizeof(double);
            if (bytes > budget) break;
            last += ms;
        }
        const int m1 = last + 1;
        chunks.emplace_back(m0, m1);
        maxF = std::max(maxF, (size_t)(f_per_m * ((m1 - m0 + ms - 1) / ms)));
        maxP = std::max(maxP, (size_t)(prow[m1] - prow[m0]) * pcol * sizeof(double));
        m0 = last + ms;
    }
    HX_TRY(pl->F.alloc(maxF));
    HX_TRY(pl->partial.alloc(maxP));
    pl->last_chunks = (int)chunks.size();
    for (auto &ch : chunks) {
        if (spin == 0)
            HX_TRY(launch_chunk<0>(pl, ts, ch.first, ch.second, nb, sh, d_rw, d_fl, add, d_alms));
        else
            HX_TRY(launch_chunk<2>(pl, ts, ch.first, ch.second, nb, sh, d_rw, d_fl, add, d_alms));
    }
    return HX_OK;
}

// ---- a sweep whose rings arrive in slabs (StreamSweep, hx_sht_common.h) --------------------------------------------------------------
// hx_map2alm_multi on host maps used to cut a job into sweeps of 5 fields / 8 maps so that a transform could start before the job's last
// byte had arrived: smaller sweeps cost more (2 x 198 ms instead of 340 for ten fields) and the last of them is exposed behind the upload.
// A ring group of the Legendre kernel needs the rings of that group only -- of ALL maps of the sweep -- so the upload goes slab of rings
// by slab of rings (ascending, poles first: the ring groups of an order then run in the order they always run in) and slab k's ring FFTs,
// operand rows and the ring groups it completes are queued behind it.  What is left behind the last byte is 1 / nslab of one sweep.
static hx_plan::TaskSet &stream_tasks(hx_plan *pl, int spin) { return spin ? pl->ts[3] : pl->ts[2]; }

bool analysis_can_stream(hx_plan *pl, int spin, int nb)
{
    if (!pl || pl->nside < 1 || pl->hsrc || pl->nssrc || pl->m_lo != 0 || pl->m_hi >= 0 || pl->m_step > 1) return false;
    if (nb < 1 || nb > analysis_max_comp(spin)) return false;
    const SweepShape sh = sweep_shape(spin, nb);
    if (!duo_shape(sh) || sh.valu) return false;
    // F and the accumulation rows of ALL orders at once
    const double f_bytes = (double)pl->nrp_pad * analysis_f_rows(spin) * sh.ncol * sizeof(double) * (pl->lmax + 1.0);
    const double p_bytes = 0.5 * (pl->lmax + 1.0) * (pl->lmax + 2.0 + LBLK) * sweep_pcol(sh) * sizeof(double);
    double budget = 80e9;
    if (scratch_budget_bytes() > 0.0) budget = scratch_budget_bytes();
    if (f_bytes + p_bytes > budget) return false;
    // ... and what the streamed sweep holds beside them -- Y of the whole sweep and two staging buffers of whole maps -- must fit the HBM
    // that is free now plus what this plan already holds in those buffers (on a device with less free memory the plan's
    // allocations failed and the call with them; the caller now falls back to the capped sweeps of whole maps, a few GB each)
    size_t fr = 0, tot = 0;
    if (hipMemGetInfo(&fr, &tot) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    const double y_bytes = (double)pl->ny * nb * sizeof(double2), stage_bytes = 2.0 * nb * (double)pl->npix * sizeof(double);
    const double held = (double)pl->F.bytes + (double)pl->partial.bytes + (double)pl->Y.bytes + (double)pl->stage[0].bytes + (double)pl->stage[1].bytes +
                        (double)pl->stage[2].bytes;
    return f_bytes + p_bytes + y_bytes + stage_bytes <= 0.95 * ((double)fr + held);
}

int analysis_stream_plan(hx_plan *pl, int spin, int nb, int nslab, StreamSweep &s)
{
    if (!analysis_can_stream(pl, spin, nb)) return fail(HX_ERR_ARG, "analysis_stream_plan: not a streamable sweep");
    if (spin) HX_TRY(ensure_rec2(pl));
    HX_TRY(build_task_set(pl, spin, PipeCfg<2>::NW * (spin ? PipeCfg<2>::RBS : PipeCfg<0>::RBS), stream_tasks(pl, spin)));
    hx_plan::TaskSet &ts = stream_tasks(pl, spin);
    const SweepShape sh = sweep_shape(spin, nb);
    s.pl = pl; s.spin = spin; s.nb = nb;
    // slab edges: whole 32-ring-pair blocks, about equal numbers of pixels (a ring pair is 8 nsub pixels, the equator ring 4)
    const int nrb = (pl->nrp + RBLK - 1) / RBLK;
    std::vector<double> cum(nrb + 1, 0.0);
    for (int rb = 0; rb < nrb; ++rb) {
        double px = 0.0;
        for (int rp = rb * RBLK; rp < std::min((rb + 1) * RBLK, pl->nrp); ++rp) px += (rp == pl->nrp - 1 ? 4.0 : 8.0) * pl->h_nsub[rp];
        cum[rb + 1] = cum[rb] + px;
    }
    nslab = std::max(1, std::min(nslab, nrb));
    s.rp_edge.assign(1, 0);
    for (int k = 1; k < nslab; ++k) {
        const int rb = (int)(std::lower_bound(cum.begin(), cum.end(), cum[nrb] * k / nslab) - cum.begin());
        if (rb * RBLK > s.rp_edge.back() && rb < nrb) s.rp_edge.push_back(rb * RBLK);
    }
    s.rp_edge.push_back(nrb * RBLK);
    s.nslab = (int)s.rp_edge.size() - 1;
    // the ring groups of order m that slab k completes: those whose LAST ring block lies in it (the ring groups of an m are in ascending order)
    const int lmax = pl->lmax;
    std::vector<MTasks> tab((size_t)s.nslab * (lmax + 1), MTasks{0, 0});
    for (int m = 0; m <= lmax; ++m) {
        const MTasks mt = ts.of_m[m];
        int t = mt.first;
        for (int k = 0; k < s.nslab; ++k) {
            const int rb_hi = s.rp_edge[k + 1] / RBLK;
            MTasks &o = tab[(size_t)k * (lmax + 1) + m];
            o.first = t;
            while (t < mt.first + mt.count && ts.tasks[t].rb0 + ts.tasks[t].nrb <= rb_hi) ++t;
            o.count = t - o.first;
        }
        if (t != mt.first + mt.count) return fail(HX_ERR_ARG, "analysis_stream_plan: ring groups of m = %d left over", m);
    }
    HX_TRY(upload(s.d_of_m, tab));
    // scratch for all orders at once (DevBuf::alloc keeps what is large enough: planning every sweep of a call first means no buffer is
    // re-allocated -- a device-wide synchronisation -- between its sweeps)
    HX_TRY(pl->Y.alloc(sizeof(double2) * (size_t)pl->ny * nb));
    HX_TRY(pl->F.alloc((size_t)pl->nrp_pad * analysis_f_rows(spin) * sh.ncol * sizeof(double) * (size_t)(lmax + 1)));
    HX_TRY(pl->partial.alloc((size_t)ts.arow[lmax + 1] * sweep_pcol(sh) * sizeof(double)));
    return HX_OK;
}

int analysis_stream_start(StreamSweep &s)
{
    hx_plan *pl = s.pl;
    const SweepShape sh = sweep_shape(s.spin, s.nb);
    // the accumulation rows start at zero: every ring group adds
    HX_HIP(hipMemsetAsync(pl->partial.p, 0, (size_t)stream_tasks(pl, s.spin).arow[pl->lmax + 1] * sweep_pcol(sh) * sizeof(double), rt().stream));
    pl->last_chunks = 1;
    return HX_OK;
}

template <int SPIN>
static int stream_slab(StreamSweep &s, int k)
{
    hx_plan *pl = s.pl;
    hx_plan::TaskSet &ts = stream_tasks(pl, SPIN);
    const SweepShape sh = sweep_shape(SPIN, s.nb);
    const int ng = sh.ng + (sh.nbx > 0 ? 1 : 0), lmax = pl->lmax, nm = lmax + 1;
    const int rp_lo = s.rp_edge[k], rp_hi = s.rp_edge[k + 1];
    hipStream_t st = rt().stream;
    PlanDev P = pl->dev();
    HX_TRY(launch_ring_subdft_maps(pl, s.nb, s.d_maps, s.d_pw, pl->Y.as<double2>(), rp_lo, std::min(rp_hi, pl->nrp)));
    {
        ProfScope ps("fourier_combine");
        const int tile_hi = k + 1 == s.nslab ? pl->nrp_pad / 32 : rp_hi / 32;  // (the padding rows behind the last ring pair are zeroed with the last slab)
        dim3 grid(nm, tile_hi - rp_lo / 32);
        hipLaunchKernelGGL(k_fourier_combine<SPIN>, grid, dim3(256), 0, st, P, pl->Y.as<double2>(), s.nb, ng, sh.ncol, 0, 1, s.d_rw,
                           ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(), pl->F.as<double>(), rp_lo / 32);
    }
    {
        ProfScope ps("legendre_analysis");
        ProfScope ps2(SPIN == 0 ? "legendre_analysis_s0" : "legendre_analysis_s2");
        LegParams A;
        A.P = P; A.tasks = ts.d_tasks.as<LegTask>(); A.F = pl->F.as<double>(); A.partial = pl->partial.as<double>();
        A.m0 = 0; A.ms = 1; A.row0 = 0; A.ng = ng; A.ncol = sh.ncol; A.pcol = sweep_pcol(sh);
        A.of_m = s.d_of_m.as<MTasks>() + (size_t)k * (lmax + 1); A.arow = ts.d_arow.as<long long>(); A.arow0 = 0; A.add_all = 1;
        HX_TRY(launch_duo<SPIN>(sh, dim3((unsigned)nm), st, A, SPIN == 0 ? pl->cn0.as<double2>() : pl->cn2.as<double2>()));
    }
    HX_HIP(hipGetLastError());
    return HX_OK;
}

int analysis_stream_slab(StreamSweep &s, int k)
{
    if (k < 0 || k >= s.nslab) return fail(HX_ERR_ARG, "analysis_stream_slab: slab %d of %d", k, s.nslab);
    return s.spin == 0 ? stream_slab<0>(s, k) : stream_slab<2>(s, k);
}

int analysis_stream_end(StreamSweep &s)
{
    hx_plan *pl = s.pl;
    hx_plan::TaskSet &ts = stream_tasks(pl, s.spin);
    const SweepShape sh = sweep_shape(s.spin, s.nb);
    const int ng = sh.ng + (sh.nbx > 0 ? 1 : 0), nm = pl->lmax + 1, pcol = sweep_pcol(sh);
    PlanDev P = pl->dev();
    ProfScope ps("alm_reduce");
    if (s.spin == 0)
        hipLaunchKernelGGL(k_alm_reduce<0>, dim3(nm), dim3(256), 0, rt().stream, P, ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(), pl->partial.as<double>(),
                           0LL, 0, 1, s.nb, ng, pcol, s.d_fl, 0, s.d_alms, pl->nlm, ts.d_arow.as<long long>(), pl->al0.as<double>());
    else
        hipLaunchKernelGGL(k_alm_reduce<2>, dim3(nm), dim3(256), 0, rt().stream, P, ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(), pl->partial.as<double>(),
                           0LL, 0, 1, s.nb, ng, pcol, s.d_fl, 0, s.d_alms, pl->nlm, ts.d_arow.as<long long>(), pl->al2.as<double>());
    HX_HIP(hipGetLastError());
    return HX_OK;
}

}  // namespace hx

// Executed matrix-instruction flops of one hx_map2alm(niter = 0) call with ncomp components:
// every wave-block of the task list issues 8 (ring quads) x 2 (parities) x NOP MFMAs per full
// 16-column group (2048 flop each) and per 4-column block (512 flop each).  Blocks whose rings
// are all still dead (below 2^-100 for spin 2, 2^-300 for spin 0) skip their MFMAs, so this is an upper bound (by 9 - 13 %).
extern "C" int hx_plan_mfma_flops(hx_plan *pl, int spin, int ncomp, double *flops)
{
    using namespace hx;
    if (!pl || !flops || ncomp < 1 || (spin != 0 && spin != 2)) return fail(HX_ERR_ARG, "hx_plan_mfma_flops: bad arguments");
    HX_TRY(ensure_ready());
    HX_TRY(build_tasks(pl, spin));
    const hx_plan::TaskSet &ts = pl->ts[spin ? 1 : 0];
    const int nop = spin ? 2 : 1, l0min = spin ? 2 : 0;
    double wave_blocks = 0.0;
    for (const LegTask &t : ts.tasks) {
        const int l0 = std::max(t.m, l0min);
        wave_blocks += (double)t.nrb * ((pl->lmax - l0) / LBLK + 1);
    }
    double per_wave_block = 0.0;
    for (int c0 = 0, nb = 0; c0 < ncomp; c0 += nb) {
        nb = analysis_next_batch(spin, ncomp - c0);
        const SweepShape sh = sweep_shape(spin, nb);
        per_wave_block += 16.0 * nop * (sh.ng * 2048.0 + sh.nbx * 512.0);  // (sweeps on the vector unit issue none)
    }
    *flops = wave_blocks * per_wave_block;
    return HX_OK;
}

// FP64 flops the Legendre analysis kernels EXECUTED since the last reset, counted by the kernels themselves: out2[0] matrix
// instructions (stages that skip theirs because every ring of the set is still dead are not counted), out2[1] vector
// unit (recursions of the pipelined kernels, everything of the single-map kernels).  Synchronises the library stream.
extern "C" int hx_executed_flops(double *out2, int reset)
{
    using namespace hx;
    if (!out2) return fail(HX_ERR_ARG, "hx_executed_flops: null output");
    HX_TRY(ensure_ready());
    HX_HIP(hipStreamSynchronize(rt().stream));
    unsigned long long v[2] = {0, 0}, w = 0;
    HX_HIP(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_exec_flops), sizeof(v)));
    HX_TRY(valu_exec_flops(&w, reset != 0));
    if (reset) {
        const unsigned long long z[2] = {0, 0};
        HX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_exec_flops), z, sizeof(z)));
    }
    out2[0] = (double)v[0];
    out2[1] = (double)v[1] + (double)w;
    return HX_OK;
}

// The same with the vector-unit share beside it: out2[0] = matrix-instruction flops (as above), out2[1] = FP64 flops of the
// recursions (4 per value lambda_lm(theta): every sweep runs the chains of all ring pairs of the task list once; spin 2 runs
// two functions per ring pair).  executed = out2[0] + out2[1] is what bench.py's roofline fraction is quoted on.
extern "C" int hx_plan_executed_flops(hx_plan *pl, int spin, int ncomp, double *out2)
{
    using namespace hx;
    if (!out2) return fail(HX_ERR_ARG, "hx_plan_executed_flops: null output");
    HX_TRY(hx_plan_mfma_flops(pl, spin, ncomp, &out2[0]));
    const hx_plan::TaskSet &ts = pl->ts[spin ? 1 : 0];
    const int nop = spin ? 2 : 1, l0min = spin ? 2 : 0;
    double wave_blocks = 0.0;
    for (const LegTask &t : ts.tasks) {
        const int l0 = std::max(t.m, l0min);
        wave_blocks += (double)t.nrb * ((pl->lmax - l0) / LBLK + 1);
    }
    int sweeps = 0;
    for (int c0 = 0, nb = 0; c0 < ncomp; c0 += nb, ++sweeps) nb = analysis_next_batch(spin, ncomp - c0);
    out2[1] = wave_blocks * sweeps * (double)RBLK * LBLK * nop * 4.0;
    return HX_OK;
}
