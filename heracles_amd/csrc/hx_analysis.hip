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
// The kernel here is the software-pipelined k_legendre_pipe (batches of 5 or more spin-0 maps / 3 or more spin-2 fields: one
// work-group per m that walks its ring groups and adds them in place with f64 atomics in a fixed order -- `partial` holds ONE span of
// rows per m and k_alm_reduce only changes the layout).  Smaller batches run on the vector unit (hx_legendre_valu.hip: one span of
// rows per ring group, summed by k_alm_reduce).  Round 1's barrier-phased kernel on the 4x4x4 instruction (k_legendre_analysis)
// was retired in round 3: 61 / 108 ms for one spin-0 map / spin-2 field against 23 / 65 ms.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "hx_sht_common.h"

namespace hx {
using namespace hxfft;

// HX_HALF_F: spin-2 operand rows of the odd parity are not stored.  With x = B+(P+_N), y = B-(P-_S) (operand of lambda+; lambda-
// alike) the two rows are x + y and x - y, and x - y is x + y with the four columns of a field reversed and the signs (-, +, +, -)
// under lambda+, (+, -, -, +) under lambda-.  The Legendre kernel loads the even row a second time at column j ^ 3 (plain loads:
// anything else in its prologue -- a DPP move, a sign flip -- moves its 256 registers of operands around and costs the stages 140
// register copies each: 496 instead of 400 ms); the signs go where they are free: the lambda- chains run with alternating
// sign (-p', +q' in their coefficient table, the seed negated for m = 1), which leaves (-, +, +, -) on the odd-parity rows under
// BOTH functions, and k_alm_reduce<2> flips those when it changes the layout.  Half the F rows: half of what k_fourier_combine<2>
// writes and the Legendre prologue fetches from HBM.
#ifndef HX_HALF_F
#define HX_HALF_F 1
#endif
template <int SPIN>
struct LegCfg {
    static constexpr int NW = SPIN == 0 ? 16 : 8;   // waves per workgroup
    static constexpr int NT = SPIN == 0 ? 1 : 2;    // 16x64 tiles per wave
    static constexpr int NOP = SPIN == 0 ? 1 : 2;   // A-operand functions per ring
    static constexpr int NPAR = (SPIN == 2 && HX_HALF_F) ? 1 : 2;  // parity rows of F per (m, ring pair)
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
    unsigned long long *counters;      // diagnostic builds only (HX_PIPE_ABL & 8: cycle accounting per stage kind)
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

// =====================================================================================
// Legendre analysis, software-pipelined (batches of >= 3 spin-0 maps / >= 3 spin-2 fields)
// =====================================================================================
// Measured on this part (tools/ubench_power.hip, profiles/r02_ubench_power.txt): a v_mfma_f64_16x16x4 stream holds
// 77 TFLOP/s at 2.38 GHz from ONE wave per SIMD, and FP64 VALU work + LDS stores placed BETWEEN the MFMAs of
// the same wave cost 7 % -- whereas the same VALU work in ANOTHER wave of that SIMD is starved by the MFMA
// stream (100 cycles per dependent FMA).  The first kernel above keeps recursion, matrix work and flush in
// separate, barrier-locked phases (matrix pipe busy 48-63 % of the time).  This one has
//   * 4 waves per work-group, one per SIMD (up to 512 registers each), one work-group per CU;
//   * every wave owns NSETS = 2 ring sets (spin 0: 64 ring pairs, both parity chains in one lane; spin 2: 32 ring
//     pairs x the two Wigner functions) with a private LDS tile pair each;
//   * the recursion of the NEXT (set, 32-l block) is interleaved, instruction by instruction, with the MFMAs of
//     the CURRENT one: stage (b, 0) = MFMA(set 0, block b) || recursion(set 1, block b),
//                      stage (b, 1) = MFMA(set 1, block b) || recursion(set 0, block b + 1);
//   * B operands of both sets, D tiles, chain states stay in registers; recursion coefficients come through a
//     triple-buffered LDS table (staged two blocks ahead at the flush barrier -- no scalar-memory loads inside
//     the loop, whose out-of-order return would force lgkmcnt(0) drains of the LDS queue);
//   * the flush combines the 4 waves' D tiles through the wave's second tile (free at that point) in a fixed
//     order, as before.
// Layouts of F and the task list are those of the first kernel (a task = 16 / 8 blocks of 32 ring pairs); the rows of
// `partial` are shared by the tasks of an m (LegParams::arow).
// Work the Legendre kernels EXECUTE, counted by the kernels themselves (one atomic per wave at its end; wave-uniform scalar
// counters): [0] FP64 flops of the matrix instructions actually issued (stages whose ring set is still dead (set_mode) skip theirs),
// [1] FP64 vector flops of the recursions (2 FMAs per generated value).  bench.py's roofline fraction is quoted on these; they agree
// with SQ_INSTS_VALU_MFMA_F64 of the PMC passes (profiles/).
__device__ unsigned long long g_exec_flops[2];

template <int SPIN>
struct PipeCfg {
    static constexpr int NW = 4;                              // waves per work-group
    static constexpr int NSETS = 2;                           // ring sets per wave
    static constexpr int RBS = SPIN == 0 ? 2 : 1;             // 32-ring-pair blocks per set
    static constexpr int NCH = SPIN == 0 ? 2 : 1;             // recursion chains per lane and set
    static constexpr int NOP = SPIN == 0 ? 1 : 2;
    static_assert(NW * NSETS * RBS == LegCfg<SPIN>::NW, "a task covers the same ring blocks as in the first kernel");
};

#ifndef HX_PIPE_ABL
#define HX_PIPE_ABL 0  // timing experiments only (tools/build_diag.sh): 1 no MFMA, 2 no recursion, 4 no flush, 8 cycle accounting
#endif

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

// NSUB: 32-l blocks per flush (their D tiles stay in registers): one work-group reduction and one pair of barriers per
// NSUB * 32 l
// ONESET: ONE ring set per wave (a task = 4 / 8 ring blocks instead of 8 / 16): the B operands of 40 columns (two groups + two
// 4-column blocks) then fit the register file, a stage does the matrix work of l-block t on one tile while the SAME set's
// recursion fills the other tile with l-block t + 1, and a flush closes the two l-blocks of a stage pair.  Ten spin-2 fields
// are one sweep instead of two: the per-stage overheads (recursion block, stage glue) are paid once for twice the matrix work.
template <int SPIN, int NG, int NBX, int NSUB = 2, int ONESET = 0>
__global__ __launch_bounds__(256, 1) void k_legendre_pipe(LegParams A, const double2 *__restrict__ coefn,
                                                          const double *__restrict__ alphan)
{
    using C = PipeCfg<SPIN>;
    constexpr int NW = C::NW, NOP = C::NOP, NCH = C::NCH;
    constexpr int NPAIR = 16;                  // slot pairs per stage: spin 0 q = 0..15, spin 2 (q = 0..7, op); a pair = positions 0, 1
    constexpr int NGA = NG > 0 ? NG : 1, NXA = NBX > 0 ? NBX : 1;
    constexpr int DQ0 = NG * 512;              // first double of the 4-column blocks in a sub-block's D staging area
    constexpr int DSZ = NG * 512 + NBX * 128;  // doubles of one sub-block's D staging area
    constexpr int NCR = 3 * NSUB, NAR = 2 * NSUB;  // blocks in the coefficient / alpha rings
    // doubles per wave of the second tile: the 16 x 64 lambda tile (2048), or the D tiles of a flush if they need more (one ring
    // set, 36 / 40 columns: 2304 / 2560)
    constexpr int TBW = ONESET && NSUB * DSZ > 2048 ? NSUB * DSZ : 2048;
    static_assert(NG >= 1 && NSUB * DSZ <= TBW, "D tiles of a wave must fit its second tile");
    static_assert(!ONESET || NSUB == 2, "one ring set: a flush closes the two l-blocks of a stage pair");
    constexpr int NSB = ONESET ? 1 : 2;       // ring sets per wave
    __shared__ double tileA[NW][2048];         // set 0, 64 KiB
    __shared__ double tileB[NW][TBW];          // set 1; doubles as the D staging area of the flush
    __shared__ double2 coefs[NCR][2][LBLK];    // recursion coefficients, block j in slot j % NCR; [1] = sign of q' flipped (spin 2)
    __shared__ double alphas[NAR][LBLK];       // output scalings alpha_l, block j in slot j % NAR
    // Spin 2, one column group, one block per flush: the D tiles are staged in an area of their own (20 KiB beside the
    // 128 KiB of tiles), so that the waves need not meet again after the reduction: its LDS reads, additions and partial-sum
    // stores ride in the first stage of the next block (DEFER: -2 % on one device; the spin-0 kernel has no 16 registers
    // to carry the operands across a vector block: 49 spilled, +20 %).
    constexpr bool DEFER = SPIN == 2 && NG == 1 && NSUB == 1 && !(HX_PIPE_ABL & 64);
    __shared__ double dstage[DEFER ? NW * DSZ : 2];
    const PlanDev &P = A.P;
    // One work-group per m.  The ring groups (tasks) of that m are swept one after the other, and every flush ADDS its rows
    // into the one span of rows the m owns (global_atomic_add_f64 without return, executed in this XCD's L2):
    // every row element is touched by one thread of one work-group only, in program order, so the sum over ring groups has
    // a fixed order (bit-reproducible) although no partial row per ring group ever exists in HBM.
    const int m = A.m0 + blockIdx.x * A.ms, lmax = P.lmax;
    const MTasks mt = A.of_m[m];
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const int off = (l0 + m) & 1;              // parity (l + m) & 1 of position 0
    const long long cb = almidx(lmax, 0, m);
    const int coff = SPIN == 0 ? 0 : 1;        // spin-2 coefficients are indexed by the target l
    const int nblk = (lmax - l0) / LBLK + 1;
    const long long orow = A.arow[m] - A.arow0;
    // (the rows are zeroed by the host before the launch: a first group that stores instead of adding needs a branch inside
    // the stage that carries the deferred reduction -- 224 vs 229 ms per sweep of 5 spin-2 fields, same device)
    auto put = [](double *p, double v) __attribute__((always_inline)) {
#if HX_PIPE_ABL & 32  // timing experiment: plain stores instead of the atomics (wrong results)
        *p = v;
#else
        __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)p, v);
#endif
    };
    int n_mf = 0, n_rec = 0;  // stages of this wave that issued their matrix instructions / ran a recursion (wave-uniform)
    for (int ti = 0; ti < mt.count; ++ti) {
    const LegTask task = A.tasks[mt.first + ti];
    // the thread index goes through an empty asm statement per ring group: everything derived from it (LDS addresses, row
    // offsets, flush roles) is then set up per group and dies after the group's prologue, as in a kernel without this loop --
    // hoisted out of the loop those values stay live through the stages (36 more registers spilled, measured)
    int tid = threadIdx.x;
    asm volatile("; ring group" : "+v"(tid));
    const int w = tid >> 6, lane = tid & 63;
    const int ai = lane & 15, ak = lane >> 4;
    if (ti) __syncthreads();  // the previous ring group's last readers of the coefficient / alpha rings and of the staging area

    // ---- rings of this lane: ring block (within the task) of set s.  The ring blocks of a task are dealt to
    // the waves round-robin, so that every wave holds polar (late) and equatorial (early) rings alike ----
    double xx[2];
    bool valid[2];
    int rpl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int se = ONESET ? 0 : s;  // (one ring set: both tile parities belong to it)
        const int rbi = SPIN == 0 ? 2 * (se * NW + w) + (lane >> 5) : se * NW + w;  // spin 0: 64 consecutive ring pairs per set
        rpl[s] = (task.rb0 + rbi) * RBLK + (lane & 31);
        valid[s] = rbi < task.nrb && rpl[s] < P.nrp;
        const double x = valid[s] ? P.z[rpl[s]] : 0.0;
        xx[s] = SPIN == 0 ? x * x : x;
    }

    // ---- B operands of both sets: lane (k = lane>>4, j = lane&15) holds F[ring k of the pair][parity of the position][op][column] ----
    double fr[NSB][NPAIR][2][NGA], frx[NSB][NPAIR][2][NXA];
#pragma unroll
    for (int s = 0; s < NSB; ++s)
#pragma unroll
        for (int sp = 0; sp < NPAIR; ++sp) {
            const int op = SPIN == 0 ? 0 : sp & 1, q = SPIN == 0 ? sp : sp >> 1;
            const int rbi = SPIN == 0 ? 2 * (s * NW + w) + (q >> 3) : s * NW + w;
            const bool on = rbi < task.nrb;
            const long long row = (long long)blockIdx.x * P.nrp_pad + (task.rb0 + rbi) * RBLK + pipe_rho(q & 7, ak);
            if (SPIN == 2 && HX_HALF_F) {
                // only the even-parity row x + y is stored; the position of the odd parity (p ^ off = 1; off: m = 1 only) loads the SAME
                // row at column j ^ 3 -- two loads per operand as before, one row of F (signs: see HX_HALF_F)
                const double *f = A.F + (row * NOP + op) * A.ncol;
#pragma unroll
                for (int pos = 0; pos < 2; ++pos) {
                    const int cx = (pos ^ off) ? 3 : 0;
#pragma unroll
                    for (int g = 0; g < NGA; ++g) fr[s][sp][pos][g] = (NG > 0 && on) ? f[g * NCOL + (ai ^ cx)] : 0.0;
#pragma unroll
                    for (int g = 0; g < NXA; ++g) frx[s][sp][pos][g] = (NBX > 0 && on) ? f[NG * NCOL + 4 * g + ((lane & 3) ^ cx)] : 0.0;
                }
            } else {
#pragma unroll
                for (int pos = 0; pos < 2; ++pos) {
                    const double *f = A.F + ((row * 2 + (pos ^ off)) * NOP + op) * A.ncol;
#pragma unroll
                    for (int g = 0; g < NGA; ++g) fr[s][sp][pos][g] = (NG > 0 && on) ? f[g * NCOL + ai] : 0.0;
#pragma unroll
                    for (int g = 0; g < NXA; ++g) frx[s][sp][pos][g] = (NBX > 0 && on) ? f[NG * NCOL + 4 * g + (lane & 3)] : 0.0;
                }
            }
        }

    // ---- seeds ----
    double vc[NSB][NCH], vp[NSB][NCH];
    int sc[NSB][NCH];
#pragma unroll
    for (int s = 0; s < NSB; ++s) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) { vc[s][c] = 0.0; vp[s][c] = 0.0; sc[s][c] = -100; }
        if (valid[s]) {
            if (SPIN == 0) {
                SVal a = spow(P.sth[rpl[s]], m);
                a.v *= P.mfac[m];
                SVal b = a;
                b.v *= sqrt(2.0 * m + 3.0) * P.z[rpl[s]];  // lambda_{m+1,m} = sqrt(2m+3) x lambda_mm
                snorm_small(a);
                snorm_small(b);
                vc[s][0] = a.v; sc[s][0] = a.e;
                vc[s][NCH - 1] = b.v; sc[s][NCH - 1] = b.e;
            } else {
                SVal sp, sm;
                spin2_seeds(m, P.sth[rpl[s]], P.omz[rpl[s]], P.kfac2[m], sp, sm);
                vc[s][0] = (lane >> 5) ? ((HX_HALF_F && off) ? -sm.v : sm.v) : sp.v;  // (HX_HALF_F: the lambda- chain alternates in sign, + at even l + m)
                sc[s][0] = (lane >> 5) ? sm.e : sp.e;
            }
        }
    }
    const int chalf = SPIN == 2 ? (lane >> 5) : 0;  // q' enters with opposite sign for d_{m,+2}: second coefficient table

    auto lds_barrier = []() __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): global stores / prefetches in flight do not hold the barrier
        __builtin_amdgcn_s_barrier();
    };
    // coefficient hand-over: thread t < 128 carries double (t & 63) of a block's 32 (p', q') pairs into table t >> 6
    // table 1 (lambda-): (p', -q'); with HX_HALF_F the chain carries (-1)^(l + m) lambda-, whose recursion has (-p', +q')
    const double csign = (SPIN == 2 && (tid >> 6) == 1 && ((tid & 1) != 0) != (HX_HALF_F != 0)) ? -1.0 : 1.0;
    if (tid < 128) {
#pragma unroll
        for (int bb = 0; bb <= NSUB; ++bb)
            (&coefs[bb][tid >> 6][0].x)[tid & 63] =
                csign * reinterpret_cast<const double *>(coefn + cb + l0 + bb * LBLK + coff)[tid & 63];
    } else if (tid < 160) {
#pragma unroll
        for (int bb = 0; bb < NSUB; ++bb) alphas[bb][tid - 128] = alphan[cb + l0 + bb * LBLK + (tid - 128)];
    }
    __syncthreads();

    // ---- one recursion step of chain c of set S; the caller stores the value it returns (the one BEFORE the step) ----
    // RM 1: chains all dead (nothing stored), 2: mixed (a dead chain stores 0), 3: all live (no exponent bookkeeping)
    auto rec_step = [&](auto SS, auto RMM, int c, int step, const double2 cc) __attribute__((always_inline)) {
        constexpr int S = ONESET ? 0 : decltype(SS)::value, RM = decltype(RMM)::value;
        // every 4 steps: promote a scaled chain that has grown past 1 (value *= 2^-300, exponent += 1) -- on the exponent
        // bits, branch-free (a previous value below 2^-722 becomes 0)
        if (RM != 3 && (step & 3) == 0) {
            const int hc = __double2hiint(vc[S][c]), hp = __double2hiint(vp[S][c]);
            const bool up = sc[S][c] < 0 && (hc & 0x7ff00000) >= 0x3ff00000;
            const int sub = up ? (300 << 20) : 0;
            const bool pz = up && (hp & 0x7ff00000) <= (300 << 20);
            vc[S][c] = __hiloint2double(hc - sub, __double2loint(vc[S][c]));
            vp[S][c] = pz ? 0.0 : __hiloint2double(hp - sub, __double2loint(vp[S][c]));
            sc[S][c] += up ? 1 : 0;
        }
        const double cur = (RM == 3 || sc[S][c] == 0) ? vc[S][c] : 0.0;
        const double vn = fma(fma(cc.x, xx[S], cc.y), vc[S][c], -vp[S][c]);
        vp[S][c] = vc[S][c];
        vc[S][c] = vn;
        return cur;
    };
    // flush roles of this thread: D values arrive as 16-byte chunks (col, chunk c): rows (c >> 1) + 8 (c & 1) and + 4
    const int fcol = tid & 15, fch = (tid >> 4) & 7, fpos = (tid >> 7) & 1;
    const int frow = (fch >> 1) + 8 * (fch & 1);
    // partial rows this thread writes: (group columns) rows 2 frow + fpos and + 8 of the block; (extra blocks) rows 2 qrow, + 1
    const int qrow = tid / (4 * NXA), qcol = tid % (4 * NXA);
    double *pgrp = A.partial + (orow + 2 * frow + fpos) * A.pcol + fcol;
    double *pquad = A.partial + (orow + 2 * qrow) * A.pcol + NG * NCOL + qcol;
    // deferred reduction (DEFER): operands of the group columns, then of the 4-column blocks, in the same registers
    double2 s4[NW];
    bool pend = false;
    int pend_slot = 0;
    double al0 = 0.0, al1 = 0.0;  // alpha_l of the two rows, read with the operands
    auto red_issue_group = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int ww = 0; ww < NW; ++ww)
            s4[ww] = *reinterpret_cast<const double2 *>(dstage + ww * DSZ + fpos * 256 + fcol * 16 + ((fch ^ (fcol & 7)) * 2));
        al0 = alphas[pend_slot][2 * frow + fpos];
        al1 = alphas[pend_slot][2 * frow + 8 + fpos];
    };
    auto red_finish_group = [&]() __attribute__((always_inline)) {
        const double sx = (s4[0].x + s4[1].x) + (s4[2].x + s4[3].x), sy = (s4[0].y + s4[1].y) + (s4[2].y + s4[3].y);
        put(pgrp, sx * al0);
        put(pgrp + 8 * (long long)A.pcol, sy * al1);
        pgrp += (long long)LBLK * A.pcol;
    };
    auto red_issue_quad = [&]() __attribute__((always_inline)) {
        if (NBX > 0 && tid < 64 * NBX) {
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) s4[ww] = *reinterpret_cast<const double2 *>(dstage + ww * DSZ + DQ0 + tid * 2);
            const double2 a2 = *reinterpret_cast<const double2 *>(&alphas[pend_slot][2 * qrow]);
            al0 = a2.x;
            al1 = a2.y;
        }
    };
    auto red_finish_quad = [&]() __attribute__((always_inline)) {
        if (NBX > 0 && tid < 64 * NBX) {
            const double sx = (s4[0].x + s4[1].x) + (s4[2].x + s4[3].x), sy = (s4[0].y + s4[1].y) + (s4[2].y + s4[3].y);
            put(pquad, sx * al0);
            put(pquad + A.pcol, sy * al1);
        }
        pquad += (long long)LBLK * A.pcol;
        pend = false;
    };
    double4_t accs[NSUB][NGA][2];
    double accxs[NSUB][NXA][2];
    // ---- a stage: MFMAs of set SM (if MF) and the recursion of the other set (mode RM).
    // FP64 vector instructions and FP64 MFMAs share one execution resource: a vector FMA placed between the MFMAs of the
    // same wave costs ~17 cycles of matrix-pipe time (4 alone), from another wave of the SIMD it is starved (tools/ubench_slot.hip,
    // tools/ubench_power.hip); 128-bit LDS traffic between MFMAs is free.  So a stage alternates
    //     [HB recursion steps as one tight vector block: coefficients already in registers, results kept in registers]
    //     [HB / 2 slot pairs of MFMAs; in their shadow ONLY LDS traffic: the stores of those HB values, the reads of the next
    //      HB coefficients and of the A operands]                                                 (32 / HB times).
    // cq = coefficients of the next 16 recursion steps, whatever set / block they belong to. ----
#if HX_PIPE_ABL & 8
    // cycle accounting (diagnostic build): [0] prologue + MFMA || mixed recursion, [1] MFMA || all-live recursion, [2] MFMA || dead / no recursion,
    // [3] live recursion alone, [4] dead recursion alone, [5] flush up to the first barrier, [6] reduction + second barrier;
    // [8 + i] = number of intervals of kind i
    unsigned long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long cyc_v = 0, cyc_m = 0, tin = 0;  // inside MFMA || live-recursion stages: vector blocks, matrix blocks
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#define HX_STAMP(i) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); cyc[i] += tn_ - tlast; cnt[i] += 1; tlast = tn_; } while (0)
#else
#define HX_STAMP(i) do { } while (0)
#endif
    constexpr int PF = 2, HB = 8, PPB = HB / 2, NHB = LBLK / HB;  // (16 steps per block need 96 more registers than the wave has)
    double2 cq[HB];
    auto stage = [&](auto SUBB, auto SMM, auto MFF, auto RMM, const double2 *cf_rec, const double2 *cf_next) __attribute__((always_inline)) {
        double4_t (&acc)[NGA][2] = accs[decltype(SUBB)::value];
        double (&accx)[NXA][2] = accxs[decltype(SUBB)::value];
        constexpr int SM = decltype(SMM)::value, SR = 1 - SM, RM = (HX_PIPE_ABL & 2) ? 0 : decltype(RMM)::value;
        constexpr int BS = ONESET ? 0 : SM;  // B operands of the set whose tile the matrix instructions read
        constexpr bool MF = decltype(MFF)::value && !(HX_PIPE_ABL & 1);
        using ISR = std::integral_constant<int, SR>;
        using IRM = std::integral_constant<int, RM>;
        const double *tm = SM == 0 ? &tileA[w][0] : &tileB[w][0];
        double *tr = SR == 0 ? &tileA[w][0] : &tileB[w][0];
        auto a_fetch = [&](int sp) __attribute__((always_inline)) {
            const int op = SPIN == 0 ? 0 : sp & 1, q = SPIN == 0 ? sp : sp >> 1;
            const int c = SPIN == 0 ? (q >> 3) * 32 + pipe_rho(q & 7, ak) : op * 32 + pipe_rho(q, ak);
            return *reinterpret_cast<const double2 *>(tm + pipe_tile_idx(c, ai));
        };
#pragma unroll
        for (int h = 0; h < NHB; ++h) {
            double2 aq[PPB];
            if (MF) {
#pragma unroll
                for (int j = 0; j < PF; ++j) aq[j] = a_fetch(h * PPB + j);  // lands while the vector block runs
            }
            // ---- vector block: chain-steps HB h .. HB h + HB - 1 (spin 0: HB / 2 steps of both parity chains) ----
#if HX_PIPE_ABL & 16
            if (MF && RM >= 2) tin = __builtin_amdgcn_s_memtime();
#endif
            double cur[HB];
            if (RM) {
#pragma unroll
                for (int k = 0; k < HB; ++k) {
                    const int kk = HB * h + k;
                    cur[k] = rec_step(ISR{}, IRM{}, SPIN == 0 ? ((kk & 1) ? NCH - 1 : 0) : 0, SPIN == 0 ? kk >> 1 : kk, cq[k]);
                }
            }
            // deferred reduction of the previous flush: operands were read from the staging area one matrix block ago
            if (DEFER && SM == 0 && h == 1 && pend) red_finish_group();
            if (DEFER && SM == 0 && h == 2 && pend) red_finish_quad();
            __builtin_amdgcn_sched_barrier(0);
#if HX_PIPE_ABL & 16
            if (MF && RM >= 2) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); cyc_v += tn_ - tin; tin = tn_; }
#endif
            // ---- matrix block ----
#pragma unroll
            for (int j = 0; j < PPB; ++j) {
                const int sp = h * PPB + j;
                if (MF && j + PF < PPB) aq[j + PF] = a_fetch(sp + PF);
#pragma unroll
                for (int pos = 0; pos < 2; ++pos) {
                    const double a = MF ? (pos ? aq[j].y : aq[j].x) : 0.0;
                    if (MF) acc[0][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[BS][sp][pos][0], acc[0][pos], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (pos == 0) {
                        if (RM >= 2) *reinterpret_cast<double2 *>(tr + pipe_tile_idx(lane, sp)) = make_double2(cur[2 * j], cur[2 * j + 1]);
                    }
                    // coefficients of the next vector block (second ... fourth quarter of this recursion, or the first quarter of
                    // the next stage's): all HB reads in the shadows of the FIRST MFMAs of the matrix block, so that they have
                    // landed when the block ends
                    {
                        constexpr int NQ = 2 * PPB;                    // MFMA shadows of the matrix block
                        constexpr int PER = (HB + NQ / 2 - 1) / (NQ / 2);  // reads per shadow, all within the first half
                        const double2 *src = h + 1 < NHB ? cf_rec + HB * (h + 1) : cf_next;
#pragma unroll
                        for (int u = 0; u < PER; ++u) {
                            const int k = (2 * j + pos) * PER + u;
                            if (k < HB) cq[k] = src[chalf * LBLK + k];
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (MF) {
#pragma unroll
                        for (int g = 1; g < NG; ++g) acc[g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[BS][sp][pos][g], acc[g][pos], 0, 0, 0);
#pragma unroll
                        for (int g = 0; g < NBX; ++g) accx[g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, frx[BS][sp][pos][g], accx[g][pos], 0, 0, 0);
                    }
                    if (DEFER && SM == 0 && pos == 1 && j == PPB - 2 && pend) {
                        if (h == 0) red_issue_group();  // into registers the stored recursion values have left
                        if (h == 1) red_issue_quad();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#if HX_PIPE_ABL & 16
            if (MF && RM >= 2) { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); cyc_m += tn_ - tin; tin = tn_; }
#endif
        }
    };
    // 0: nothing to do, 1: all chains of the set dead, 2: mixed, 3: all live
    auto set_mode = [&](auto SS) __attribute__((always_inline)) {
        constexpr int S = ONESET ? 0 : decltype(SS)::value;
        bool dead = !valid[S] || sc[S][0] < 0, live = !valid[S] || sc[S][0] == 0;
        if (NCH == 2) {
            dead = dead && (!valid[S] || sc[S][NCH - 1] < 0);
            live = live && (!valid[S] || sc[S][NCH - 1] == 0);
        }
        return __all(dead) ? 1 : (__all(live) ? 3 : 2);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>;
    using I3 = std::integral_constant<int, 3>;
    using BT = std::integral_constant<bool, true>;
    using BF = std::integral_constant<bool, false>;
    // dispatch on (MFMA of set SM wanted, recursion mode of the other set)
    auto run_stage = [&](auto SUBB, auto SMM, bool mf, int rm, const double2 *cf_rec, const double2 *cf_next) __attribute__((always_inline)) {
        n_mf = __builtin_amdgcn_readfirstlane(n_mf + (mf ? 1 : 0));
        n_rec = __builtin_amdgcn_readfirstlane(n_rec + (rm ? 1 : 0));
        if (mf) {
            if (rm == 3) stage(SUBB, SMM, BT{}, I3{}, cf_rec, cf_next);
            else if (rm == 2) stage(SUBB, SMM, BT{}, I2{}, cf_rec, cf_next);
            else if (rm == 1) stage(SUBB, SMM, BT{}, I1{}, cf_rec, cf_next);
            else stage(SUBB, SMM, BT{}, I0{}, cf_rec, cf_next);
        } else {
            if (rm == 3) stage(SUBB, SMM, BF{}, I3{}, cf_rec, cf_next);
            else if (rm == 2) stage(SUBB, SMM, BF{}, I2{}, cf_rec, cf_next);
            else if (rm == 1) stage(SUBB, SMM, BF{}, I1{}, cf_rec, cf_next);
            else stage(SUBB, SMM, BF{}, I0{}, cf_rec, cf_next);
        }
    };

    auto kind_of = [](bool mf, int rm) __attribute__((always_inline)) { return mf ? (rm == 3 ? 1 : (rm == 2 ? 0 : 2)) : (rm >= 2 ? 3 : 4); };
    (void)kind_of;
    HX_STAMP(0);
    // ---- prologue: recursion of (set 0, block 0) ----
    bool tl_live[2] = {false, false};  // the tile of set s holds the values of its current block
    {
#pragma unroll
        for (int k = 0; k < HB; ++k) cq[k] = coefs[0][chalf][k];
        const int rm = set_mode(I0{});
        tl_live[0] = rm >= 2 || (HX_PIPE_ABL & 2);
        run_stage(I0{}, I1{}, false, rm, &coefs[0][0][0], &coefs[ONESET ? 1 : 0][0][0]);  // "MFMA set 1" off: only the recursion of set 0
    }
    // Global loads of the hand-over (threads < 128: coefficient doubles; threads 128..159: alpha_l) are issued inside the
    // flush BEFORE the one that stores them to LDS, and IN FRONT of that flush's partial-sum stores: vmcnt retires in order,
    // so a load issued behind the stores could not be waited for without waiting for those stores to reach HBM, and a use
    // right behind the load would expose its latency.  The reduction itself reads alpha_l from LDS: no vector-memory wait
    // inside the flush.  A flush that closes blocks b .. b + NSUB - 1 stores the coefficients of blocks b + NSUB + 1 .. b + 2 NSUB
    // and the alphas of blocks b + NSUB .. b + 2 NSUB - 1.
    double hpre[NSUB];
    auto prefetch = [&](int b0) __attribute__((always_inline)) {  // for the flush that closes blocks b0 ..
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            hpre[u] = 0.0;
            if (tid < 128) hpre[u] = reinterpret_cast<const double *>(coefn + cb + l0 + (b0 + NSUB + 1 + u) * LBLK + coff)[tid & 63];
            else if (tid < 160) hpre[u] = alphan[cb + l0 + (b0 + NSUB + u) * LBLK + (tid - 128)];
        }
    };
    prefetch(0);
    auto block = [&](auto SUBB, int bb) __attribute__((always_inline)) {
        constexpr int SUB = decltype(SUBB)::value;
#pragma unroll
        for (int g = 0; g < NGA; ++g) accs[SUB][g][0] = accs[SUB][g][1] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int g = 0; g < NXA; ++g) accxs[SUB][g][0] = accxs[SUB][g][1] = 0.0;
        const double2 *cf_b = &coefs[bb % NCR][0][0], *cf_n = &coefs[(bb + 1) % NCR][0][0];
        // stage (bb, 0): MFMA(set 0, block bb) || recursion(set 1, block bb)
        {
            const int rm = set_mode(I1{});
            run_stage(SUBB, I0{}, tl_live[0], rm, cf_b, cf_n);
            HX_STAMP(kind_of(tl_live[0], rm));
            tl_live[1] = rm >= 2 || (HX_PIPE_ABL & 2);
        }
        // stage (bb, 1): MFMA(set 1, block bb) || recursion(set 0, block bb + 1)
        {
            const int rm = bb + 1 < nblk ? set_mode(I0{}) : 0;
            run_stage(SUBB, I1{}, tl_live[1], rm, cf_n, cf_n);
            HX_STAMP(kind_of(tl_live[1], rm));
            tl_live[0] = rm >= 2 || (HX_PIPE_ABL & 2);
        }
    };
    // one ring set: stage (t, tile t & 1) = matrix work of l-block t || recursion of l-block t + 1 into the other tile; a pair
    // of stages (l-blocks b, b + 1 -> accumulator sets 0, 1) per flush
    auto block_pair = [&](int bb) __attribute__((always_inline)) {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
            for (int g = 0; g < NGA; ++g) accs[sub][g][0] = accs[sub][g][1] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int g = 0; g < NXA; ++g) accxs[sub][g][0] = accxs[sub][g][1] = 0.0;
        }
        const double2 *cf_1 = &coefs[(bb + 1) % NCR][0][0], *cf_2 = &coefs[(bb + 2) % NCR][0][0];
        {
            const int rm = bb + 1 < nblk ? set_mode(I0{}) : 0;
            run_stage(I0{}, I0{}, tl_live[0], rm, cf_1, cf_2);  // l-block bb from tile A, l-block bb + 1 into tile B
            HX_STAMP(kind_of(tl_live[0], rm));
            tl_live[1] = rm >= 2 || (HX_PIPE_ABL & 2);
        }
        {
            const int rm = bb + 2 < nblk ? set_mode(I0{}) : 0;
            run_stage(std::integral_constant<int, NSUB - 1>{}, I1{}, tl_live[1], rm, cf_2, cf_2);  // l-block bb + 1 from tile B, bb + 2 into tile A
            HX_STAMP(kind_of(tl_live[1], rm));
            tl_live[0] = rm >= 2 || (HX_PIPE_ABL & 2);
        }
    };
    for (int b = 0; b < nblk; b += NSUB) {
        if (ONESET) {
            block_pair(b);
        } else {
            block(I0{}, b);
            if (NSUB > 1 && b + 1 < nblk) block(std::integral_constant<int, NSUB - 1>{}, b + 1);
        }
        if (HX_PIPE_ABL & 4) {
            double chk = 0.0;  // keeps every accumulator alive
#pragma unroll
            for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
                for (int g = 0; g < NGA; ++g)
                    chk += accs[sub][g][0][0] + accs[sub][g][0][1] + accs[sub][g][0][2] + accs[sub][g][0][3] + accs[sub][g][1][0] + accs[sub][g][1][1] + accs[sub][g][1][2] + accs[sub][g][1][3];
#pragma unroll
                for (int g = 0; g < NXA; ++g) chk += accxs[sub][g][0] + accxs[sub][g][1];
            }
            if (chk == 1.2345e-300) A.partial[0] = 1.0;
            continue;
        }
        // ---- flush: D tiles of the 4 waves through tileB (consumed by the stage above), fixed order ----
        // D of v_mfma_f64_16x16x4_f64: row = (lane>>4) + 4 reg, col = lane&15.  Staging of (group g, position p): column-major,
        // 16-byte chunk c = 2 (lane>>4) + (reg>>1) of column col at  (g 2 + p) 256 + col 16 + (c ^ (col & 7)) 2  (128-bit stores,
        // conflict-free through the swizzle);  D of v_mfma_f64_4x4x4_4b: lane (i = lane>>4, blk = (lane>>2)&3, j = lane&3) =
        // row 4 blk + i, column j: both positions in one 16-byte store at  DQ0 + ((row 4 NBX + 4 x + j) 2); sub-block sub at + sub DSZ
        const bool two = NSUB > 1 && b + 1 < nblk;
        if (DEFER) lds_barrier();  // every wave has read the staging area of the previous flush
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            double *dt = DEFER ? dstage + w * DSZ : &tileB[w][0] + sub * DSZ;
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int pos = 0; pos < 2; ++pos)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        *reinterpret_cast<double2 *>(dt + (g * 2 + pos) * 256 + ai * 16 + (((2 * ak + h) ^ (ai & 7)) * 2)) =
                            make_double2(accs[sub][g][pos][2 * h], accs[sub][g][pos][2 * h + 1]);
#pragma unroll
            for (int g = 0; g < NBX; ++g)
                *reinterpret_cast<double2 *>(dt + DQ0 + ((4 * ((lane >> 2) & 3) + ak) * 4 * NBX + 4 * g + (lane & 3)) * 2) =
                    make_double2(accxs[sub][g][0], accxs[sub][g][1]);
        }
#pragma unroll
        for (int u = 0; u < NSUB; ++u) {
            if (tid < 128) (&coefs[(b + NSUB + 1 + u) % NCR][tid >> 6][0].x)[tid & 63] = csign * hpre[u];
            else if (tid < 160) alphas[(b + NSUB + u) % NAR][tid - 128] = hpre[u];
        }
        lds_barrier();
        HX_STAMP(5);
        prefetch(b + NSUB);
        if (DEFER) {
            pend = true;
            pend_slot = b % NAR;
            HX_STAMP(7);
            continue;
        }
        // (rows of a task are padded to whole 32-l blocks: no bounds tests within a block)
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            if (sub > 0 && !two) break;
            const double *alb = alphas[(b + sub) % NAR];
            const double *dt0 = &tileB[0][0] + sub * DSZ;
            const long long rsub = (long long)sub * LBLK * A.pcol;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                double2 s4[NW];
#pragma unroll
                for (int ww = 0; ww < NW; ++ww)
                    s4[ww] = *reinterpret_cast<const double2 *>(dt0 + ww * TBW + (g * 2 + fpos) * 256 + fcol * 16 + ((fch ^ (fcol & 7)) * 2));
                const double sx = (s4[0].x + s4[1].x) + (s4[2].x + s4[3].x), sy = (s4[0].y + s4[1].y) + (s4[2].y + s4[3].y);
                put(pgrp + rsub + g * NCOL, sx * alb[2 * frow + fpos]);
                put(pgrp + rsub + g * NCOL + 8 * (long long)A.pcol, sy * alb[2 * frow + 8 + fpos]);
            }
            if (NBX > 0 && tid < 64 * NBX) {
                double2 s4[NW];
#pragma unroll
                for (int ww = 0; ww < NW; ++ww) s4[ww] = *reinterpret_cast<const double2 *>(dt0 + ww * TBW + DQ0 + tid * 2);
                const double sx = (s4[0].x + s4[1].x) + (s4[2].x + s4[3].x), sy = (s4[0].y + s4[1].y) + (s4[2].y + s4[3].y);
                put(pquad + rsub, sx * alb[2 * qrow]);
                put(pquad + rsub + A.pcol, sy * alb[2 * qrow + 1]);
            }
        }
        pgrp += (long long)NSUB * LBLK * A.pcol;
        pquad += (long long)NSUB * LBLK * A.pcol;
        if (ONESET) {  // coefficients of the first vector block of the next stage pair (recursion of l-block b + 3: stored by this flush)
#pragma unroll
            for (int k = 0; k < HB; ++k) cq[k] = coefs[(b + NSUB + 1) % NCR][chalf][k];
        }
        HX_STAMP(7);
        lds_barrier();  // D tiles consumed: tileB may be overwritten by the recursion of the next stage
        // (taking this barrier inside the next stage instead -- behind its first vector block, in front of its first store to tile B --
        // was measured: 412 vs 401 ms for the 40-column sweep, 117.1 vs 115.5 for 10 spin-0 maps, same device: not kept)
        HX_STAMP(6);
    }
    if (DEFER && pend) {
        red_issue_group();
        red_finish_group();
        red_issue_quad();
        red_finish_quad();
    }
#if HX_PIPE_ABL & 8
    if (lane == 0 && A.counters)
        for (int i = 0; i < 8; ++i) {
            atomicAdd(&A.counters[i], cyc[i]);
            atomicAdd(&A.counters[8 + i], cnt[i]);
        }
    if (lane == 0 && A.counters) {
        atomicAdd(&A.counters[16], cyc_v);
        atomicAdd(&A.counters[17], cyc_m);
    }
#endif
    }  // ring groups of this m
    if ((threadIdx.x & 63) == 0) {
        // a stage = 16 slot pairs x 2 positions x (NG 16x16x4 + NBX 4x4x4_4b) instructions; a recursion stage = 64 lanes x 32 steps x 2 FMAs
        atomicAdd(&g_exec_flops[0], (unsigned long long)n_mf * (32ull * (NG * 2048ull + NBX * 512ull)));
        atomicAdd(&g_exec_flops[1], (unsigned long long)n_rec * (64ull * LBLK * 4ull));
    }
}

// =====================================================================================
// Legendre analysis, TWO independent work-groups per CU ("duo", round 4)
// =====================================================================================
// k_legendre_pipe keeps ONE work-group per CU: whatever one of its waves cannot overlap inside its own instruction stream -- the
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
//   added in place into the rows of this m (as k_legendre_pipe: one work-group per m, ring groups in order).
// One ring set per wave (spin 0: 64 ring pairs, both parity chains in a lane; spin 2: 32 ring pairs x the two functions): a task
// is 8 / 4 ring blocks.
// HALFB (spin 2, HX_HALF_F): the operand of the odd-parity position is the operand of the even one with the four columns of every
// field reversed (see HX_HALF_F): B_odd[k][j] = B_even[k][j ^ 3].  A matrix instruction that takes B_even for an odd-parity row block
// therefore yields that block's rows with the columns of every field reversed -- D_odd[i][j] = (A B_even)[i][j ^ 3] -- which
// k_alm_reduce undoes when it changes the layout.  ONE operand per (slot pair, column group) for both positions: 40 columns (ten
// fields) need 128 operand registers instead of 256 and fit a 256-register wave, with no instruction spent on the permutation.
// (Measured on the way, tools/ubench_mblock.hip: the permutation as two v_mov_b32_dpp per operand costs ~12 cycles of matrix-pipe
// time per VALU instruction -- 417 instead of 357 cycles per slot pair of 40 columns; as two ds_swizzle_b32 it is free.)
#ifndef HX_DUO_FMACDPP
#define HX_DUO_FMACDPP 1  // p' x + q' as (row broadcast, multiply-add with a broadcast source) instead of (two row broadcasts, multiply-add)
#endif
// Gaps in the matrix stream (round 4, second session; tools/ubench_duo.hip, profiles/r04_ubench_duo_gaps.txt).  A wave that streams FP64
// matrix instructions back to back starves the FP64 vector work of the other wave on its SIMD completely; with `s_nop 3` behind every
// (16 x 16 x 4, 4 x 4 x 4) pair and s_setprio 3 on the other wave, exactly two of its FMAs get through per gap, at 7.8 cycles of
// matrix-pipe time each (s_nop 7: 2.8 per gap at 11.8).  In the kernel the other group's recursion and reduction then advance under this
// group's matrix block (cycle accounting per wave-block, ten fields: recursion 4180 -> 3260, reduction 1965 -> 740, second barrier 547 ->
// 208) -- and the matrix block pays for every instruction let in (5390 -> 7850 with s_nop 7): 344 -> 337 ms for ten fields, a wash or a
// loss elsewhere (five fields 196 -> 203, ten spin-0 maps 100 -> 102); s_nop 3 is a gain of ~1 % everywhere.  HX_DUO_GAP = n: s_nop n - 1
// behind every such pair (0: none; -1: 5 for the 40-column shape -- ten fields 344 ms without, 340 / 326 / 328 / 333 / 337 / 347 with n = 4 / 5 / 6 / 7 / 8 / 10 --, 6 for the 36-column one -- nine fields 304 ms without, 310 / 300.5 / 299.6 with n = 4 / 5 / 6 --, none for the 24-column one -- six fields 222 -> 225 ms with n = 3 ... 5 --, 4 otherwise (2 and 3 lose): five fields 196 -> 194, eight 275 -> 272, ten spin-0 maps 100.2 -> 99.2, sixteen 147.6 -> 139.7); HX_DUO_PRIO: s_setprio 3 outside the matrix block.
#ifndef HX_DUO_GAP
#define HX_DUO_GAP -1
#endif
#ifndef HX_DUO_PRIO
#define HX_DUO_PRIO 1
#endif
#ifndef HX_DUO_ORDER
#define HX_DUO_ORDER -1  // -1: per shape (GORDER)
#endif
// Step of the scaled recursion chains of k_legendre_duo: value = v 2^(SB e), live (in the sums) from 2^-SB on.  Spin 2 runs with 100: the
// matrix work of 4 % of its blocks goes away (ten fields 321 -> 312 ms, results bit-identical: what is left out is below 2^-75 of a value
// of lambda).  Spin 0 stays at 300: the same rule saves 3.9 % of its matrix instructions and no time (97.3 -> 98.4 ms: two chains per lane
// to test in a kernel that already spills).
#ifndef HX_DUO_SCALE_BITS2
#define HX_DUO_SCALE_BITS2 100
#endif
#ifndef HX_DUO_SCALE_BITS0
#define HX_DUO_SCALE_BITS0 300
#endif

#ifndef HX_DUO_ABL
#define HX_DUO_ABL 0  // timing experiments only: 1 no matrix instructions, 2 no recursion, 4 no flush, 8 plain stores instead of atomics, 32 cycle accounting
#endif
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
                if (SPIN == 2 && HX_HALF_F && ((l + m) & 1)) {  // odd-parity rows carry the signs (-, +, +, -) on (E_re, E_im, B_re, B_im)
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

#if HX_DUO_ABL & 128  // diagnostic build: when and where the work-group of every m ran (100 MHz wall clock, XCC, HW_ID)
__device__ unsigned long long g_duo_stamp[8192 * 3];
#endif
// NSUB: 32-l blocks per flush (their D tiles stay in registers; one staging round and one pair of barriers per NSUB blocks)
template <int SPIN, int NG, int NBX, int NSUB = 1>
__global__ __launch_bounds__(256, 2) void k_legendre_duo(LegParams A, const double2 *__restrict__ coefn)
{
    using C = PipeCfg<SPIN>;
    constexpr int NW = 4, NOP = C::NOP, NCH = C::NCH, NPAIR = 16;
    constexpr int NXA = NBX > 0 ? NBX : 1;
    constexpr int DQ0 = NG * 512, DSZ = NG * 512 + NBX * 128;
    constexpr bool HALFB = SPIN == 2;
    constexpr int SB = SPIN == 2 ? HX_DUO_SCALE_BITS2 : HX_DUO_SCALE_BITS0;  // scaled chains: value = v 2^(SB e), live from 2^-SB on (sval_rebase)
    constexpr int GAPN = HX_DUO_GAP >= 0 ? HX_DUO_GAP : ((SPIN == 2 && NG == 2 && NBX == 2) ? 5 : (SPIN == 2 && NG == 2 && NBX == 1) ? 6 : (SPIN == 2 && NG == 1 && NBX == 2) ? 0 : 4);
    // where the gaps stand: 0 behind every (16 x 16 x 4, 4 x 4 x 4) pair; 1 between the two instructions of a pair (ten spin-0 maps 99.4 ->
    // 98.1 ms; ten fields 344: not there); 2 behind the 16 x 16 x 4 and behind the 4 x 4 x 4 instructions of a position (ten fields 326.5 -> 324)
    constexpr int GORDER = HX_DUO_ORDER >= 0 ? HX_DUO_ORDER : ((SPIN == 2 && NG == 2 && NBX == 2) ? 2 : (SPIN == 0 && NG == 1 && NBX == 1) ? 1 : 0);
    constexpr int NPB = HALFB ? 1 : 2;  // operand positions kept in registers
    // doubles per wave of its tile: the 16 x 64 lambda tile (2048), or the D tiles of a flush if they need more (two blocks of 36 / 40
    // columns: 2304 / 2560 -- 80 KiB per work-group, still two per CU)
    constexpr int TW = NSUB * DSZ > 2048 ? NSUB * DSZ : 2048;
    static_assert(NG >= 1 && NG <= 2 && TW * NW * 8 <= 81920, "two work-groups per CU");
    static_assert(SPIN == 0 || HX_HALF_F, "spin 2: the lambda- chain carries (-1)^(l + m) lambda- (one operand row, wave-uniform coefficients)");
    __shared__ double tile[NW][TW];            // 64 KiB (80 at most); doubles as the D staging area of the flush
    __shared__ int lead_in[2][NW];             // while no wave of the ring group has issued a matrix instruction yet: did wave w, in flush (slot) b?
#if HX_DUO_ABL & 64  // diagnostic: ONE work-group per CU (the phase durations of a wave that is alone on its SIMD)
    __shared__ double lds_hog[3072];
    if (A.ncol < 0) lds_hog[threadIdx.x] = 1.0;
    if (A.ncol < -1) A.partial[1] = lds_hog[threadIdx.x ^ 1];
#endif
    const PlanDev &P = A.P;
    const int m = A.m0 + blockIdx.x * A.ms, lmax = P.lmax;
#if HX_DUO_ABL & 128
    const unsigned long long stamp_begin = wall_clock64();
#endif
    const MTasks mt = A.of_m[m];
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const int off = (l0 + m) & 1;
    const int nblk = (lmax - l0) / LBLK + 1;
    const long long orow = A.arow[m] - A.arow0;
    // recursion coefficients (p', q') of this m: WAVE-UNIFORM (spin 2: the lambda- chain, upper half of the wave, runs with (-p', +q'), i.e.
    // with the same coefficients and -x: HX_HALF_F).  Lane j of every 16-lane row loads the pair of step j (one 16-byte load per 16 steps
    // and wave, requested a block ahead) and a step takes its pair from there with two row broadcasts (v_mov_b64_dpp row_newbcast) -- no
    // LDS hand-over, no LDS read per step (32 of the 48 LDS instructions of a block's recursion: 1760 -> cycles per block for a lone
    // wave), no scalar load (their latency cannot be covered: out-of-order return allows no load in flight across a wait: 5000 cycles)
    const double2 *__restrict__ cfm = coefn + almidx(lmax, 0, m) + l0 + (SPIN == 0 ? 0 : 1) + (threadIdx.x & 15);
    // the FIRST ring group of an m stores its rows, the others add to them: the rows need not be zeroed before the launch (3 GB per sweep of
    // ten fields) and a fortieth of the atomics is plain stores
    bool first_group = true;
    auto put = [&](double *p, double v) __attribute__((always_inline)) {
#if HX_DUO_ABL & 8
        *p = v;
#else
        if (first_group) *p = v;
        else __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)p, v);
#endif
    };
    auto lds_barrier = []() __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): global atomics in flight do not hold the barrier
        __builtin_amdgcn_s_barrier();
    };
    int n_mf = 0, n_rec = 0;
    if (HX_DUO_PRIO) __builtin_amdgcn_s_setprio(3);
#if HX_DUO_ABL & 32
    // cycle accounting (diagnostic build): [0] task prologue, [1] recursion (live / mixed), [2] recursion (dead), [3] matrix block, [4] staging + wait at the
    // first barrier, [5] reduction, [6] wait at the second barrier
    unsigned long long cyc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#define DUO_STAMP(i) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); cyc[i] += tn_ - tlast; tlast = tn_; } while (0)
#else
#define DUO_STAMP(i) do { } while (0)
#endif
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
#if HX_DUO_FMACDPP
#define HX_BC(K) case K: tq = row_bcast_fmac<K>(row_bcast<K>(cl[kk >> 4].y), cl[kk >> 4].x, xx); break;
#else
#define HX_BC(K) case K: tq = fma(row_bcast<K>(cl[kk >> 4].x), xx, row_bcast<K>(cl[kk >> 4].y)); break;
#endif
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

        // flush roles (as k_legendre_pipe): 16-byte chunk (col, c) of a group tile = rows (c >> 1) + 8 (c & 1) and + 4
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
        DUO_STAMP(0);
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
                int rm = (HX_DUO_ABL & 2) ? 3 : set_mode(b);
                const double2 cl[2] = {cnext[0], cnext[1]};
                if (HX_DUO_ABL & 2) {
                } else if (rm == 3) recursion(I3{}, cl);
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
                if (rm >= 2) DUO_STAMP(1); else DUO_STAMP(2);
                mine = mine || rm >= 2;
                if (rm >= 2 && !(HX_DUO_ABL & 1)) {
                    n_mf = __builtin_amdgcn_readfirstlane(n_mf + 1);
                    if (HX_DUO_PRIO) __builtin_amdgcn_s_setprio(0);
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
                                // one 16 x 16 x 4 (+ one 4 x 4 x 4) instruction, then a gap of a few cycles: see HX_DUO_GAP
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
                    if (HX_DUO_PRIO) __builtin_amdgcn_s_setprio(3);
                }
                DUO_STAMP(3);
            }
            if (HX_DUO_ABL & 4) {
                double chk = 0.0;
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) {
#pragma unroll
                    for (int g = 0; g < NG; ++g) chk += acc[sub][g][0][0] + acc[sub][g][0][1] + acc[sub][g][0][2] + acc[sub][g][0][3] + acc[sub][g][1][0] + acc[sub][g][1][1] + acc[sub][g][1][2] + acc[sub][g][1][3];
#pragma unroll
                    for (int g = 0; g < NXA; ++g) chk += accx[sub][g][0] + accx[sub][g][1];
                }
                if (chk == 1.2345e-300) A.partial[0] = 1.0;
                continue;
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
            DUO_STAMP(4);
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
            DUO_STAMP(5);
            lds_barrier();  // D tiles consumed: the tiles may be overwritten by the next block's recursion
            DUO_STAMP(6);
        }
    }  // ring groups of this m
#if HX_DUO_ABL & 32
    if ((threadIdx.x & 63) == 0 && A.counters)
        for (int i = 0; i < 8; ++i) atomicAdd(&A.counters[i], cyc[i]);
    if ((threadIdx.x & 63) == 0 && A.counters) { atomicAdd(&A.counters[8], (unsigned long long)n_rec); atomicAdd(&A.counters[9], (unsigned long long)n_mf); }
#endif
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&g_exec_flops[0], (unsigned long long)n_mf * (32ull * (NG * 2048ull + NBX * 512ull)));
        atomicAdd(&g_exec_flops[1], (unsigned long long)n_rec * (64ull * LBLK * 4ull));
    }
#if HX_DUO_ABL & 128
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        unsigned xcc, hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        g_duo_stamp[blockIdx.x * 3 + 0] = stamp_begin;
        g_duo_stamp[blockIdx.x * 3 + 1] = wall_clock64();
        g_duo_stamp[blockIdx.x * 3 + 2] = ((unsigned long long)xcc << 32) | hwid;
    }
#endif
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
// k_legendre_duo (two work-groups per CU, round 4) runs every sweep of the matrix unit; HX_LEG_KERNEL=pipe restores k_legendre_pipe
// (one work-group per CU, rounds 2-3): A/B switch
static bool leg_duo()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("HX_LEG_KERNEL");
        v = (e && !strcmp(e, "pipe")) ? 0 : 1;
    }
    return v == 1;
}
// HX_PIPE_ONESET=0 keeps ten spin-2 fields as two sweeps of five (the kernel of the first half of round 2): A/B switch
static bool oneset_enabled()
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("HX_PIPE_ONESET");
        v = (e && atoi(e) == 0) ? 0 : 1;
    }
    return v == 1;
}
static SweepShape sweep_shape(int spin, int nb)
{
    const int cols = 2 * nb;
    SweepShape sh = {0, 0, 0, 0, 0, 0};
    if (leg_duo()) {
        // one ring set per wave; spin 2 keeps one operand of the two positions in registers (HALFB): up to two 16-column groups + two
        // 4-column blocks (ten fields); spin 0 one group + one block (ten maps) or two groups (sixteen)
        if (cols <= 8) { sh.valu = 1; return sh; }
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
    if (spin == 2 && cols > 2 * NCOL && cols <= 2 * NCOL + 8 && oneset_enabled()) {  // 9 or 10 fields: two groups + one or two blocks, one ring set per wave
        sh.ng = 2;
        sh.nbx = (cols - 2 * NCOL) / 4;
        sh.oneset = 1;
        sh.ncol = cols;
        return sh;
    }
    if (cols <= 8) {  // <= 4 spin-0 maps / <= 2 spin-2 fields: one sweep per map / field on the vector unit (hx_legendre_valu.hip)
        sh.valu = 1;
        return sh;
    }
    sh.ng = cols / NCOL;
    const int rem = cols % NCOL;
    if (sh.ng == 0 || rem > 4 || (sh.ng == 2 && rem > 0)) {  // two extra blocks next to a full group spill (172 VGPRs)
        sh.ng += 1;  // a partly filled group
    } else {
        sh.nbx = (rem + 3) / 4;
    }
    sh.ncol = NCOL * sh.ng + 4 * sh.nbx;
    return sh;
}
int analysis_max_comp(int spin) { return (spin == 2 && (leg_duo() || oneset_enabled())) ? 8 * NGMAX + 4 : 8 * NGMAX; }

// Components of the next sweep when `remaining` are left.  Resident inputs: the split that costs least by the measured sweep times
// (ms at nside 4096 / lmax 6144; only their ratios matter): a sweep pays for its PADDED columns and ~76 ms of recursion, flush and
// dead stages whatever it holds, so 13 spin-2 fields are 8 + 5 (two full shapes: 316 + 224) rather than 7 + 6 (two padded
// 32-column sweeps: 632), 26 spin-0 maps 16 + 10 rather than 13 + 13, ten fields one 40-column sweep, and one or two left-over
// maps / fields go to the vector-unit kernel.  Ties take the larger sweep first.
static double sweep_cost(int spin, int units)
{
    if (leg_duo()) {  // k_legendre_duo, round 4, with the gaps of HX_DUO_GAP (gpurun_out/ab_gap3.txt, ab_gap4.txt)
        if (spin == 0) return units <= 4 ? 21.6 * units : (units <= 8 ? 84.1 : (units <= 10 ? 99.2 : 139.7));
        return units <= 2 ? 61.0 * units : (units <= 4 ? 164.5 : (units == 5 ? 194.1 : (units == 6 ? 222.0 : (units <= 8 ? 272.0 : (units == 9 ? 299.6 : 326.0)))));
    }
    if (spin == 0) return units <= 4 ? 21.6 * units : (units <= 8 ? 100.0 : (units <= 10 ? 115.0 : 162.0));
    return units <= 2 ? 61.0 * units : (units <= 4 ? 196.0 : (units == 5 ? 224.0 : (units <= 8 ? 316.0 : (units == 9 ? 360.0 : 400.0))));
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

// 32-l blocks per flush of the pipelined kernel: one (the reduction of a flush then rides in the next block's first stage);
// HX_PIPE_NSUB = 2 selects the two-block flush through the wave's second tile (a tuning knob, read once: every partial
// row is the same fixed-order sum either way, the results are bit-identical)
static int pipe_nsub(int spin)
{
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("HX_PIPE_NSUB");
        v = e ? atoi(e) : 0;
    }
    (void)spin;
    return v == 2 ? 2 : 1;
}

static bool duo_shape(const SweepShape &sh) { return sh.duo != 0; }
// doubles per accumulation row of a sweep (rows of the one-ring-set kernel start on 128-byte lines: the atomics of a flush -- 16 lanes
// x 8 B per row and column group -- then touch whole aligned lines instead of straddling two: -10 ms for its 40-column sweep, nothing
// for the other shapes of k_legendre_pipe)
static int sweep_pcol(const SweepShape &sh)
{
    if (duo_shape(sh)) {
        static int pad = -1;
        if (pad < 0) { const char *e = getenv("HX_DUO_PCOL"); pad = e ? atoi(e) : 0; }  // (experiments)
        if (pad > 0) return (sh.ncol + pad - 1) / pad * pad;
        return sh.ncol > 2 * NCOL ? (sh.ncol + 15) / 16 * 16 : sh.ncol;
    }
    if (sh.oneset) return (sh.ncol + 15) / 16 * 16;
    return sh.ncol;
}

// k_legendre_duo for a sweep shape (one work-group per order of the grid)
template <int SPIN>
static int launch_duo(const SweepShape &sh, dim3 pgrid, hipStream_t st, const LegParams &A, const double2 *cn)
{
    const dim3 db(256);
    const int key = sh.ng * 10 + sh.nbx;
    // two l-blocks per flush wherever the second accumulator set fits the 256 registers (ten spin-0 maps 105 -> 101 ms, five spin-2 fields
    // 206 -> 198, eight 289 -> 276, nine 320 -> 304; sixteen spin-0 maps spill: 147 -> 157; ten fields spill 43 registers: 345 -> 423);
    // HX_DUO_NSUB=1 keeps one block per flush (bit-identical results)
    static int nsub1 = -1;
    if (nsub1 < 0) { const char *e = getenv("HX_DUO_NSUB"); nsub1 = (e && atoi(e) == 1) ? 1 : 0; }
    if (key == 10 && !nsub1) hipLaunchKernelGGL((k_legendre_duo<SPIN, 1, 0, 2>), pgrid, db, 0, st, A, cn);
    else if (key == 11 && !nsub1) hipLaunchKernelGGL((k_legendre_duo<SPIN, 1, 1, 2>), pgrid, db, 0, st, A, cn);
    else if (key == 20 && !nsub1 && SPIN == 2) hipLaunchKernelGGL((k_legendre_duo<2, 2, 0, 2>), pgrid, db, 0, st, A, cn);
    else if (key == 12 && !nsub1 && SPIN == 2) hipLaunchKernelGGL((k_legendre_duo<2, 1, 2, 2>), pgrid, db, 0, st, A, cn);
    else if (key == 21 && !nsub1 && SPIN == 2) hipLaunchKernelGGL((k_legendre_duo<2, 2, 1, 2>), pgrid, db, 0, st, A, cn);  // (its D tiles need 72 KiB of LDS)
    else if (key == 10) hipLaunchKernelGGL((k_legendre_duo<SPIN, 1, 0>), pgrid, db, 0, st, A, cn);
    else if (key == 11) hipLaunchKernelGGL((k_legendre_duo<SPIN, 1, 1>), pgrid, db, 0, st, A, cn);
    else if (key == 20) hipLaunchKernelGGL((k_legendre_duo<SPIN, 2, 0>), pgrid, db, 0, st, A, cn);
    else if (key == 12 && SPIN == 2) hipLaunchKernelGGL((k_legendre_duo<2, 1, 2>), pgrid, db, 0, st, A, cn);
    else if (key == 21 && SPIN == 2) hipLaunchKernelGGL((k_legendre_duo<2, 2, 1>), pgrid, db, 0, st, A, cn);
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
        A.counters = nullptr;
        A.of_m = ts.d_of_m.as<MTasks>(); A.arow = ts.d_arow.as<long long>(); A.arow0 = ts.arow[m0]; A.add_all = 0;
        A.tasks = ts.d_tasks.as<LegTask>();  // the pipelined kernel indexes the whole list through of_m
        if (!duo_shape(sh))  // (k_legendre_duo's first ring group of an m stores its rows)
            HX_HIP(hipMemsetAsync(pl->partial.p, 0, (size_t)(ts.arow[m1] - ts.arow[m0]) * pcol * sizeof(double), st));
#if defined(HX_DIAG) && ((HX_PIPE_ABL & 8) || (HX_DUO_ABL & 32))
        HX_TRY(pl->d_dbg.alloc(144));
        HX_HIP(hipMemsetAsync(pl->d_dbg.p, 0, 144, st));
        A.counters = pl->d_dbg.as<unsigned long long>();
#endif
        const double2 *cn = SPIN == 0 ? pl->cn0.as<double2>() : pl->cn2.as<double2>();
        const double *al = SPIN == 0 ? pl->al0.as<double>() : pl->al2.as<double>();
        dim3 pblock(PipeCfg<SPIN>::NW * 64), pgrid((unsigned)nm);
        if (duo_shape(sh)) {
            HX_TRY(launch_duo<SPIN>(sh, pgrid, st, A, cn));
        }
        else if (sh.ng == 1 && sh.nbx == 0 && pipe_nsub(SPIN) == 2)
            hipLaunchKernelGGL((k_legendre_pipe<SPIN, 1, 0, 2>), pgrid, pblock, 0, st, A, cn, al);
        else if (sh.ng == 1 && sh.nbx == 0)
            hipLaunchKernelGGL((k_legendre_pipe<SPIN, 1, 0, 1>), pgrid, pblock, 0, st, A, cn, al);
        else if (sh.ng == 1 && sh.nbx == 1 && pipe_nsub(SPIN) == 2)
            hipLaunchKernelGGL((k_legendre_pipe<SPIN, 1, 1, 2>), pgrid, pblock, 0, st, A, cn, al);
        else if (sh.ng == 1 && sh.nbx == 1)
            hipLaunchKernelGGL((k_legendre_pipe<SPIN, 1, 1, 1>), pgrid, pblock, 0, st, A, cn, al);
        else if (sh.oneset && SPIN == 2 && sh.nbx == 2)
            hipLaunchKernelGGL((k_legendre_pipe<2, 2, 2, 2, 1>), pgrid, pblock, 0, st, A, cn, al);
        else if (sh.oneset && SPIN == 2 && sh.nbx == 1)
            hipLaunchKernelGGL((k_legendre_pipe<2, 2, 1, 2, 1>), pgrid, pblock, 0, st, A, cn, al);
        else if (sh.ng == 2 && sh.nbx == 0)
            hipLaunchKernelGGL((k_legendre_pipe<SPIN, 2, 0, 1>), pgrid, pblock, 0, st, A, cn, al);  // a second accumulator set spills 20-76 registers
        else
            return fail(HX_ERR_ARG, "legendre analysis: no kernel for %d groups + %d blocks", sh.ng, sh.nbx);
#if defined(HX_DIAG) && (HX_DUO_ABL & 128)
        if (const char *fn = duo_shape(sh) ? getenv("HX_DUO_STAMP_FILE") : nullptr) {
            HX_HIP(hipStreamSynchronize(st));
            std::vector<unsigned long long> h(8192 * 3);
            HX_HIP(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_duo_stamp), sizeof(unsigned long long) * h.size()));
            char name[512];
            snprintf(name, sizeof name, "%s.spin%d", fn, SPIN);
            if (FILE *f = fopen(name, "w")) {
                for (int k = 0; k < std::min(nm, 8192); ++k)
                    fprintf(f, "%d %llu %llu %llu %llu %d\n", m0 + k * ms, h[3 * k], h[3 * k + 1], h[3 * k + 2] >> 32, h[3 * k + 2] & 0xffffffffull, ts.of_m[m0 + k * ms].count);
                fclose(f);
            }
        }
#endif
#if defined(HX_DIAG) && (HX_DUO_ABL & 32)
        if (duo_shape(sh)) {
            unsigned long long hc[18];
            HX_HIP(hipStreamSynchronize(st));
            HX_HIP(hipMemcpy(hc, pl->d_dbg.p, 144, hipMemcpyDeviceToHost));
            const char *nm[7] = {"prologue", "rec live", "rec dead", "matrix", "stage+bar1", "reduce", "bar2"};
            double tot = 0;
            for (int i = 0; i < 7; ++i) tot += (double)hc[i];
            for (int i = 0; i < 7; ++i)
                fprintf(stderr, "[hx] duo spin %d: %-11s %5.1f %%  %8.1f cycles per wave-block\n", SPIN, nm[i], 100.0 * hc[i] / tot, (double)hc[i] / (double)hc[8]);
            fprintf(stderr, "[hx] duo spin %d: %llu wave-blocks, %llu with matrix work, %.1f cycles per wave-block in all\n", SPIN, hc[8], hc[9], tot / (double)hc[8]);
        }
#endif
#if defined(HX_DIAG) && (HX_PIPE_ABL & 8)
        {
            unsigned long long hc[18];
            HX_HIP(hipStreamSynchronize(st));
            HX_HIP(hipMemcpy(hc, pl->d_dbg.p, 144, hipMemcpyDeviceToHost));
            if (hc[9]) fprintf(stderr, "[hx] pipe spin %d: per mfma||rec stage: vector blocks %.1f cycles, matrix blocks %.1f\n", SPIN, (double)hc[16] / hc[9], (double)hc[17] / hc[9]);
            const char *nm[8] = {"mfma||mixed", "mfma||live", "mfma||dead", "rec alone", "dead alone", "flush->bar1", "bar2 wait", "reduce"};
            double tot = 0;
            for (int i = 0; i < 8; ++i) tot += (double)hc[i];
            for (int i = 0; i < 8; ++i)
                fprintf(stderr, "[hx] pipe spin %d m [%d,%d): %-13s %5.1f %%  %12llu intervals, %8.1f cycles each\n", SPIN, m0, m1, nm[i],
                        100.0 * hc[i] / tot, hc[8 + i], hc[8 + i] ? (double)hc[i] / hc[8 + i] : 0.0);
        }
#endif
    }
    {
        ProfScope ps("alm_reduce");
        hipLaunchKernelGGL(k_alm_reduce<SPIN>, dim3(nm), dim3(256), 0, st, P, ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(),
                           pl->partial.as<double>(), ts.arow[m0], m0, ms, nb, ng, pcol, d_fl, add, d_alms, pl->nlm, ts.d_arow.as<long long>(),
                           duo_shape(sh) ? (SPIN == 0 ? pl->al0.as<double>() : pl->al2.as<double>()) : (const double *)nullptr);
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
        A.counters = nullptr;
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
