// hx_legendre_valu.hip -- Legendre / Wigner-d analysis of ONE map (spin 0) or ONE (Q,U) field (spin 2) on the FP64 vector unit.
//
// The call the reference itself makes is a single-map transform (heracles/mapping.py:171 -> heracles/healpy.py:183-189: one
// hp.map2alm per (field, bin)).  With two (spin 0) or four (spin 2) real columns there is nothing for a matrix instruction to
// contract against: the tile route of hx_analysis.hip (recursion -> LDS tile -> A operand of an MFMA) moves every value of
// lambda_lm(theta) through LDS once each way and is bound by LDS bandwidth at ~25 ms per spin-0 sweep at nside 4096 whatever the
// number of columns, and the 4x4x4 instruction spends as long on one map as on two.  Here the sum over rings stays on the vector
// unit, which runs the recursion anyway:
//   lane     = ring pair, R ring pairs per lane (a task = 64 R consecutive ring pairs of one m, a single wave);
//   per (ring pair, l): 2 FMAs of the normalised recursion + 2 (spin 0) / 8 (spin 2) FMAs into per-lane accumulators indexed by
//              l -- SURVEY 8d's F0 = 8 flop per (ring pair, l, m) and 3 F0 exactly, nothing else in the inner loop;
//   per block of LB l:  the NA * LB = 32 accumulators of the 64 lanes are summed by a transposed butterfly (v_permlane32_swap /
//              v_permlane16_swap of gfx950, then DPP moves inside the row): 124 vector instructions per block against
//              R * LB * (2 + NA) * ... of recursion and accumulation -- about a quarter on top;
//   no LDS traffic beyond the 16 coefficient pairs of a block, no matrix instruction, no work-group barrier that waits for
//   another wave: four independent waves per CU (two per SIMD when the registers allow).
// Chains below 2^-300 are carried with an exponent (as in k_legendre_pipe); a wave-uniform mode per (ring slot, block) picks
// dead (step only) / mixed / live (no exponent bookkeeping) code.
// Output: one span of rows per task (ring group) in `partial`, summed over the tasks of an m in fixed order by k_alm_reduce:
// bitwise repeatable.
#include <algorithm>
#include <cmath>

#include "hx_sht_common.h"

namespace hx {
using namespace hxfft;

// tuning knobs (tools/build_valu_variants.sh)
constexpr int VALU_R0 = 8;
constexpr int VALU_R2 = 6;
constexpr int VALU_LB0 = 8;
constexpr int VALU_LB2 = 4;
template <int SPIN>
struct ValuCfg {
    static constexpr int R = SPIN == 0 ? VALU_R0 : VALU_R2;     // ring pairs per lane
    static constexpr int LB = SPIN == 0 ? VALU_LB0 : VALU_LB2;  // l values per accumulator block (even)
    static constexpr int NA = SPIN == 0 ? 2 : 4;    // accumulators per l: (re, im) / (G_re, G_im, K_re, K_im)
    static constexpr int NF = SPIN == 0 ? 4 : 8;    // operand doubles per (m, ring pair)
    static constexpr int NRB = 2 * R;               // 32-ring-pair blocks per task
    static_assert(LB * NA == 32 || LB * NA == 16, "the butterfly below reduces 16 or 32 accumulators");
};

int valu_task_blocks(int spin) { return spin == 0 ? ValuCfg<0>::NRB : ValuCfg<2>::NRB; }
int valu_partial_cols(int spin) { return spin == 0 ? ValuCfg<0>::NA : ValuCfg<2>::NA; }
int valu_operand_doubles(int spin) { return spin == 0 ? ValuCfg<0>::NF : ValuCfg<2>::NF; }

// FP64 vector flops this TU's kernels execute (one atomic per wave at its end): see g_exec_flops in hx_analysis.hip
__device__ unsigned long long g_valu_flops;

struct ValuParams {
    PlanDev P;
    const LegTask *__restrict__ tasks;  // tasks [t0, t1) of the m-chunk
    const double *__restrict__ F;       // [(m - m0) / ms][rp][NF]
    double *__restrict__ partial;       // [row - row0][NA]
    int m0, ms;                         // the chunk holds the orders m0 + k ms (tasks of other orders in the list return at once)
    long long row0;
};

// =====================================================================================
// Y -> operands of one map / field:  F[m - m0][rp][NF]
//   spin 0: (s_re, s_im, d_re, d_im), s = F_N + F_S (l + m even), d = F_N - F_S (odd)
//   spin 2: (U, V, U', V') complex with U = P+_N, V = P-_S, U' = P-_N, V' = P+_S, P+- = -(Q +- iU)/2:
//           with G = sum_rings lambda+ U +- lambda- V', K = sum_rings lambda- U' +- lambda+ V  (+ for l + m even, - for odd)
//           E = G + K, B = -i (G - K)  (the columns B+(P) = [Pr, Pi, Pi, -Pr], B-(P) = [Pr, Pi, -Pi, Pr] of k_fourier_combine)
// grid: x = m - m0, y = blocks of 256 ring pairs; one thread per ring pair.
// =====================================================================================
template <int SPIN>
__global__ __launch_bounds__(256) void k_fourier_combine_valu(PlanDev P, const double2 *__restrict__ Y, int c0, int m0, int ms,
                                                              const double *__restrict__ rw, const LegTask *__restrict__ tasks,
                                                              const MTasks *__restrict__ of_m, double *__restrict__ F)
{
    constexpr int NF = ValuCfg<SPIN>::NF;
    const int m = m0 + blockIdx.x * ms;
    const MTasks mt = of_m[m];
    if (mt.count == 0 || ((int)blockIdx.y + 1) * 256 <= tasks[mt.first].rb0 * RBLK) return;  // pruned ring pairs: never read
    const int rp = blockIdx.y * 256 + threadIdx.x;
    if (rp >= P.nrp_pad) return;
    double *row = F + ((long long)blockIdx.x * P.nrp_pad + rp) * NF;
    double o[NF];
#pragma unroll
    for (int k = 0; k < NF; ++k) o[k] = 0.0;
    if (rp < P.nrp) {
        const RingAtM ram = ring_at_m_of(P, rp, m, rw);
        if (SPIN == 0) {
            double2 fn, fs;
            ring_modes_ns(P, Y, c0, rp, m, ram, fn, fs);
            o[0] = fn.x + fs.x; o[1] = fn.y + fs.y;
            o[2] = fn.x - fs.x; o[3] = fn.y - fs.y;
        } else {
            double2 qn, qs, un, us;
            ring_modes_ns(P, Y, c0, rp, m, ram, qn, qs);
            ring_modes_ns(P, Y, c0 + 1, rp, m, ram, un, us);
            const double2 ppn = cscale(cadd(qn, mul_pi(un)), -0.5), pmn = cscale(csub(qn, mul_pi(un)), -0.5);
            const double2 pps = cscale(cadd(qs, mul_pi(us)), -0.5), pms = cscale(csub(qs, mul_pi(us)), -0.5);
            o[0] = ppn.x; o[1] = ppn.y; o[2] = pms.x; o[3] = pms.y;
            o[4] = pmn.x; o[5] = pmn.y; o[6] = pps.x; o[7] = pps.y;
        }
    }
#pragma unroll
    for (int k = 0; k < NF; k += 2) *reinterpret_cast<double2 *>(row + k) = make_double2(o[k], o[k + 1]);
}

// =====================================================================================
// cross-lane sum of 32 accumulators: afterwards lane L holds the sum over the 64 lanes of v[L >> 1] (in v[0])
// =====================================================================================
__device__ __forceinline__ double dpp_f64(double old, double src, const int ctrl_sel)
{
    // ctrl_sel: 0 row_ror:8 banks 0,1 | 1 row_ror:8 banks 2,3 | 2 identity banks 2,3 | 3 half-mirror banks 0,2 | 4 half-mirror banks 1,3
    //           5 identity banks 1,3 | 6 quad_perm [2,3,0,1] | 7 quad_perm [1,0,3,2]
    int lo = __double2loint(src), hi = __double2hiint(src), olo = __double2loint(old), ohi = __double2hiint(old);
    switch (ctrl_sel) {
    case 0: lo = __builtin_amdgcn_update_dpp(olo, lo, 0x128, 0xF, 0x3, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0x128, 0xF, 0x3, false); break;
    case 1: lo = __builtin_amdgcn_update_dpp(olo, lo, 0x128, 0xF, 0xC, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0x128, 0xF, 0xC, false); break;
    case 2: lo = __builtin_amdgcn_update_dpp(olo, lo, 0xE4, 0xF, 0xC, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0xE4, 0xF, 0xC, false); break;
    case 3: lo = __builtin_amdgcn_update_dpp(olo, lo, 0x141, 0xF, 0x5, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0x141, 0xF, 0x5, false); break;
    case 4: lo = __builtin_amdgcn_update_dpp(olo, lo, 0x141, 0xF, 0xA, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0x141, 0xF, 0xA, false); break;
    case 5: lo = __builtin_amdgcn_update_dpp(olo, lo, 0xE4, 0xF, 0xA, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0xE4, 0xF, 0xA, false); break;
    case 6: lo = __builtin_amdgcn_update_dpp(olo, lo, 0x4E, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0x4E, 0xF, 0xF, false); break;
    default: lo = __builtin_amdgcn_update_dpp(olo, lo, 0xB1, 0xF, 0xF, false); hi = __builtin_amdgcn_update_dpp(ohi, hi, 0xB1, 0xF, 0xF, false); break;
    }
    return __hiloint2double(hi, lo);
}

// v[i] (i < N, N = 16 or 32) per lane -> v[0] of lane L = sum over all 64 lanes of v[L >> SH], SH = log2(64 / N).  Every halving
// step pairs value i with value i + h and lane L with a partner lane: the lanes whose step bit is clear keep the sum of value i
// over the pair, the others that of value i + h.  Steps: lane bit 5 (v_permlane32_swap: upper half of the first operand <->
// lower half of the second), bit 4 (v_permlane16_swap: odd rows of the first <-> even rows of the second), bit 3 (row_ror:8),
// bit 2 (row_half_mirror: j <-> 7 - j), bit 1 (quad_perm [2,3,0,1]); the lane bits that are left are plain sums.  Fixed
// association: bitwise repeatable.
template <int N>
__device__ __forceinline__ void wave_reduce(double (&v)[N], int lane)
{
    static_assert(N == 16 || N == 32, "16 or 32 accumulators");
    constexpr int H1 = N / 2, H2 = N / 4, H3 = N / 8, H4 = N / 16;
#pragma unroll
    for (int i = 0; i < H1; ++i) {
        const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v[i]), (unsigned)__double2loint(v[i + H1]), false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v[i]), (unsigned)__double2hiint(v[i + H1]), false, false);
        v[i] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
#pragma unroll
    for (int i = 0; i < H2; ++i) {
        const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v[i]), (unsigned)__double2loint(v[i + H2]), false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v[i]), (unsigned)__double2hiint(v[i + H2]), false, false);
        v[i] = __hiloint2double((int)hi[0], (int)lo[0]) + __hiloint2double((int)hi[1], (int)lo[1]);
    }
#pragma unroll
    for (int i = 0; i < H3; ++i) {  // lane bit 3: X = v[i], Y = v[i + H3]
        double r = dpp_f64(0.0, v[i], 0);       // lanes 0-7 of a row: the partner's X
        r = dpp_f64(r, v[i + H3], 1);           // lanes 8-15: the partner's Y
        const double k = dpp_f64(v[i], v[i + H3], 2);  // own X (lanes 0-7) / own Y (lanes 8-15)
        v[i] = k + r;
    }
#pragma unroll
    for (int i = 0; i < H4; ++i) {  // lane bit 2
        double r = dpp_f64(0.0, v[i], 3);
        r = dpp_f64(r, v[i + H4], 4);
        const double k = dpp_f64(v[i], v[i + H4], 5);
        v[i] = k + r;
    }
    if (N == 32) {  // lane bit 1 halves the last pair
        const bool up = (lane & 2) != 0;
        const double k = up ? v[1] : v[0], s = up ? v[0] : v[1];
        v[0] = k + dpp_f64(0.0, s, 6);
    } else {
        v[0] += dpp_f64(0.0, v[0], 6);
    }
    v[0] += dpp_f64(0.0, v[0], 7);
}

// =====================================================================================
// the kernel: one wave per task (m, ring group)
// =====================================================================================
constexpr int VALU_WAVES = 1;  // waves per SIMD the register allocation is made for (tuning knob of tools/build_valu_variants.sh)
constexpr int VALU_CHK = 64;  // l between two promotion / liveness checks of a wave (a multiple of LB)
template <int SPIN>
__global__ __launch_bounds__(64, VALU_WAVES) void k_legendre_valu(ValuParams A, const double2 *__restrict__ coefn, const double *__restrict__ alphan)
{
    using C = ValuCfg<SPIN>;
    constexpr int R = C::R, LB = C::LB, NA = C::NA, NF = C::NF, NCH = 2;
    // recursion coefficients and output scalings alpha_l arrive in CHUNKS of CH = 32 l (this / the next chunk): the next chunk is
    // requested when a chunk starts and stored when it ends -- four (spin 0) / eight (spin 2) blocks later.  Handed over block by
    // block, the request had one block (~1300 cycles) to come back from L2 / HBM and every block ended waiting for it.
    constexpr int CH = 32, NSB = CH / LB;
    static_assert(CH % LB == 0, "a chunk is a whole number of blocks");
    __shared__ double2 cfs[2][CH];
    __shared__ double als[2][CH];
    const PlanDev &P = A.P;
    const LegTask task = A.tasks[blockIdx.x];
    const int m = task.m, lmax = P.lmax, lane = threadIdx.x;
    if ((m - A.m0) % A.ms) return;       // an order of another rank (m-sharded route: every ms-th order is ours)
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const int off = (l0 + m) & 1;        // parity of l + m at the first l (spin 2, m = 1 only)
    const long long cb = almidx(lmax, 0, m);
    const int coff = SPIN == 0 ? 0 : 1;  // spin-2 coefficients are indexed by the target l

    // ---- rings, operands, seeds ----
    double xx[R], vc[R][NCH], vp[R][NCH], f[R][NF];
    const double *fm = A.F + (long long)((m - A.m0) / A.ms) * P.nrp_pad * NF;  // operand rows of this m: [rp][NF]
    int sc[R][NCH];
    unsigned vmask = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = r * 64 + lane, rp = task.rb0 * RBLK + k;
        const bool valid = k < task.nrb * RBLK && rp < P.nrp;
        vmask |= valid ? (1u << r) : 0u;
        const double x = valid ? P.z[rp] : 0.0;
        xx[r] = SPIN == 0 ? x * x : x;
#pragma unroll
        for (int q = 0; q < NF; ++q) f[r][q] = 0.0;  // loaded (masked by the chains' state) by the first block that accumulates
#pragma unroll
        for (int c = 0; c < NCH; ++c) { vc[r][c] = 0.0; vp[r][c] = 0.0; sc[r][c] = -100; }
        if (valid) {
            if (SPIN == 0) {
                SVal a = spow(P.sth[rp], m);
                a.v *= P.mfac[m];
                SVal b = a;
                b.v *= sqrt(2.0 * m + 3.0) * x;  // lambda_{m+1,m} = sqrt(2m+3) x lambda_mm
                snorm_small(a);
                snorm_small(b);
                vc[r][0] = a.v; sc[r][0] = a.e;
                vc[r][1] = b.v; sc[r][1] = b.e;
            } else {
                SVal sp, sm;
                spin2_seeds(m, P.sth[rp], P.omz[rp], P.kfac2[m], sp, sm);
                vc[r][0] = sp.v; sc[r][0] = sp.e;
                vc[r][1] = sm.v; sc[r][1] = sm.e;
            }
        }
    }

    // promotion of a scaled chain that has grown past 1 (value *= 2^-300, exponent += 1), on the exponent bits, branch-free
    auto promote = [&](int r, int c) __attribute__((always_inline)) {
        const int hc = __double2hiint(vc[r][c]), hp = __double2hiint(vp[r][c]);
        const bool up = sc[r][c] < 0 && (hc & 0x7ff00000) >= 0x3ff00000;
        const int sub = up ? (300 << 20) : 0;
        const bool pz = up && (hp & 0x7ff00000) <= (300 << 20);
        vc[r][c] = __hiloint2double(hc - sub, __double2loint(vc[r][c]));
        vp[r][c] = pz ? 0.0 : __hiloint2double(hp - sub, __double2loint(vp[r][c]));
        sc[r][c] += up ? 1 : 0;
    };

    // one l step of slot r: two FMAs of the recursion, NA (spin 0) / 2 NA (spin 2) into the accumulators (ACC).
    // The two registers of a chain swap roles from step to step (the new value overwrites the older one): no register
    // moves; a block is an even number of steps per chain, so (vc, vp) = (current, previous) at its ends.
    double acc[LB * NA];
    auto step = [&](auto ACCC, int r, int s, const double2 c, const double (&fe)[NF]) __attribute__((always_inline)) {
        constexpr bool ACC = decltype(ACCC)::value;
        if (SPIN == 0) {
            const int ch = s & 1, odd = (s >> 1) & 1;
            double &va = odd ? vp[r][ch] : vc[r][ch], &vb = odd ? vc[r][ch] : vp[r][ch];  // va: current, vb: previous -> next
            const double t = fma(c.x, xx[r], c.y);
            if (ACC) {
                acc[s * NA + 0] = fma(va, fe[2 * ch + 0], acc[s * NA + 0]);
                acc[s * NA + 1] = fma(va, fe[2 * ch + 1], acc[s * NA + 1]);
            }
            vb = fma(t, va, -vb);
        } else {
            const int odd = s & 1;
            double &a0 = odd ? vp[r][0] : vc[r][0], &b0 = odd ? vc[r][0] : vp[r][0];
            double &a1 = odd ? vp[r][1] : vc[r][1], &b1 = odd ? vc[r][1] : vp[r][1];
            const double t0 = fma(c.x, xx[r], c.y), t1 = fma(c.x, xx[r], -c.y);
            if (ACC) {
                // G += lambda+ U +- lambda- V',  K += lambda- U' +- lambda+ V  (sign alternates with l)
                if (odd) {
                    acc[s * NA + 0] = fma(-a1, fe[6 % NF], fma(a0, fe[0], acc[s * NA + 0]));
                    acc[s * NA + 1] = fma(-a1, fe[7 % NF], fma(a0, fe[1], acc[s * NA + 1]));
                    acc[s * NA + 2 % NA] = fma(-a0, fe[2], fma(a1, fe[4 % NF], acc[s * NA + 2 % NA]));
                    acc[s * NA + 3 % NA] = fma(-a0, fe[3], fma(a1, fe[5 % NF], acc[s * NA + 3 % NA]));
                } else {
                    acc[s * NA + 0] = fma(a1, fe[6 % NF], fma(a0, fe[0], acc[s * NA + 0]));
                    acc[s * NA + 1] = fma(a1, fe[7 % NF], fma(a0, fe[1], acc[s * NA + 1]));
                    acc[s * NA + 2 % NA] = fma(a0, fe[2], fma(a1, fe[4 % NF], acc[s * NA + 2 % NA]));
                    acc[s * NA + 3 % NA] = fma(a0, fe[3], fma(a1, fe[5 % NF], acc[s * NA + 3 % NA]));
                }
            }
            b0 = fma(t0, a0, -b0);
            b1 = fma(t1, a1, -b1);
        }
    };
    using BT = std::integral_constant<bool, true>;
    using BF = std::integral_constant<bool, false>;

    if (lane < CH) {
        cfs[0][lane] = coefn[cb + l0 + coff + lane];
        als[0][lane] = alphan[cb + l0 + lane];
    }
    __syncthreads();
    double *prow = A.partial + (task.pout - A.row0) * NA;
    int buf = 0, lb = l0, sb = 0;  // sb: block within the chunk
    constexpr int SH = LB * NA == 32 ? 1 : 2;  // after the butterfly lane L holds accumulator L >> SH = (l - lb) NA + a
    // coefficient hand-over: the next chunk's values are requested at the start of a chunk and stored to LDS at its end
    double2 cpre = make_double2(0.0, 0.0);
    double apre = 0.0;
    auto stage_begin = [&]() __attribute__((always_inline)) {
        if (sb == 0 && lane < CH) {  // (lb is the first l of the chunk here)
            cpre = coefn[cb + lb + CH + coff + lane];
            apre = alphan[cb + lb + CH + lane];
        }
    };
    auto stage_end = [&]() __attribute__((always_inline)) {
        if (++sb == NSB) {
            sb = 0;
            if (lane < CH) {
                cfs[buf ^ 1][lane] = cpre;
                als[buf ^ 1][lane] = apre;
            }
            __syncthreads();
            buf ^= 1;
        }
    };
    // Scaled chains are promoted (and the state of the wave is looked at) every CHK l: the 16 promotions of a check cost as much
    // as a block of recursions.  A chain grows by less than ~2^7 per step (lambda_{m+1,m} / lambda_mm = sqrt(2m+3) x at worst), i.e.
    // by far less than 2^300 between two checks, and what it would have contributed between passing 2^-300 and its promotion is
    // below 2^-80 of the sum.
    constexpr int CHK = VALU_CHK / LB > 0 ? VALU_CHK / LB : 1;  // blocks between two checks
    int bk = 0, n_dead = 0, n_acc = 0;  // blocks of recursions only / of recursions + accumulation (executed-work counter)
    // ---- phase 0: every chain of the wave is still scaled -- recursions only, rows of zeros.  (A loop of its own: as a branch
    // inside the main loop it costs the register allocator 126 AGPRs and ~50 copies per block of the main path.) ----
    for (; lb <= lmax; lb += LB, ++bk) {
        if (bk % CHK == 0) {
            bool dead = true;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                promote(r, 0);
                promote(r, 1);
                dead = dead && (sc[r][0] < 0 && sc[r][1] < 0);  // (lanes without a ring carry -100)
            }
            if (!__all(dead)) break;
        }
        stage_begin();
        const double2 *cf = cfs[buf] + sb * LB;
#pragma unroll
        for (int s = 0; s < LB; ++s) {
            const double2 c = cf[s];
#pragma unroll
            for (int r = 0; r < R; ++r) step(BF{}, r, s, c, f[r]);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(lane & ((1 << SH) - 1))) {
            const int idx = lane >> SH, s = idx / NA, a = idx % NA;
            prow[(long long)(lb - l0 + s) * NA + a] = 0.0;
        }
        stage_end();
        ++n_dead;
    }
    // ---- phase 1: ONE accumulation loop for mixed and live blocks.  Until every chain of the wave is live ("steady"), a block
    // starts with the promotions and with operands that are zero for the chains still scaled (their values are finite:
    // 0 x value = 0), re-read from F. ----
    bool steady = false;
    bk = 0;  // (the block that left phase 0 has been promoted: the first check below finds nothing to promote and loads the operands)
    for (; lb <= lmax; lb += LB, ++bk) {
        stage_begin();
#pragma unroll
        for (int i = 0; i < LB * NA; ++i) acc[i] = 0.0;
        const double2 *cf = cfs[buf] + sb * LB;
        if (!steady && bk % CHK == 0) {
            bool live = true;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                promote(r, 0);
                promote(r, 1);
                const bool val = (vmask >> r) & 1;
                live = live && (!val || (sc[r][0] == 0 && sc[r][1] == 0));
            }
            steady = __all(live);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                // (a lane without a ring reads the last row of the m and selects 0: its chains carry exponent -100 for ever)
                const double *fp = fm + (long long)min(task.rb0 * RBLK + r * 64 + lane, P.nrp_pad - 1) * NF;
#pragma unroll
                for (int q = 0; q < NF; q += 2) {
                    // spin 0: (s, d) belong to chains 0, 1; spin 2: lambda+ (chain 0) multiplies U and V, lambda- (chain 1) U' and V'
                    const bool lv = sc[r][(SPIN == 0 ? q >= 2 : q >= 4) ? 1 : 0] == 0;
                    const double2 t = *reinterpret_cast<const double2 *>(fp + q);
                    const bool neg = SPIN == 2 && off && (q & 2);  // V, V': the alternating sign starts with -
                    f[r][q] = lv ? (neg ? -t.x : t.x) : 0.0;
                    f[r][q + 1] = lv ? (neg ? -t.y : t.y) : 0.0;
                }
            }
        }
        // l outermost, the R ring slots inside: consecutive instructions belong to different chains (a single wave per SIMD has
        // nothing else to cover the latency of a dependent FMA), one coefficient read per l.  Scheduling barriers keep hipcc
        // from running the recursions of several l ahead of their accumulation, which holds every intermediate value in a register.
        double2 cn = cf[0];
#pragma unroll
        for (int s = 0; s < LB; ++s) {
            const double2 c = cn;
            if (s + 1 < LB) cn = cf[s + 1];  // lands while this l runs
#pragma unroll
            for (int r = 0; r < R; ++r) step(BT{}, r, s, c, f[r]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- sum over the lanes, scale, store ----
        wave_reduce<LB * NA>(acc, lane);
        const int idx = lane >> SH, s = idx / NA, a = idx % NA;
        double out = acc[0] * als[buf][sb * LB + s];
        int col = a;
        if (SPIN == 2) {
            // lanes of accumulator a and a ^ 2 (G <-> K) differ in lane bit SH + 1: E = G + K, B = -i (G - K)
            const double oth = __shfl_xor(out, 2 << SH);
            // a = 0: G_re -> E_re = G_re + K_re (col 0); a = 1: G_im -> E_im (col 1); a = 2: K_re -> B_im = K_re - G_re (col 3);
            // a = 3: K_im -> B_re = G_im - K_im (col 2)
            out = a < 2 ? out + oth : (a == 2 ? out - oth : oth - out);
            col = a < 2 ? a : 5 - a;
        }
        if (!(lane & ((1 << SH) - 1))) prow[(long long)(lb - l0 + s) * NA + col] = out;  // rows are padded to whole 32-l blocks
        stage_end();
        ++n_acc;
    }
    if (lane == 0) {
        // per (lane, slot, l): 2 (spin 0) / 4 (spin 2) FMAs of recursion, + 2 / 8 of accumulation
        constexpr unsigned long long REC = SPIN == 0 ? 2 : 4, ACC = SPIN == 0 ? 2 : 8;
        atomicAdd(&g_valu_flops, (unsigned long long)(n_dead * REC + n_acc * (REC + ACC)) * (64ull * R * LB * 2ull));
    }
}

// =====================================================================================
// synthesis (alm -> ring modes) of one map / field on the vector unit: the mirror image of k_legendre_valu
// =====================================================================================
// healpy's map2alm runs Jacobi iterations by default (iter = 3: SURVEY 8a-2): three of the seven transforms of every
// Mapper.transform the reference issues (heracles/healpy.py:183-189) are SYNTHESES of a single map, and so is every hp.alm2map.
// In this direction the sum runs over l, inside one lane: with lane = ring pair there is no cross-lane reduction at all --
//   per (ring pair, l): 2 FMAs of the normalised recursion + 2 (spin 0) / 8 (spin 2) FMAs  acc += lambda_l(theta) x (alpha_l a_lm),
//   alpha_l a_lm wave-uniform (a broadcast read from the chunk tables in LDS, 32 l at a time, as in the analysis kernel);
//   spin 0: the even and the odd chain have an accumulator pair each:  F_N = E + O, F_S = E - O;
//   spin 2: P+_N = sum lambda+ a+,  P-_S = sum (-1)^(l+m) lambda+ a-,  P-_N = sum lambda- a-,  P+_S = sum (-1)^(l+m) lambda- a+,
//           a+- = -(E +- iB);  Q = (P+ + P-) / 2,  U = (P+ - P-) / 2i.
// Every accumulator belongs to ONE chain.  A scaled chain (exponent < 0) is not masked inside the loop: what it adds between two
// checks is dropped at the next check (its accumulators are zeroed as long as it is not live: it has never contributed before),
// and at the end.  Output Fv[m][ring pair][NV]: (N_re, N_im, S_re, S_im) per component -- spin 2: Q then U.
// NB maps (spin 0) / fields (spin 2) per sweep share the recursion: 2 + 2 NB (spin 0) / 4 + 8 NB (spin 2) FMAs per (ring pair, l)
// instead of NB (2 + 2) / NB (4 + 8).  Their accumulators take the registers of half the ring slots: R = 4 (tasks of 8 ring blocks).
template <int SPIN, int NB>
struct SynValuCfg {
    static constexpr int R = NB == 1 ? ValuCfg<SPIN>::R : 4;  // NB = 1: the task sets of the analysis kernel (2 R ring blocks per task)
    // l per unrolled block (even): hipcc requests the table values of a whole block up front -- LB (2 + NAV) doubles; spin 2 with
    // 8-l blocks took 58 registers beyond the 256 a lane can address
    static constexpr int LB = SPIN == 0 ? (NB <= 2 ? 8 : 4) : (NB == 1 ? 4 : 2);
    static constexpr int NA1 = SPIN == 0 ? 2 : 4;  // doubles of alpha_l a_lm per l and map / field
    static constexpr int NV1 = SPIN == 0 ? 4 : 8;  // output doubles per (m, ring pair) and map / field
    static constexpr int NAV = NA1 * NB, NV = NV1 * NB;
};
int synth_valu_max_units(int spin) { return spin == 0 ? 4 : 2; }
int synth_valu_task_blocks(int spin, int units) { return units == 1 ? valu_task_blocks(spin) : 2 * 4; }

struct SynValuParams {
    PlanDev P;
    const LegTask *__restrict__ tasks;
    const double2 *__restrict__ alm;  // component c at + c alm_stride: spin 0: the maps' alms; spin 2: (E, B) per field
    long long alm_stride;
    double *__restrict__ Fv;          // [m][rp][NV]: component c = (N_re, N_im, S_re, S_im) at 4 c (spin 2: Q, U of field b at 8 b)
};

// (four spin-0 maps: 294 registers if the compiler may -- copies in AGPRs around the checks and hand-overs; held to 256, i.e. two
// waves per SIMD like the other shapes, the copies become a few scratch accesses outside the steady loop)
template <int SPIN, int NB>
// (<0, 4>: two waves per SIMD at the price of 165 spilled registers -- 53.0 ms for four maps against 60.4 at one wave without a spill, round 6)
__global__ __launch_bounds__(64, (SPIN == 0 && NB == 4) ? 2 : VALU_WAVES) void k_legendre_synth_valu(SynValuParams A, const double2 *__restrict__ coefn, const double *__restrict__ alphan)
{
    using C = SynValuCfg<SPIN, NB>;
    constexpr int R = C::R, LB = C::LB, NAV = C::NAV, NV = C::NV, NA1 = C::NA1, NV1 = C::NV1, NCH = 2;
    constexpr int CH = 32, NSB = CH / LB;
    __shared__ double2 cfs[2][CH];
    __shared__ double avs[2][CH][NAV];
    const PlanDev &P = A.P;
    const LegTask task = A.tasks[blockIdx.x];
    const int m = task.m, lmax = P.lmax, lane = threadIdx.x;
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const int off = (l0 + m) & 1;        // parity of l + m at the first l (spin 2, m = 1 only)
    const long long cb = almidx(lmax, 0, m);
    const int coff = SPIN == 0 ? 0 : 1;  // spin-2 coefficients are indexed by the target l

    double xx[R], vc[R][NCH], vp[R][NCH], acc[R][NV];
    int sc[R][NCH];
    unsigned vmask = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int k = r * 64 + lane, rp = task.rb0 * RBLK + k;
        const bool valid = k < task.nrb * RBLK && rp < P.nrp;
        vmask |= valid ? (1u << r) : 0u;
        const double x = valid ? P.z[rp] : 0.0;
        xx[r] = SPIN == 0 ? x * x : x;
#pragma unroll
        for (int q = 0; q < NV; ++q) acc[r][q] = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) { vc[r][c] = 0.0; vp[r][c] = 0.0; sc[r][c] = -100; }
        if (valid) {
            if (SPIN == 0) {
                SVal a = spow(P.sth[rp], m);
                a.v *= P.mfac[m];
                SVal b = a;
                b.v *= sqrt(2.0 * m + 3.0) * x;
                snorm_small(a);
                snorm_small(b);
                vc[r][0] = a.v; sc[r][0] = a.e;
                vc[r][1] = b.v; sc[r][1] = b.e;
            } else {
                SVal sp, sm;
                spin2_seeds(m, P.sth[rp], P.omz[rp], P.kfac2[m], sp, sm);
                vc[r][0] = sp.v; sc[r][0] = sp.e;
                vc[r][1] = sm.v; sc[r][1] = sm.e;
            }
        }
    }
    auto promote = [&](int r, int c) __attribute__((always_inline)) {
        const int hc = __double2hiint(vc[r][c]), hp = __double2hiint(vp[r][c]);
        const bool up = sc[r][c] < 0 && (hc & 0x7ff00000) >= 0x3ff00000;
        const int sub = up ? (300 << 20) : 0;
        const bool pz = up && (hp & 0x7ff00000) <= (300 << 20);
        vc[r][c] = __hiloint2double(hc - sub, __double2loint(vc[r][c]));
        vp[r][c] = pz ? 0.0 : __hiloint2double(hp - sub, __double2loint(vp[r][c]));
        sc[r][c] += up ? 1 : 0;
    };
    // accumulators of chain c of slot r, map / field b: spin 0: NV1 b + (2 c, 2 c + 1); spin 2: lambda+ (c = 0) NV1 b + 0..3, lambda- (c = 1) + 4..7
    auto drop = [&](int r, int c) __attribute__((always_inline)) {
        const bool z = sc[r][c] != 0;
        constexpr int NQ = NV1 / 2;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[r][b * NV1 + c * NQ + q] = z ? 0.0 : acc[r][b * NV1 + c * NQ + q];
    };
    // one l step of slot r (the registers of a chain swap roles from step to step, as in k_legendre_valu); av = alpha_l a_lm
    auto step = [&](auto ACCC, int r, int s, const double2 c, const double (&av)[NAV]) __attribute__((always_inline)) {
        constexpr bool ACC = decltype(ACCC)::value;
        if (SPIN == 0) {
            const int ch = s & 1, odd = (s >> 1) & 1;
            double &va = odd ? vp[r][ch] : vc[r][ch], &vb = odd ? vc[r][ch] : vp[r][ch];
            const double t = fma(c.x, xx[r], c.y);
            if (ACC) {
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    acc[r][b * NV1 + 2 * ch + 0] = fma(va, av[b * NA1 + 0], acc[r][b * NV1 + 2 * ch + 0]);
                    acc[r][b * NV1 + 2 * ch + 1] = fma(va, av[b * NA1 + 1], acc[r][b * NV1 + 2 * ch + 1]);
                }
            }
            vb = fma(t, va, -vb);
        } else {
            const int odd = s & 1;
            double &a0 = odd ? vp[r][0] : vc[r][0], &b0 = odd ? vc[r][0] : vp[r][0];
            double &a1 = odd ? vp[r][1] : vc[r][1], &b1 = odd ? vc[r][1] : vp[r][1];
            const double t0 = fma(c.x, xx[r], c.y), t1 = fma(c.x, xx[r], -c.y);
            if (ACC) {
                // av = (a+_re, a+_im, a-_re, a-_im) per field; the southern sums alternate in sign with l (the overall sign (-1)^off at the end)
#pragma unroll
                for (int b = 0; b < NB; ++b) {
                    const int o = b * NV1, v = b * NA1;
                    acc[r][o + 0] = fma(a0, av[v + 0], acc[r][o + 0]);
                    acc[r][o + 1] = fma(a0, av[v + 1], acc[r][o + 1]);
                    acc[r][o + 2] = fma(odd ? -a0 : a0, av[v + 2 % NA1], acc[r][o + 2]);
                    acc[r][o + 3] = fma(odd ? -a0 : a0, av[v + 3 % NA1], acc[r][o + 3]);
                    acc[r][o + 4 % NV1] = fma(a1, av[v + 2 % NA1], acc[r][o + 4 % NV1]);
                    acc[r][o + 5 % NV1] = fma(a1, av[v + 3 % NA1], acc[r][o + 5 % NV1]);
                    acc[r][o + 6 % NV1] = fma(odd ? -a1 : a1, av[v + 0], acc[r][o + 6 % NV1]);
                    acc[r][o + 7 % NV1] = fma(odd ? -a1 : a1, av[v + 1], acc[r][o + 7 % NV1]);
                }
            }
            b0 = fma(t0, a0, -b0);
            b1 = fma(t1, a1, -b1);
        }
    };
    using BT = std::integral_constant<bool, true>;
    using BF = std::integral_constant<bool, false>;

    // chunk tables: lane < CH carries l = (first l of the chunk) + lane: coefficients of the step and alpha_l a_lm (0 beyond lmax)
    double2 cpre = make_double2(0.0, 0.0);
    double apre[NAV];
    auto fetch = [&](int lc) __attribute__((always_inline)) {
        if (lane < CH) {
            const int l = lc + lane;
            cpre = coefn[cb + l + coff];
            const bool in = l <= lmax;
            const double al = alphan[cb + l];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                if (SPIN == 0) {
                    const double2 a = in ? A.alm[b * A.alm_stride + cb + l] : make_double2(0.0, 0.0);
                    apre[b * NA1 + 0] = al * a.x; apre[b * NA1 + 1] = al * a.y;
                } else {
                    const double2 E = in ? A.alm[2 * b * A.alm_stride + cb + l] : make_double2(0.0, 0.0);
                    const double2 B = in ? A.alm[(2 * b + 1) * A.alm_stride + cb + l] : make_double2(0.0, 0.0);
                    apre[b * NA1 + 0] = al * (-E.x + B.y); apre[b * NA1 + 1] = al * (-E.y - B.x);              // a+ = -(E + iB)
                    apre[b * NA1 + 2 % NA1] = al * (-E.x - B.y); apre[b * NA1 + 3 % NA1] = al * (-E.y + B.x);  // a- = -(E - iB)
                }
            }
        }
    };
    auto stash = [&](int b) __attribute__((always_inline)) {
        if (lane < CH) {
            cfs[b][lane] = cpre;
#pragma unroll
            for (int q = 0; q < NAV; ++q) avs[b][lane][q] = apre[q];
        }
    };
    fetch(l0);
    stash(0);
    __syncthreads();
    int buf = 0, lb = l0, sb = 0;
    auto stage_begin = [&]() __attribute__((always_inline)) {
        if (sb == 0) fetch(lb + CH);
    };
    auto stage_end = [&]() __attribute__((always_inline)) {
        if (++sb == NSB) {
            sb = 0;
            stash(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    };
    constexpr int CHK = VALU_CHK / LB > 0 ? VALU_CHK / LB : 1;
    int bk = 0, n_dead = 0, n_acc = 0;
    // ---- phase 0: every chain of the wave still scaled: recursions only ----
    const double zero_av[NAV] = {};
    for (; lb <= lmax; lb += LB, ++bk) {
        if (bk % CHK == 0) {
            bool dead = true;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                promote(r, 0);
                promote(r, 1);
                dead = dead && (sc[r][0] < 0 && sc[r][1] < 0);
            }
            if (!__all(dead)) break;
        }
        stage_begin();
        const double2 *cf = cfs[buf] + sb * LB;
#pragma unroll
        for (int s = 0; s < LB; ++s) {
            const double2 c = cf[s];
#pragma unroll
            for (int r = 0; r < R; ++r) step(BF{}, r, s, c, zero_av);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage_end();
        ++n_dead;
    }
    // ---- phase 1: accumulation.  Until every chain of the wave is live, a check every CHK l promotes the scaled chains and
    // drops what the chains that are not live have added since the last check ----
    bool steady = false;
    bk = 0;  // (the block that left phase 0 has been promoted; nothing has been accumulated yet)
    for (; lb <= lmax; lb += LB, ++bk) {
        stage_begin();
        if (!steady && bk % CHK == 0) {
            bool live = true;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                drop(r, 0);
                drop(r, 1);
                promote(r, 0);
                promote(r, 1);
                const bool val = (vmask >> r) & 1;  // (lanes without a ring carry zeros and exponent -100 for ever)
                live = live && (!val || (sc[r][0] == 0 && sc[r][1] == 0));
            }
            steady = __all(live);
        }
        const double2 *cf = cfs[buf] + sb * LB;
        const double (*av)[NAV] = avs[buf] + sb * LB;
        // wide sweeps (8 table doubles per l): hipcc otherwise requests the values of the WHOLE block in the block before it -- 80
        // registers beside 64 accumulators, 294 in all, one wave per SIMD; the empty asm statements pin the look-ahead to one l
        constexpr bool PIN = NAV > 4;
        double2 cn = cf[0];
        double an[NAV];
        if (PIN) asm volatile("" ::: "memory");
#pragma unroll
        for (int q = 0; q < NAV; ++q) an[q] = av[0][q];
#pragma unroll
        for (int s = 0; s < LB; ++s) {
            const double2 c = cn;
            double a[NAV];
#pragma unroll
            for (int q = 0; q < NAV; ++q) a[q] = an[q];
            if (PIN) asm volatile("" ::: "memory");
            if (s + 1 < LB) {  // the next l's values land while this l runs
                cn = cf[s + 1];
#pragma unroll
                for (int q = 0; q < NAV; ++q) an[q] = av[s + 1][q];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) step(BT{}, r, s, c, a);
            __builtin_amdgcn_sched_barrier(0);
        }
        stage_end();
        ++n_acc;
    }
    // ---- ring modes of this m ----
    double *fm = A.Fv + (long long)m * P.nrp_pad * NV;
    const double ssgn = (SPIN == 2 && off) ? -1.0 : 1.0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        drop(r, 0);  // chains that never became live
        drop(r, 1);
        const int k = r * 64 + lane, rp = task.rb0 * RBLK + k;
        if (k >= task.nrb * RBLK || rp >= P.nrp_pad) continue;
        double o[NV];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const double *ac = acc[r] + b * NV1;
            double *ob = o + b * NV1;
            if (SPIN == 0) {
                ob[0] = ac[0] + ac[2]; ob[1] = ac[1] + ac[3];
                ob[2] = ac[0] - ac[2]; ob[3] = ac[1] - ac[3];
            } else {
                const double ppn_r = ac[0], ppn_i = ac[1], pms_r = ssgn * ac[2], pms_i = ssgn * ac[3];
                const double pmn_r = ac[4 % NV1], pmn_i = ac[5 % NV1], pps_r = ssgn * ac[6 % NV1], pps_i = ssgn * ac[7 % NV1];
                // Q = (P+ + P-) / 2;  U = (P+ - P-) / 2i: U_re = (Im P+ - Im P-) / 2, U_im = -(Re P+ - Re P-) / 2
                ob[0] = 0.5 * (ppn_r + pmn_r); ob[1] = 0.5 * (ppn_i + pmn_i);
                ob[2] = 0.5 * (pps_r + pms_r); ob[3] = 0.5 * (pps_i + pms_i);
                ob[4 % NV1] = 0.5 * (ppn_i - pmn_i); ob[5 % NV1] = -0.5 * (ppn_r - pmn_r);
                ob[6 % NV1] = 0.5 * (pps_i - pms_i); ob[7 % NV1] = -0.5 * (pps_r - pms_r);
            }
        }
        double *row = fm + (long long)rp * NV;
#pragma unroll
        for (int q = 0; q < NV; q += 2) *reinterpret_cast<double2 *>(row + q) = make_double2(o[q], o[q + 1]);
    }
    if (lane == 0) {
        constexpr unsigned long long REC = SPIN == 0 ? 2 : 4, ACC = (SPIN == 0 ? 2 : 8) * NB;
        atomicAdd(&g_valu_flops, (unsigned long long)(n_dead * REC + n_acc * (REC + ACC)) * (64ull * R * LB * 2ull));
    }
}

// Fv of `units` maps (spin 0: 1, 2 or 4) / fields (spin 2: 1 or 2) whose alms start at d_alm; ts = the task set of
// synth_valu_task_blocks(spin, units) ring blocks per task; rings outside the task list (pruned) stay zero
template <int SPIN, int NB>
static int launch_synth_valu_t(hx_plan *pl, hx_plan::TaskSet &ts, const double2 *d_alm, double *d_Fv)
{
    hipStream_t st = rt().stream;
    constexpr int NV = SynValuCfg<SPIN, NB>::NV;
    HX_HIP(hipMemsetAsync(d_Fv, 0, sizeof(double) * (size_t)(pl->lmax + 1) * pl->nrp_pad * NV, st));
    SynValuParams A;
    A.P = pl->dev(); A.tasks = ts.d_tasks.as<LegTask>(); A.alm = d_alm; A.alm_stride = pl->nlm; A.Fv = d_Fv;
    const double2 *cn = SPIN == 0 ? pl->cn0.as<double2>() : pl->cn2.as<double2>();
    const double *al = SPIN == 0 ? pl->al0.as<double>() : pl->al2.as<double>();
    ProfScope ps("legendre_synthesis");
    ProfScope ps2("legendre_synth_valu");
    hipLaunchKernelGGL((k_legendre_synth_valu<SPIN, NB>), dim3((unsigned)ts.tasks.size()), dim3(64), 0, st, A, cn, al);
    HX_HIP(hipGetLastError());
    return HX_OK;
}
int launch_synth_valu(hx_plan *pl, int spin, int units, hx_plan::TaskSet &ts, const double2 *d_alm, double *d_Fv)
{
    if (spin == 0 && units == 1) return launch_synth_valu_t<0, 1>(pl, ts, d_alm, d_Fv);
    if (spin == 0 && units == 2) return launch_synth_valu_t<0, 2>(pl, ts, d_alm, d_Fv);
    if (spin == 0 && units == 4) return launch_synth_valu_t<0, 4>(pl, ts, d_alm, d_Fv);
    if (spin == 2 && units == 1) return launch_synth_valu_t<2, 1>(pl, ts, d_alm, d_Fv);
    if (spin == 2 && units == 2) return launch_synth_valu_t<2, 2>(pl, ts, d_alm, d_Fv);
    return fail(HX_ERR_ARG, "launch_synth_valu: %d units of spin %d", units, spin);
}

// =====================================================================================
// host: one m-chunk of one map / field
// =====================================================================================
template <int SPIN>
static int launch_valu_chunk_t(hx_plan *pl, hx_plan::TaskSet &ts, int m0, int m1, int c0, const double *d_rw)
{
    hipStream_t st = rt().stream;
    PlanDev P = pl->dev();
    const int t0 = ts.of_m[m0].first;
    const int t1 = ts.of_m[m1 - 1].first + ts.of_m[m1 - 1].count;
    const int ms = std::max(pl->m_step, 1);
    {
        ProfScope ps("fourier_combine");
        dim3 grid((m1 - m0 + ms - 1) / ms, (pl->nrp_pad + 255) / 256);
        hipLaunchKernelGGL(k_fourier_combine_valu<SPIN>, grid, dim3(256), 0, st, P, pl->Y.as<double2>(), c0, m0, ms, d_rw,
                           ts.d_tasks.as<LegTask>(), ts.d_of_m.as<MTasks>(), pl->F.as<double>());
    }
    if (t1 > t0) {
        ProfScope ps("legendre_analysis");
        ProfScope ps2(SPIN == 0 ? "legendre_analysis_s0" : "legendre_analysis_s2");
        ProfScope ps3("legendre_valu");
        ValuParams A;
        A.P = P; A.tasks = ts.d_tasks.as<LegTask>() + t0; A.F = pl->F.as<double>(); A.partial = pl->partial.as<double>();
        A.m0 = m0; A.ms = ms; A.row0 = ts.rows_before_m[m0];
        const double2 *cn = SPIN == 0 ? pl->cn0.as<double2>() : pl->cn2.as<double2>();
        const double *al = SPIN == 0 ? pl->al0.as<double>() : pl->al2.as<double>();
        hipLaunchKernelGGL(k_legendre_valu<SPIN>, dim3((unsigned)(t1 - t0)), dim3(64), 0, st, A, cn, al);
    }
    HX_HIP(hipGetLastError());
    return HX_OK;
}

int valu_exec_flops(unsigned long long *v, bool reset)
{
    HX_HIP(hipMemcpyFromSymbol(v, HIP_SYMBOL(g_valu_flops), sizeof(*v)));
    const unsigned long long z = 0;
    if (reset) HX_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_valu_flops), &z, sizeof(z)));
    return HX_OK;
}

int launch_valu_chunk(hx_plan *pl, int spin, hx_plan::TaskSet &ts, int m0, int m1, int c0, const double *d_rw)
{
    return spin == 0 ? launch_valu_chunk_t<0>(pl, ts, m0, m1, c0, d_rw) : launch_valu_chunk_t<2>(pl, ts, m0, m1, c0, d_rw);
}

}  // namespace hx
