// hx_synth_duo.hip -- batched Legendre synthesis (alm -> ring modes F_m(ring)) on the FP64 matrix unit, round 5.
//
// healpy's map2alm runs Jacobi iterations by default (iter = 3; the reference passes none: heracles/healpy.py:183-189), so three of
// the seven transforms behind every default HipHealpixMapper.transform are SYNTHESES.  Batches of them ran on the vector unit
// (k_legendre_synth_valu, four maps / two fields per sweep): ten fields 408 ms against 322 ms for their analysis.  This kernel is the
// mirror of k_legendre_duo (hx_analysis.hip) and uses what that kernel taught:
//   F[ring][col] = sum_l Lambda[l][ring] (alpha_l a[l][col])   as   D[16 chains x 16 cols] += A[16 chains x 4 l] B[4 l x 16 cols]
//   * TWO independent work-groups of 4 waves per CU (<= 256 registers, 18 KiB of LDS each), one per (m, ring group), never synchronised
//     with each other: one group's recursion and barrier waits sit under the other's matrix block;
//   * a wave owns 64 recursion chains (spin 0: 64 ring pairs, both parity chains of the two-step recursion in a lane; spin 2: 32 ring
//     pairs x the two Wigner functions, the lambda- chain carried as (-1)^(l + m) lambda- so that both halves of the wave share ONE
//     pair of recursion coefficients per step, fetched by row broadcast from per-lane loads -- no LDS, no scalar loads);
//   * per 16-l block: 16 recursion steps into the wave's 8 KiB tile -> 16 (tile, k-slot, parity) positions of matrix instructions whose
//     A operand (16 chains x 4 l of one parity) comes out of that tile and whose B operand (alpha_l a_lm: 4 l x 16 columns) comes out of
//     a table in LDS that the four waves share (double-buffered, one barrier per block; its global image is made by k_synth_table, so
//     the main kernel spends no vector instruction on it);
//   * the D tiles are indexed (chain, column): they stay in registers over the WHOLE l sweep and are written once -- no flush, no
//     atomics, no lead-in logic;
//   * even-parity and odd-parity l accumulate separately (N = even + odd, S = even - odd), so one recursion serves both hemispheres.
// Columns: spin 0: (re, im) per map; spin 2: (a+_re, a+_im, a-_re, a-_im) per field, a+- = -(E +- iB).  Shapes (NG groups of 16 columns +
// NBX blocks of 4) as in the analysis: up to 40 columns = 20 maps / 10 fields per sweep.
// Output Fv[m][ring pair][4 per component]: (N_re, N_im, S_re, S_im) per map (spin 0) / for Q then U of a field (spin 2) -- the layout of
// the vector-unit kernel, read by k_synth_spectrum_v.  A lane ends up holding half of two components; one exchange with its neighbour
// (lane ^ 1) leaves it with one whole component, stored as 32 (spin 0: 16) contiguous bytes.
#include <algorithm>

#include "hx_sht_common.h"

namespace hx {

constexpr int SLB = 16;  // l per block: 8 rows of (even, odd)
// blocks per stage of the shared B-operand table.  During the matrix work of block b the pieces of block b + SB go from HBM straight
// into the OTHER stage's buffer (global_load_lds_dwordx4: no register, no vector or LDS-store instruction, and -- unlike loads into
// registers, which the compiler made the matrix block wait for -- nothing waits for them before the barrier that closes the stage:
// vmcnt(0), barrier, once per SB blocks)
constexpr int SYN_SB = 4;
constexpr int SB = SYN_SB;

constexpr int SYN_SCALE_BITS2 = 100;
constexpr int SYN_SCALE_BITS0 = 300;
constexpr int SYN_GAP = -1;  // s_nop (n - 1) behind every (16 x 16 x 4, 4 x 4 x 4) group of a position: lets the other group's vector work in (see HX_DUO_GAP)
constexpr int SYN_PRIO = 1;
// (Measured and not kept, round 5: the additive recursion coefficient q' of a step from a wave-private LDS table -- a broadcast read
// instead of the row-broadcast move, 2 vector instructions per step instead of 3: ten fields 330.6 against 319.8 ms; the recursion is
// bound by the latency of its dependent chain, and an LDS trip per step lengthens the chain.)

template <int K>
__device__ __forceinline__ double syn_row_bcast(double v)
{
    double d;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "n"(K));
    return d;
}
template <int K>
__device__ __forceinline__ double syn_row_bcast_fmac(double t, double p, double x)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(p), "v"(x), "n"(K));
    return t;
}

// one 16-byte load per lane straight into LDS: lane l's data lands at lds_dst + 16 l (through M0, saved and restored); address = scalar
// base + 32-bit lane offset.  The compiler does not know about the load: nothing of its own waits for it, the kernel drains it with an
// explicit vmcnt(0) in front of the barrier that closes a stage.
__device__ __forceinline__ void syn_glds16(const double2 *sbase, int voff_bytes, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff_bytes), "s"(sbase), "s"(lds_dst)
                 : "memory");
}
// (a pointer the compiler holds in vector registers although it is the same in every lane: make it scalar)
__device__ __forceinline__ const double2 *syn_uniform(const double2 *p)
{
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const double2 *)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ unsigned syn_lds_addr(const void *p)
{
    return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void *)p;
}

// Tile of a wave: chain c < 64, row r < 8 (l = lb + 2 r + p), position p at double index c 16 + chunk 2 + p with the 16-byte chunk
// swizzled so that both the stores (8 consecutive chains, one row: ds_write_b128 groups) and the A-operand reads (ds_read_b128 lane
// groups {0-3, 12-15 | 20-27}: chains i of one 16-chain tile at rows r and r + 1) fall on distinct banks.
__device__ __forceinline__ int syn_tile_idx(int c, int r) { return c * 16 + (((r ^ (c >> 1) ^ ((c & 1) << 2)) & 7) * 2); }

struct SynDuoParams {
    PlanDev P;
    const LegTask *__restrict__ tasks;
    const double *__restrict__ tab;        // k_synth_table's image: block (boff[m] + b) at that index x (NG 256 + NBX 64) doubles
    const long long *__restrict__ boff;    // first block of every m
    double *__restrict__ Fv;               // [m][rp][rowlen]
    int rowlen;                            // doubles per (m, ring pair): 4 per map / 8 per field
    int nunits;                            // maps / fields of the sweep (columns beyond them are zero and are not stored)
};

// ---- the B operands: tab[block][NG groups: [8 rows][16 cols][2 parities] | NBX blocks: [8 rows][4 cols][2]] = alpha_l a_lm ----
template <int SPIN>
__global__ __launch_bounds__(256) void k_synth_table(int lmax, const double2 *__restrict__ alm, long long alm_stride, int nunits,
                                                     const double *__restrict__ alphan, int ng, int nbx, const long long *__restrict__ boff,
                                                     double *__restrict__ tab)
{
    const int m = blockIdx.x;
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    if (l0 > lmax) return;
    const int nblk = (lmax - l0) / SLB + 1;
    const int np = ng * 128 + nbx * 32;  // 16-byte pieces per block
    const long long cb = almidx(lmax, 0, m);
    double2 *out = reinterpret_cast<double2 *>(tab) + boff[m] * np;
    for (long long i = (long long)blockIdx.y * blockDim.x + threadIdx.x; i < (long long)nblk * np; i += (long long)gridDim.y * blockDim.x) {
        const int b = (int)(i / np), pc = (int)(i % np);
        int row, col;  // col: global column index of the sweep
        if (pc < ng * 128) {
            row = (pc & 127) >> 4;
            col = (pc >> 7) * 16 + (pc & 15);
        } else {
            const int q = pc - ng * 128;
            row = (q & 31) >> 2;
            col = ng * 16 + (q >> 5) * 4 + (q & 3);
        }
        double v[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int l = l0 + b * SLB + 2 * row + p;
            double r = 0.0;
            if (l <= lmax) {
                const double al = alphan[cb + l];
                if (SPIN == 0) {
                    const int u = col >> 1;
                    if (u < nunits) {
                        const double2 a = alm[u * alm_stride + cb + l];
                        r = al * ((col & 1) ? a.y : a.x);
                    }
                } else {
                    const int u = col >> 2, q = col & 3;
                    if (u < nunits) {
                        const double2 E = alm[2 * u * alm_stride + cb + l], B = alm[(2 * u + 1) * alm_stride + cb + l];
                        // a+ = -(E + iB), a- = -(E - iB)
                        r = al * (q == 0 ? -E.x + B.y : q == 1 ? -E.y - B.x : q == 2 ? -E.x - B.y : -E.y + B.x);
                    }
                }
            }
            v[p] = r;
        }
        out[i] = make_double2(v[0], v[1]);
    }
}

template <int SPIN, int NG, int NBX>
__global__ __launch_bounds__(256, 2) void k_synth_duo(SynDuoParams A, const double2 *__restrict__ coefn)
{
    constexpr int NW = 4, NCH = SPIN == 0 ? 2 : 1, NXA = NBX > 0 ? NBX : 1;
    constexpr int RD = NG * 256 + NBX * 64;  // doubles of one block's table
    constexpr int NP = RD / 2;               // its 16-byte pieces: thread t stages piece t (and t + 256)
    // gaps in the matrix stream let the other group's vector work in (HX_DUO_GAP of the analysis kernel): ten spin-0 maps 91.1 -> 87.6 ms,
    // five fields 178.4 -> 170.9 with s_nop 4; the 40-column shape is the same with and without (319-326 ms for 0, 3 ... 8)
    constexpr int GAPN = SYN_GAP >= 0 ? SYN_GAP : ((NG == 2 && NBX == 2) ? 0 : 5);
    constexpr int SCB = SPIN == 2 ? SYN_SCALE_BITS2 : SYN_SCALE_BITS0;  // step of the scaled chains (as HX_DUO_SCALE_BITS in hx_analysis.hip)
    static_assert(NG >= 1 && NG <= 2 && NBX >= 0 && NBX <= 2 && NP <= 512, "shape");
    __shared__ double tile[NW][64 * 16];  // 8 KiB per wave
    __shared__ double tab[2][SB][RD];     // <= 40 KiB: two stages of SB blocks
    const PlanDev &P = A.P;
    const LegTask task = A.tasks[blockIdx.x];
    const int m = task.m, lmax = P.lmax;
    const int l0 = SPIN == 0 ? m : (m > 2 ? m : 2);
    const int off = (l0 + m) & 1;
    const int nblk = (lmax - l0) / SLB + 1;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, ai = lane & 15, ak = lane >> 4;

    // chain of this lane: wave w holds ring block w (spin 2: both functions) / blocks 2 w, 2 w + 1 (spin 0) of the task
    const int rbi = SPIN == 0 ? 2 * w + (lane >> 5) : w;
    const int rpl = (task.rb0 + rbi) * RBLK + (lane & 31);
    const bool valid = rbi < task.nrb && rpl < P.nrp;
    const double x = valid ? P.z[rpl] : 0.0;
    const double xx = SPIN == 0 ? x * x : ((lane >> 5) ? -x : x);

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
            vc[0] = (lane >> 5) ? (off ? -sm.v : sm.v) : sp.v;  // the lambda- chain is carried as (-1)^(l + m) lambda-
            sc[0] = (lane >> 5) ? sm.e : sp.e;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) sval_rebase<SCB>(vc[c], sc[c]);
    }
    auto rec_step = [&](auto RMM, int c, int step, const double tq) __attribute__((always_inline)) {
        constexpr int RM = decltype(RMM)::value;
        if (RM != 3 && (step & 3) == 0) {  // promotion of scaled chains (value = v 2^(SCB e), live iff e = 0), on the exponent bits
            const int hc = __double2hiint(vc[c]), hp = __double2hiint(vp[c]);
            const bool up = sc[c] < 0 && (hc & 0x7ff00000) >= 0x3ff00000;
            const int sub = up ? (SCB << 20) : 0;
            const bool pz = up && (hp & 0x7ff00000) <= (SCB << 20);
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
    double *tw = &tile[w][0];
    // the 16 steps of a block; cl = this lane's coefficient pair (p', q') of step (lane & 15): wave-uniform per step, taken by row broadcast
    auto recursion = [&](auto RMM, const double2 cl) __attribute__((always_inline)) {
        constexpr int RM = decltype(RMM)::value;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double cur[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int kk = 8 * h + k;
                double tq = 0.0;
                switch (kk) {
#define HX_BC(K) case K: tq = syn_row_bcast_fmac<K>(syn_row_bcast<K>(cl.y), cl.x, xx); break;
                    HX_BC(0) HX_BC(1) HX_BC(2) HX_BC(3) HX_BC(4) HX_BC(5) HX_BC(6) HX_BC(7)
                    HX_BC(8) HX_BC(9) HX_BC(10) HX_BC(11) HX_BC(12) HX_BC(13) HX_BC(14) HX_BC(15)
#undef HX_BC
                }
                cur[k] = rec_step(RMM, SPIN == 0 ? ((kk & 1) ? NCH - 1 : 0) : 0, SPIN == 0 ? kk >> 1 : kk, tq);
            }
            if (RM >= 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *reinterpret_cast<double2 *>(tw + syn_tile_idx(lane, 4 * h + j)) = make_double2(cur[2 * j], cur[2 * j + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // (SCB < 300: a lane is dead for block b only below 2^-(SCB + E_b) at the entry -- see set_mode of k_legendre_duo; blocks of 16 l:
    // E_b = 26, 12, 8, 8, ... from tools/calibrate_dead_blocks.py 16: needed 17.0 / 3.8 / 0 at lmax 12288 (m 12088, nside 8192) as at
    // lmax 6144 (16.9 / 3.7): >= 8 bits to spare at every size, profiles/r06_dead_block_calibration.txt)
    auto set_mode = [&](int b) __attribute__((always_inline)) {
        const int eb = b == 0 ? 26 : (b == 1 ? 12 : 8);
        auto lane_dead = [&](int c) __attribute__((always_inline)) {
            if (SCB == 300) return sc[c] < 0;
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

    double4_t acc[4][2][NG];
    double accx[4][2][NXA];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[mt][p][g] = (double4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int g = 0; g < NXA; ++g) accx[mt][p][g] = 0.0;
        }

    const double2 *__restrict__ cfm = coefn + almidx(lmax, 0, m) + l0 + (SPIN == 0 ? 0 : 1) + (tid & 15);
    const double2 *tsrc = syn_uniform(reinterpret_cast<const double2 *>(A.tab) + A.boff[m] * NP);
    auto lds_barrier = []() __attribute__((always_inline)) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
    };
    // pieces (16 bytes) of a block's table: NP of them, 64 per wave instruction; instruction i is wave (i & 3)'s
    constexpr int NI = (NP + 63) / 64;
    auto dma_block = [&](int blk, double *dst) __attribute__((always_inline)) {
        const double2 *src = syn_uniform(tsrc + (long long)blk * NP);
#pragma unroll
        for (int i = 0; i < NI; ++i)
            if ((i & 3) == w && i * 64 + lane < NP) syn_glds16(src, (i * 64 + lane) * 16, syn_lds_addr(dst) + i * 1024);
    };
    // table of stage 0 (blocks 0 .. SB - 1)
#pragma unroll
    for (int j = 0; j < SB; ++j)
        if (j < nblk) dma_block(j, &tab[0][j][0]);
    double2 cnext = cfm[0];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if (SYN_PRIO) __builtin_amdgcn_s_setprio(3);
    for (int b = 0; b < nblk; ++b) {
        // block b + SB belongs to the next stage: its pieces are requested now and stored behind this block's matrix work
        const int stg = (b / SB) & 1, bj = b % SB;
        const double2 cl = cnext;
        if (b + SB < nblk) dma_block(b + SB, &tab[stg ^ 1][bj][0]);
        cnext = cfm[(b + 1) * SLB];
        int rm = set_mode(b);
        if (rm == 3) recursion(I3{}, cl);
        else if (rm == 2) {
            recursion(I2{}, cl);
            if (SCB != 300) {  // a mixed block in which no chain came to life stored zeros only
                bool lv = valid && sc[0] == 0;
                if (NCH == 2) lv = lv || (valid && sc[NCH - 1] == 0);
                if (!__any(lv)) rm = 1;
            }
        } else recursion(I1{}, cl);
        if (rm >= 2) {
            if (SYN_PRIO) __builtin_amdgcn_s_setprio(0);
            const double *tb = &tab[stg][bj][0];
            auto a_fetch = [&](int mt, int ks) __attribute__((always_inline)) {
                return *reinterpret_cast<const double2 *>(tw + syn_tile_idx(16 * mt + ai, 4 * ks + ak));
            };
            double2 bg[2][NG], bq[2][NXA];
            auto b_fetch = [&](int ks) __attribute__((always_inline)) {
                const int row = 4 * ks + ak;
#pragma unroll
                for (int g = 0; g < NG; ++g) bg[ks & 1][g] = *reinterpret_cast<const double2 *>(tb + g * 256 + row * 32 + ai * 2);
#pragma unroll
                for (int g = 0; g < NXA; ++g)
                    bq[ks & 1][g] = NBX > 0 ? *reinterpret_cast<const double2 *>(tb + NG * 256 + g * 64 + row * 8 + (lane & 3) * 2) : make_double2(0.0, 0.0);
            };
            b_fetch(0);
            double2 aq[3];
            aq[0] = a_fetch(0, 0);
            aq[1] = a_fetch(1, 0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {  // position (k-slot, chain tile)
                const int ks = s >> 2, mt = s & 3;
                if (s + 2 < 8) aq[(s + 2) % 3] = a_fetch((s + 2) & 3, (s + 2) >> 2);
                if (s == 2) b_fetch(1);
                const double2 a = aq[s % 3];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const double av = p ? a.y : a.x;
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const double bv = p ? bg[ks & 1][g].y : bg[ks & 1][g].x;
                        acc[mt][p][g] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[mt][p][g], 0, 0, 0);
                    }
#pragma unroll
                    for (int g = 0; g < NBX; ++g) {
                        const double bv = p ? bq[ks & 1][g].y : bq[ks & 1][g].x;
                        accx[mt][p][g] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, accx[mt][p][g], 0, 0, 0);
                    }
                    if (GAPN > 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        asm volatile("s_nop %0" ::"n"(GAPN > 0 ? GAPN - 1 : 0));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (SYN_PRIO) __builtin_amdgcn_s_setprio(3);
        }
        if (bj == SB - 1) {  // the next stage's table is complete (every wave has waited for its own pieces); this stage's buffer may be refilled
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            lds_barrier();
        }
    }

    // (D of v_mfma_f64_16x16x4: register r of lane (ak = lane >> 4, ai = lane & 15) is row ak + 4 r, column ai; of the 4 x 4 x 4 blocks: lane
    // (ak, blk = (lane >> 2) & 3, j = lane & 3) is row 4 blk + ak, column j)
    // ---- ring modes of this (m, ring group): fold parities (and, spin 2, the two functions of a ring pair) in the lane, 32-byte stores ----
    double *fm = A.Fv + (long long)m * P.nrp_pad * A.rowlen;
    const double sgn = (SPIN == 2 && off) ? -1.0 : 1.0;
    if (SPIN == 0) {
        const int rp0 = (task.rb0 + 2 * w) * RBLK;
        auto put0 = [&](int rp, int unit, int c, double d0, double d1) __attribute__((always_inline)) {
            // lane c = 0 holds (N_re, S_re), lane c = 1 (N_im, S_im): after the exchange lane 0 has (N_re, N_im), lane 1 (S_re, S_im)
            const double N = d0 + d1, S = d0 - d1;
            const double other = __shfl_xor(c ? N : S, 1);
            // chains of ring pairs beyond the task / the plan carry zeros; rows exist up to nrp_pad
            if (rp < P.nrp_pad && (rp - task.rb0 * RBLK) < task.nrb * RBLK && unit < A.nunits)
                *reinterpret_cast<double2 *>(fm + (long long)rp * A.rowlen + unit * 4 + 2 * c) = c ? make_double2(other, S) : make_double2(N, other);
        };
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r) put0(rp0 + 16 * mt + ak + 4 * r, 8 * g + (ai >> 1), ai & 1, acc[mt][0][g][r], acc[mt][1][g][r]);
#pragma unroll
            for (int g = 0; g < NBX; ++g)
                put0(rp0 + 16 * mt + 4 * ((lane >> 2) & 3) + ak, 8 * NG + 2 * g + ((lane & 3) >> 1), lane & 1, accx[mt][0][g], accx[mt][1][g]);
        }
    } else {
        const int rp0 = (task.rb0 + w) * RBLK;
        const int q = lane & 3;
        // X: a+ columns: sum of the parities; a- columns: (-1)^off x their difference (see the header of the vector-unit kernel for the sums);
        // chains 0..31 are lambda+ of ring pairs 0..31 (X = P+_N | P-_S), chains 32..63 lambda- of the same ring pairs (X = P+_S | P-_N)
        auto put2 = [&](int rp, int unit, double p0, double p1, double m0, double m1) __attribute__((always_inline)) {
            const double Xp = q < 2 ? p0 + p1 : sgn * (p0 - p1), Xm = q < 2 ? m0 + m1 : sgn * (m0 - m1);
            const double Ym = __shfl_xor(Xm, 2), Yp = __shfl_xor(Xp, 2);
            // lane q = 0: (Q_N_re, Q_S_re, U_N_im, U_S_im); lane q = 1: (Q_N_im, Q_S_im, U_N_re, U_S_re); Q = (P+ + P-) / 2, U = (P+ - P-) / 2i
            const double us = (q & 1) ? 0.5 : -0.5;
            const double a0 = 0.5 * (Xp + Ym), a1 = 0.5 * (Xm + Yp), a2 = us * (Xp - Ym), a3 = us * (Xm - Yp);
            // exchange with lane q ^ 1: lane 0 keeps Q (sends its U halves, receives Q_im), lane 1 keeps U
            const double r0 = __shfl_xor((q & 1) ? a0 : a2, 1), r1 = __shfl_xor((q & 1) ? a1 : a3, 1);
            if (q < 2 && w < task.nrb && rp < P.nrp_pad && unit < A.nunits) {
                double *dst = fm + (long long)rp * A.rowlen + unit * 8 + 4 * q;  // (N_re, N_im, S_re, S_im) of Q (q = 0) / U (q = 1)
                if (q == 0) {
                    *reinterpret_cast<double2 *>(dst) = make_double2(a0, r0);
                    *reinterpret_cast<double2 *>(dst + 2) = make_double2(a1, r1);
                } else {
                    *reinterpret_cast<double2 *>(dst) = make_double2(a2, r0);
                    *reinterpret_cast<double2 *>(dst + 2) = make_double2(a3, r1);
                }
            }
        };
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    put2(rp0 + 16 * mt + ak + 4 * r, 4 * g + (ai >> 2), acc[mt][0][g][r], acc[mt][1][g][r], acc[mt + 2][0][g][r], acc[mt + 2][1][g][r]);
#pragma unroll
            for (int g = 0; g < NBX; ++g)
                put2(rp0 + 16 * mt + 4 * ((lane >> 2) & 3) + ak, 4 * NG + g, accx[mt][0][g], accx[mt][1][g], accx[mt + 2][0][g], accx[mt + 2][1][g]);
        }
    }
}

// ---- host ------------------------------------------------------------------------------------------------------------------------
struct SynShape {
    int ng, nbx;
};
// the smallest shape that holds `units` maps (8 per group, 2 per extra block) / fields (4 per group, 1 per extra block)
static SynShape syn_shape(int spin, int units)
{
    const int per_g = spin ? 4 : 8, per_x = spin ? 1 : 2;
    SynShape best = {2, 2};
    int best_cols = 1 << 30;
    for (int ng = 1; ng <= 2; ++ng)
        for (int nbx = 0; nbx <= 2; ++nbx)
            if (ng * per_g + nbx * per_x >= units && ng * 16 + nbx * 4 < best_cols) {
                best = {ng, nbx};
                best_cols = ng * 16 + nbx * 4;
            }
    return best;
}
int synth_duo_max_units(int spin) { return spin ? 10 : 20; }
int synth_duo_rowlen(int spin, int units) { return (spin ? 8 : 4) * units; }

template <int SPIN, int NG, int NBX>
static void launch_syn_t(unsigned grid, hipStream_t st, const SynDuoParams &A, const double2 *cn)
{
    hipLaunchKernelGGL((k_synth_duo<SPIN, NG, NBX>), dim3(grid), dim3(256), 0, st, A, cn);
}

// Fv (layout of this file's header) of `units` maps / fields whose alms start at d_alm.  d_tab: scratch of synth_duo_table_bytes().
size_t synth_duo_table_bytes(hx_plan *pl, int spin, int units)
{
    const SynShape sh = syn_shape(spin, units);
    long long nblk = 0;
    for (int m = 0; m <= pl->lmax; ++m) {
        const int l0 = spin == 0 ? m : std::max(m, 2);
        if (l0 <= pl->lmax) nblk += (pl->lmax - l0) / SLB + 1;
    }
    return (size_t)nblk * (sh.ng * 256 + sh.nbx * 64) * sizeof(double);
}

int launch_synth_duo(hx_plan *pl, int spin, int units, hx_plan::TaskSet &ts, const double2 *d_alm, double *d_tab, double *d_Fv)
{
    if (units < 1 || units > synth_duo_max_units(spin)) return fail(HX_ERR_ARG, "launch_synth_duo: %d units of spin %d", units, spin);
    hipStream_t st = rt().stream;
    const SynShape sh = syn_shape(spin, units);
    // first table block of every m (the same for every shape: blocks are counted, not bytes)
    DevBuf &boff = spin ? pl->syn_boff2 : pl->syn_boff0;
    if (!boff.p) {
        std::vector<long long> h(pl->lmax + 2, 0);
        for (int m = 0; m <= pl->lmax; ++m) {
            const int l0 = spin == 0 ? m : std::max(m, 2);
            h[m + 1] = h[m] + (l0 <= pl->lmax ? (pl->lmax - l0) / SLB + 1 : 0);
        }
        HX_TRY(upload(boff, h));
    }
    const double2 *cn = spin == 0 ? pl->cn0.as<double2>() : pl->cn2.as<double2>();
    const double *al = spin == 0 ? pl->al0.as<double>() : pl->al2.as<double>();
    const int rowlen = synth_duo_rowlen(spin, units);
    // (rows of pruned rings are not written: k_synth_spectrum_v is told where they begin, hx_plan::syn_mlim*)
    {
        ProfScope ps("synth_table");
        const dim3 grid(pl->lmax + 1, 8);
        if (spin == 0)
            hipLaunchKernelGGL(k_synth_table<0>, grid, dim3(256), 0, st, pl->lmax, d_alm, (long long)pl->nlm, units, al, sh.ng, sh.nbx, boff.as<long long>(), d_tab);
        else
            hipLaunchKernelGGL(k_synth_table<2>, grid, dim3(256), 0, st, pl->lmax, d_alm, (long long)pl->nlm, units, al, sh.ng, sh.nbx, boff.as<long long>(), d_tab);
    }
    SynDuoParams A;
    A.P = pl->dev(); A.tasks = ts.d_tasks.as<LegTask>(); A.tab = d_tab; A.boff = boff.as<long long>(); A.Fv = d_Fv; A.rowlen = rowlen; A.nunits = units;
    ProfScope ps("legendre_synthesis");
    ProfScope ps2("legendre_synth_duo");
    const unsigned grid = (unsigned)ts.tasks.size();
#define HX_SYN_CASE(S, G, X) if (spin == S && sh.ng == G && sh.nbx == X) launch_syn_t<S, G, X>(grid, st, A, cn); else
    HX_SYN_CASE(0, 1, 0) HX_SYN_CASE(0, 1, 1) HX_SYN_CASE(0, 1, 2) HX_SYN_CASE(0, 2, 0) HX_SYN_CASE(0, 2, 1) HX_SYN_CASE(0, 2, 2)
    HX_SYN_CASE(2, 1, 0) HX_SYN_CASE(2, 1, 1) HX_SYN_CASE(2, 1, 2) HX_SYN_CASE(2, 2, 0) HX_SYN_CASE(2, 2, 1) HX_SYN_CASE(2, 2, 2)
    return fail(HX_ERR_ARG, "launch_synth_duo: no kernel for spin %d shape (%d, %d)", spin, sh.ng, sh.nbx);
#undef HX_SYN_CASE
    HX_HIP(hipGetLastError());
    return HX_OK;
}

}  // namespace hx
