// ubench_slot.hip -- cost of the ingredients of one "slot pair" of the pipelined Legendre kernel (hx_analysis.hip,
// k_legendre_pipe) for a wave that is alone on its SIMD: 2 x (v_mfma_f64_16x16x4 [+ v_mfma_f64_4x4x4_4b]) with, in their
// shadow, 128-bit LDS reads of the A operands / recursion coefficients, two recursion steps (FP64 FMAs) and a 128-bit
// tile store.  Ingredients are switched on one by one; output = shader cycles per pair (s_memtime), median over waves.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_slot.hip -o tools/bin/ubench_slot
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

// bits: 1 A reads, 2 coefficient reads, 4 tile store, 8 "pre" FMA, 16 chain FMA, 32 no sched barriers,
//       64 VALU behind ALL MFMAs of the position instead of behind the first
template <int NG, int NBX, int F, int PF>
__global__ __launch_bounds__(256, 1) void k_slot(double *out, unsigned long long *cyc, int iters, const double *__restrict__ src)
{
    __shared__ double tileA[4][2048], tileB[4][2048];
    __shared__ double2 coefs[64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, ai = lane & 15, ak = lane >> 4;
    for (int i = threadIdx.x; i < 4 * 2048; i += 256) { (&tileA[0][0])[i] = src[i & 4095]; (&tileB[0][0])[i] = src[(i + 99) & 4095]; }
    if (threadIdx.x < 128) (&coefs[0].x)[threadIdx.x] = src[threadIdx.x] * 1e-3 + 1.0;
    __syncthreads();
    double fr[16][2][NG > 0 ? NG : 1], frx[16][2][NBX > 0 ? NBX : 1];
#pragma unroll
    for (int sp = 0; sp < 16; ++sp)
#pragma unroll
        for (int pos = 0; pos < 2; ++pos) {
#pragma unroll
            for (int g = 0; g < (NG > 0 ? NG : 1); ++g) fr[sp][pos][g] = src[(lane * 37 + sp * 5 + pos + g * 3) & 4095];
#pragma unroll
            for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) frx[sp][pos][g] = src[(lane * 41 + sp * 7 + pos + g) & 4095];
        }
    double4_t acc[NG > 0 ? NG : 1][4];
    double accx[NBX > 0 ? NBX : 1][4];
#pragma unroll
    for (int g = 0; g < (NG > 0 ? NG : 1); ++g) acc[g][0] = acc[g][1] = acc[g][2] = acc[g][3] = (double4_t){0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) accx[g][0] = accx[g][1] = accx[g][2] = accx[g][3] = 0;
    double vc = src[lane] * 1e-3, vp = src[lane + 64] * 1e-3, tq = 0.9, xx = 0.3 + 1e-3 * lane;
    const double *tm = &tileA[w][0];
    double *tr = &tileB[w][0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        double2 aq[16], cq[17][2];
        auto fetch = [&](int sp) __attribute__((always_inline)) {
            if (F & 1) aq[sp] = *reinterpret_cast<const double2 *>(tm + (4 * (sp >> 1) + ak + (sp & 1) * 32) * 32 + ((ai ^ ((4 * (sp >> 1) + ak) & 7)) * 2));
            else aq[sp] = make_double2(vp, xx);
            if (F & 2) { cq[sp][0] = coefs[(lane >> 5) * 32 + 2 * sp]; cq[sp][1] = coefs[(lane >> 5) * 32 + 2 * sp + 1]; }
            else { cq[sp][0] = make_double2(1.0 + sp, 0.5); cq[sp][1] = make_double2(1.1 + sp, 0.4); }
        };
#pragma unroll
        for (int sp = 0; sp < PF; ++sp) fetch(sp);
        if (F & 256) {  // MFMAs only, per block of 4 slot pairs: the 8 16x16x4 first, then the 8 4x4x4
#pragma unroll
            for (int sp = PF; sp < 16; ++sp) fetch(sp);
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int sp = 4 * blk + (u >> 1), pos = u & 1;
                    const double a = pos ? aq[sp].y : aq[sp].x;
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[sp][pos][g], acc[g][pos], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int sp = 4 * blk + (u >> 1), pos = u & 1;
                    const double a = pos ? aq[sp].y : aq[sp].x;
                    const int AI = (F & 128) ? pos + 2 * (sp & 1) : pos;
#pragma unroll
                    for (int g = 0; g < NBX; ++g) accx[g][AI] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, frx[sp][pos][g], accx[g][AI], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            vc = vc * 1e-3 + 1e-4; vp = vp * 1e-3 + 2e-4;
            continue;
        }
#pragma unroll
        for (int sp = 0; sp < 16; ++sp) {
            if (sp + PF < 16) fetch(sp + PF);
            double2 o;
#pragma unroll
            for (int pos = 0; pos < 2; ++pos) {
                const double a = pos ? aq[sp].y : aq[sp].x;
                const int AI = (F & 128) ? pos + 2 * (sp & 1) : pos;
                if (NG > 0) acc[0][AI] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[sp][pos][0], acc[0][AI], 0, 0, 0);
                if (F & 64) {
#pragma unroll
                    for (int g = 1; g < NG; ++g) acc[g][AI] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[sp][pos][g], acc[g][AI], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NBX; ++g) accx[g][AI] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, frx[sp][pos][g], accx[g][AI], 0, 0, 0);
                }
                if (!(F & 32)) __builtin_amdgcn_sched_barrier(0);
                const double cur = vc;
                if (F & 16) { const double vn = fma(tq, vc, -vp); vp = vc; vc = vn; }
                if (pos) o.y = cur; else o.x = cur;
                if ((F & 4) && pos) *reinterpret_cast<double2 *>(tr + lane * 32 + ((sp ^ (lane & 7)) * 2)) = o;
                if (F & 8) {
                    const double2 cc = pos == 0 ? cq[sp][1] : cq[sp + 1 < 16 ? sp + 1 : sp][0];
                    tq = fma(cc.x, xx, cc.y);
                }
                if (!(F & 32)) __builtin_amdgcn_sched_barrier(0);
                if (!(F & 64)) {
#pragma unroll
                    for (int g = 1; g < NG; ++g) acc[g][AI] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[sp][pos][g], acc[g][AI], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NBX; ++g) accx[g][AI] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, frx[sp][pos][g], accx[g][AI], 0, 0, 0);
                }
                if (!(F & 32)) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // keep the chain bounded
        vc = vc * 1e-3 + 1e-4; vp = vp * 1e-3 + 2e-4;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = vc + vp + tq;
#pragma unroll
    for (int g = 0; g < (NG > 0 ? NG : 1); ++g) for (int q = 0; q < 4; ++q) r += acc[g][q][0] + acc[g][q][1] + acc[g][q][2] + acc[g][q][3];
#pragma unroll
    for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) r += accx[g][0] + accx[g][1] + accx[g][2] + accx[g][3];
    out[blockIdx.x * 256 + threadIdx.x] = r + tr[lane];
    if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}


// Per-pair VALU groups.  MODE 0: group behind the first MFMA of the pair, dependent chain (step a -> step b);
// MODE 1: the same with the two-step form (both new values from the two old ones: independent FMAs, u = t1 t0 - 1 prepared
// one pair ahead); MODE 2: as 1, group behind ALL MFMAs of the pair; MODE 3: as 1, one group per TWO pairs
template <int NG, int NBX, int MODE, int PF>
__global__ __launch_bounds__(256, 1) void k_pair(double *out, unsigned long long *cyc, int iters, const double *__restrict__ src)
{
    __shared__ double tileA[4][2048], tileB[4][2048];
    __shared__ double2 coefs[64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, ai = lane & 15, ak = lane >> 4;
    for (int i = threadIdx.x; i < 4 * 2048; i += 256) { (&tileA[0][0])[i] = src[i & 4095]; (&tileB[0][0])[i] = src[(i + 99) & 4095]; }
    if (threadIdx.x < 128) (&coefs[0].x)[threadIdx.x] = src[threadIdx.x] * 1e-3 + 1.0;
    __syncthreads();
    double fr[16][2][NG > 0 ? NG : 1], frx[16][2][NBX > 0 ? NBX : 1];
#pragma unroll
    for (int sp = 0; sp < 16; ++sp)
#pragma unroll
        for (int pos = 0; pos < 2; ++pos) {
#pragma unroll
            for (int g = 0; g < (NG > 0 ? NG : 1); ++g) fr[sp][pos][g] = src[(lane * 37 + sp * 5 + pos + g * 3) & 4095];
#pragma unroll
            for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) frx[sp][pos][g] = src[(lane * 41 + sp * 7 + pos + g) & 4095];
        }
    double4_t acc[NG > 0 ? NG : 1][2];
    double accx[NBX > 0 ? NBX : 1][2];
#pragma unroll
    for (int g = 0; g < (NG > 0 ? NG : 1); ++g) acc[g][0] = acc[g][1] = (double4_t){0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) accx[g][0] = accx[g][1] = 0;
    double vc = src[lane] * 1e-3, vp = src[lane + 64] * 1e-3, t0q = 0.9, t1q = 0.8, uq = -0.3, xx = 0.3 + 1e-3 * lane;
    const double *tm = &tileA[w][0];
    double *tr = &tileB[w][0];
    const unsigned long long ts = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        double2 aq[16], cq[17][2];
        auto fetch = [&](int sp) __attribute__((always_inline)) {
            aq[sp] = *reinterpret_cast<const double2 *>(tm + (4 * (sp >> 1) + ak + (sp & 1) * 32) * 32 + ((ai ^ ((4 * (sp >> 1) + ak) & 7)) * 2));
            cq[sp][0] = coefs[(lane >> 5) * 32 + 2 * sp];
            cq[sp][1] = coefs[(lane >> 5) * 32 + 2 * sp + 1];
        };
        auto valu = [&](int sp) __attribute__((always_inline)) {
            double2 o;
            if (MODE == 0) {
                o.x = vc;
                const double v1 = fma(t0q, vc, -vp);
                o.y = v1;
                const double v2 = fma(t1q, v1, -vc);
                vp = v1; vc = v2;
            } else {
                o.x = vc;
                const double v1 = fma(t0q, vc, -vp);
                const double v2 = fma(uq, vc, -(t1q * vp));
                o.y = v1;
                vp = v1; vc = v2;
            }
            *reinterpret_cast<double2 *>(tr + lane * 32 + ((sp ^ (lane & 7)) * 2)) = o;
            const int sn = sp + 1 < 16 ? sp + 1 : sp;
            t0q = fma(cq[sn][0].x, xx, cq[sn][0].y);
            t1q = fma(cq[sn][1].x, xx, cq[sn][1].y);
            if (MODE != 0) uq = fma(t1q, t0q, -1.0);
        };
#pragma unroll
        for (int sp = 0; sp < PF; ++sp) fetch(sp);
#pragma unroll
        for (int sp = 0; sp < 16; ++sp) {
            if (sp + PF < 16) fetch(sp + PF);
            if (NG > 0) acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[sp].x, fr[sp][0][0], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE <= 1) valu(sp);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 1; g < NG; ++g) acc[g][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[sp].x, fr[sp][0][g], acc[g][0], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NBX; ++g) accx[g][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(aq[sp].x, frx[sp][0][g], accx[g][0], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NG; ++g) acc[g][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[sp].y, fr[sp][1][g], acc[g][1], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < NBX; ++g) accx[g][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(aq[sp].y, frx[sp][1][g], accx[g][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (MODE == 2) valu(sp);
            if (MODE == 3 && (sp & 1)) { valu(sp - 1); valu(sp); }
            __builtin_amdgcn_sched_barrier(0);
        }
        vc = vc * 1e-3 + 1e-4; vp = vp * 1e-3 + 2e-4;
    }
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    double r = vc + vp + t0q + t1q + uq;
#pragma unroll
    for (int g = 0; g < (NG > 0 ? NG : 1); ++g) r += acc[g][0][0] + acc[g][0][1] + acc[g][0][2] + acc[g][0][3] + acc[g][1][0] + acc[g][1][1] + acc[g][1][2] + acc[g][1][3];
#pragma unroll
    for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) r += accx[g][0] + accx[g][1];
    out[blockIdx.x * 256 + threadIdx.x] = r + tr[lane];
    if (lane == 0) cyc[blockIdx.x * 4 + w] = te - ts;
}


// Block-wise: the FP64 recursion of 16 steps as ONE tight VALU block (coefficients already in registers, results kept in
// registers), then 8 pairs of MFMAs with ONLY LDS traffic in their shadow (stores of the 16 buffered values, reads of the
// next 16 coefficients and of the A operands).  HB = steps per VALU block (16 or 32).
template <int NG, int NBX, int HB, int PF>
__global__ __launch_bounds__(256, 1) void k_block(double *out, unsigned long long *cyc, int iters, const double *__restrict__ src)
{
    __shared__ double tileA[4][2048], tileB[4][2048];
    __shared__ double2 coefs[64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, ai = lane & 15, ak = lane >> 4;
    for (int i = threadIdx.x; i < 4 * 2048; i += 256) { (&tileA[0][0])[i] = src[i & 4095]; (&tileB[0][0])[i] = src[(i + 99) & 4095]; }
    if (threadIdx.x < 128) (&coefs[0].x)[threadIdx.x] = src[threadIdx.x] * 1e-3 + 1.0;
    __syncthreads();
    double fr[16][2][NG > 0 ? NG : 1], frx[16][2][NBX > 0 ? NBX : 1];
#pragma unroll
    for (int sp = 0; sp < 16; ++sp)
#pragma unroll
        for (int pos = 0; pos < 2; ++pos) {
#pragma unroll
            for (int g = 0; g < (NG > 0 ? NG : 1); ++g) fr[sp][pos][g] = src[(lane * 37 + sp * 5 + pos + g * 3) & 4095];
#pragma unroll
            for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) frx[sp][pos][g] = src[(lane * 41 + sp * 7 + pos + g) & 4095];
        }
    double4_t acc[NG > 0 ? NG : 1][2];
    double accx[NBX > 0 ? NBX : 1][2];
#pragma unroll
    for (int g = 0; g < (NG > 0 ? NG : 1); ++g) acc[g][0] = acc[g][1] = (double4_t){0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) accx[g][0] = accx[g][1] = 0;
    double vc = src[lane] * 1e-3, vp = src[lane + 64] * 1e-3, xx = 0.3 + 1e-3 * lane;
    const double *tm = &tileA[w][0];
    double *tr = &tileB[w][0];
    constexpr int NH = 32 / HB;           // VALU blocks per stage
    constexpr int PPB = 16 / NH;          // pairs per MFMA block
    double2 cq[HB];
#pragma unroll
    for (int k = 0; k < HB; ++k) cq[k] = coefs[(lane >> 5) * 32 + k];
    const unsigned long long ts = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            double cur[HB];
            // ---- VALU block ----
#pragma unroll
            for (int k = 0; k < HB; ++k) {
                cur[k] = vc;
                const double vn = fma(fma(cq[k].x, xx, cq[k].y), vc, -vp);
                vp = vc; vc = vn;
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- MFMA block with LDS traffic in its shadow ----
            double2 aq[PPB];
#pragma unroll
            for (int j = 0; j < PF; ++j) {
                const int sp = h * PPB + j;
                aq[j] = *reinterpret_cast<const double2 *>(tm + (4 * (sp >> 1) + ak + (sp & 1) * 32) * 32 + ((ai ^ ((4 * (sp >> 1) + ak) & 7)) * 2));
            }
#pragma unroll
            for (int j = 0; j < PPB; ++j) {
                const int sp = h * PPB + j;
                if (j + PF < PPB) {
                    const int sn = sp + PF;
                    aq[j + PF] = *reinterpret_cast<const double2 *>(tm + (4 * (sn >> 1) + ak + (sn & 1) * 32) * 32 + ((ai ^ ((4 * (sn >> 1) + ak) & 7)) * 2));
                }
#pragma unroll
                for (int pos = 0; pos < 2; ++pos) {
                    const double a = pos ? aq[j].y : aq[j].x;
                    if (NG > 0) acc[0][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[sp][pos][0], acc[0][pos], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (pos == 0) *reinterpret_cast<double2 *>(tr + lane * 32 + ((sp ^ (lane & 7)) * 2)) = make_double2(cur[2 * j], cur[2 * j + 1]);
                    else {  // coefficients of the next VALU block
                        cq[2 * j] = coefs[(lane >> 5) * 32 + ((2 * sp + HB) & 31)];
                        cq[2 * j + 1] = coefs[(lane >> 5) * 32 + ((2 * sp + 1 + HB) & 31)];
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 1; g < NG; ++g) acc[g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, fr[sp][pos][g], acc[g][pos], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NBX; ++g) accx[g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, frx[sp][pos][g], accx[g][pos], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        vc = vc * 1e-3 + 1e-4; vp = vp * 1e-3 + 2e-4;
    }
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    double r = vc + vp;
#pragma unroll
    for (int g = 0; g < (NG > 0 ? NG : 1); ++g) r += acc[g][0][0] + acc[g][0][1] + acc[g][0][2] + acc[g][0][3] + acc[g][1][0] + acc[g][1][1] + acc[g][1][2] + acc[g][1][3];
#pragma unroll
    for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) r += accx[g][0] + accx[g][1];
    out[blockIdx.x * 256 + threadIdx.x] = r + tr[lane];
    if (lane == 0) cyc[blockIdx.x * 4 + w] = te - ts;
}

template <int NG, int NBX, int HB, int PF>
int runb(const char *label, double *d_out, unsigned long long *d_cyc, const double *d_src, int cus)
{
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_block<NG, NBX, HB, PF>), dim3(cus), dim3(256), 0, 0, d_out, d_cyc, iters, d_src);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(cus * 4);
    CK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * cus * 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double ideal = 2.0 * (NG * 64.0 + NBX * 16.0);
    const double per = (double)h[h.size() / 2] / (iters * 16.0);
    printf("BLOCK NG %d NBX %d PF %d steps per VALU block %2d %-24s: %7.1f cycles per pair (MFMA alone %5.0f)  %5.1f %%\n", NG, NBX, PF, HB, label, per, ideal, 100.0 * ideal / per);
    fflush(stdout);
    return 0;
}


// MFMA-only, ordering experiment: within a block of PPB slot pairs, all MFMAs that accumulate into the SAME register run
// back to back (ORD 1: group g / position p major; ORD 0: pair major, as k_slot)
template <int NG, int NBX, int ORD, int PPB>
__global__ __launch_bounds__(256, 1) void k_order(double *out, unsigned long long *cyc, int iters, const double *__restrict__ src)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double fr[16][2][NG > 0 ? NG : 1], frx[16][2][NBX > 0 ? NBX : 1], aq[16][2];
#pragma unroll
    for (int sp = 0; sp < 16; ++sp)
#pragma unroll
        for (int pos = 0; pos < 2; ++pos) {
            aq[sp][pos] = src[(lane * 31 + sp * 3 + pos) & 4095];
#pragma unroll
            for (int g = 0; g < (NG > 0 ? NG : 1); ++g) fr[sp][pos][g] = src[(lane * 37 + sp * 5 + pos + g * 3) & 4095];
#pragma unroll
            for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) frx[sp][pos][g] = src[(lane * 41 + sp * 7 + pos + g) & 4095];
        }
    double4_t acc[NG > 0 ? NG : 1][2];
    double accx[NBX > 0 ? NBX : 1][2];
#pragma unroll
    for (int g = 0; g < (NG > 0 ? NG : 1); ++g) acc[g][0] = acc[g][1] = (double4_t){0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) accx[g][0] = accx[g][1] = 0;
    const unsigned long long ts = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int blk = 0; blk < 16 / PPB; ++blk) {
            if (ORD == 0) {
#pragma unroll
                for (int j = 0; j < PPB; ++j)
#pragma unroll
                    for (int pos = 0; pos < 2; ++pos) {
                        const int sp = blk * PPB + j;
#pragma unroll
                        for (int g = 0; g < NG; ++g) acc[g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[sp][pos], fr[sp][pos][g], acc[g][pos], 0, 0, 0);
#pragma unroll
                        for (int g = 0; g < NBX; ++g) accx[g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(aq[sp][pos], frx[sp][pos][g], accx[g][pos], 0, 0, 0);
                    }
            } else {
#pragma unroll
                for (int pos = 0; pos < 2; ++pos) {
#pragma unroll
                    for (int g = 0; g < NG; ++g)
#pragma unroll
                        for (int j = 0; j < PPB; ++j) {
                            const int sp = blk * PPB + j;
                            acc[g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(aq[sp][pos], fr[sp][pos][g], acc[g][pos], 0, 0, 0);
                        }
#pragma unroll
                    for (int g = 0; g < NBX; ++g)
#pragma unroll
                        for (int j = 0; j < PPB; ++j) {
                            const int sp = blk * PPB + j;
                            accx[g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(aq[sp][pos], frx[sp][pos][g], accx[g][pos], 0, 0, 0);
                        }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long te = __builtin_amdgcn_s_memtime();
    double r = 0;
#pragma unroll
    for (int g = 0; g < (NG > 0 ? NG : 1); ++g) r += acc[g][0][0] + acc[g][0][1] + acc[g][0][2] + acc[g][0][3] + acc[g][1][0] + acc[g][1][1] + acc[g][1][2] + acc[g][1][3];
#pragma unroll
    for (int g = 0; g < (NBX > 0 ? NBX : 1); ++g) r += accx[g][0] + accx[g][1];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 4 + w] = te - ts;
}

template <int NG, int NBX, int ORD, int PPB>
int runo(double *d_out, unsigned long long *d_cyc, const double *d_src, int cus)
{
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_order<NG, NBX, ORD, PPB>), dim3(cus), dim3(256), 0, 0, d_out, d_cyc, iters, d_src);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(cus * 4);
    CK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * cus * 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double ideal = 2.0 * (NG * 64.0 + NBX * 16.0);
    const double per = (double)h[h.size() / 2] / (iters * 16.0);
    printf("ORDER NG %d NBX %d  %s, blocks of %d pairs: %7.1f cycles per pair (MFMA alone %5.0f)  %5.1f %%\n", NG, NBX,
           ORD ? "accumulator-major" : "pair-major       ", PPB, per, ideal, 100.0 * ideal / per);
    fflush(stdout);
    return 0;
}

template <int NG, int NBX, int MODE, int PF>
int runp(const char *label, double *d_out, unsigned long long *d_cyc, const double *d_src, int cus)
{
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_pair<NG, NBX, MODE, PF>), dim3(cus), dim3(256), 0, 0, d_out, d_cyc, iters, d_src);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(cus * 4);
    CK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * cus * 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double ideal = 2.0 * (NG * 64.0 + NBX * 16.0);
    const double per = (double)h[h.size() / 2] / (iters * 16.0);
    printf("PAIR NG %d NBX %d PF %d mode %d %-44s: %7.1f cycles per pair (MFMA alone %5.0f)  %5.1f %%\n", NG, NBX, PF, MODE, label, per, ideal, 100.0 * ideal / per);
    fflush(stdout);
    return 0;
}

template <int NG, int NBX, int F, int PF>
int run(const char *label, double *d_out, unsigned long long *d_cyc, const double *d_src, int cus)
{
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_slot<NG, NBX, F, PF>), dim3(cus), dim3(256), 0, 0, d_out, d_cyc, iters, d_src);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(cus * 4);
    CK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * cus * 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double ideal = 2.0 * (NG * 64.0 + NBX * 16.0);
    const double per = (double)h[h.size() / 2] / (iters * 16.0);
    printf("NG %d NBX %d PF %d flags %3d %-44s: %7.1f cycles per pair (MFMA alone %5.0f)  %5.1f %%\n", NG, NBX, PF, F, label, per, ideal, 100.0 * ideal / per);
    fflush(stdout);
    return 0;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    double *d_out, *d_src; unsigned long long *d_cyc;
    CK(hipMalloc(&d_out, sizeof(double) * cus * 256)); CK(hipMalloc(&d_cyc, sizeof(unsigned long long) * cus * 4)); CK(hipMalloc(&d_src, sizeof(double) * 4096));
    std::vector<double> src(4096);
    srand(50);
    for (auto &v : src) v = (rand() / (double)RAND_MAX - 0.5);
    CK(hipMemcpy(d_src, src.data(), sizeof(double) * 4096, hipMemcpyHostToDevice));
#define R(NG, NBX, F, PF, L) if (run<NG, NBX, F, PF>(L, d_out, d_cyc, d_src, cus)) return 1
    R(1, 1, 0, 2, "MFMA only");
    R(1, 1, 256, 2, "MFMA only, 8 x 16x16x4 then 8 x 4x4x4 per block");
    R(1, 1, 384, 2, "the same, 4x4x4 on 4 accumulators");
    R(1, 1, 1, 2, "MFMA + A reads");
    R(1, 1, 257, 2, "grouped + A reads");
    if (getenv("ORDER_ONLY")) return 0;
    R(1, 1, 128, 2, "MFMA only, 4 accumulators");
    R(1, 0, 128, 2, "MFMA only, 4 accumulators");
    R(2, 0, 128, 2, "MFMA only, 4 accumulators");
    R(1, 1, 64, 2, "MFMA only, quad right behind");
    R(1, 1, 1, 2, "+ A reads");
    R(1, 1, 3, 2, "+ coefficient reads");
    R(1, 1, 7, 2, "+ tile store");
    R(1, 1, 15, 2, "+ pre FMA");
    R(1, 1, 31, 2, "+ chain FMA (all)");
    R(1, 1, 31, 4, "all, PF 4");
    R(1, 1, 63, 2, "all, no sched barriers");
    R(1, 1, 95, 2, "all, VALU behind all MFMAs of the position");
    R(1, 1, 16, 2, "chain FMA only");
    R(1, 1, 24, 2, "chain + pre FMA only");
    R(1, 1, 4, 2, "tile store only");
    R(1, 1, 2, 2, "coefficient reads only");
    R(1, 0, 0, 2, "MFMA only");
    R(1, 0, 31, 2, "all");
    R(2, 0, 0, 2, "MFMA only");
    R(2, 0, 31, 2, "all");
    R(2, 0, 95, 2, "all, VALU behind all MFMAs");
    R(0, 0, 31, 2, "recursion alone");
    R(0, 0, 31, 4, "recursion alone PF 4");
    R(0, 0, 24, 2, "FMAs alone");
    if (runo<1, 1, 0, 4>(d_out, d_cyc, d_src, cus) || runo<1, 1, 1, 4>(d_out, d_cyc, d_src, cus) || runo<1, 1, 1, 8>(d_out, d_cyc, d_src, cus) ||
        runo<1, 1, 1, 16>(d_out, d_cyc, d_src, cus) || runo<2, 0, 0, 4>(d_out, d_cyc, d_src, cus) || runo<2, 0, 1, 4>(d_out, d_cyc, d_src, cus) ||
        runo<1, 0, 0, 4>(d_out, d_cyc, d_src, cus) || runo<1, 0, 1, 4>(d_out, d_cyc, d_src, cus) || runo<0, 2, 0, 4>(d_out, d_cyc, d_src, cus) ||
        runo<0, 2, 1, 4>(d_out, d_cyc, d_src, cus)) return 1;
#define RB(NG, NBX, HB, PF, L) if (runb<NG, NBX, HB, PF>(L, d_out, d_cyc, d_src, cus)) return 1
    RB(1, 1, 16, 2, "");
    RB(1, 1, 32, 2, "");
    RB(1, 1, 16, 3, "");
    RB(1, 0, 16, 2, "");
    RB(2, 0, 16, 2, "");
    RB(2, 0, 32, 2, "");
    RB(0, 1, 16, 2, "");
    RB(0, 2, 16, 2, "");
#define RP(NG, NBX, M, PF, L) if (runp<NG, NBX, M, PF>(L, d_out, d_cyc, d_src, cus)) return 1
    RP(1, 1, 0, 2, "group per pair behind 1st MFMA, dependent");
    RP(1, 1, 1, 2, "group per pair behind 1st MFMA, two-step");
    RP(1, 1, 2, 2, "group per pair behind all MFMAs, two-step");
    RP(1, 1, 3, 2, "group per two pairs, two-step");
    RP(1, 0, 1, 2, "group per pair behind 1st MFMA, two-step");
    RP(2, 0, 1, 2, "group per pair behind 1st MFMA, two-step");
    RP(2, 0, 0, 2, "group per pair behind 1st MFMA, dependent");
    return 0;
}
