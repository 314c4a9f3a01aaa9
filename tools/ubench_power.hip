// ubench_power.hip -- what the FP64 matrix pipe of this MI355X sustains as a function of its duty cycle,
// of the instruction shape (16x16x4 vs four 4x4x4_4b with the A operand rotated by DPP) and of what runs
// beside it (FP64 VALU chains + LDS stores, as the Legendre recursion does).  For every variant:
// wall TFLOP/s of the MFMA work, shader ticks per MFMA (s_memtime) and the in-kernel clock
// (s_memtime / s_memrealtime x 100 MHz), after >= 0.4 s of back-to-back launches (DVFS settled).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_power.hip -o gpurun_out/ubench_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

__device__ inline double rot_row(double v, int ctrl_sel)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    if (ctrl_sel == 1) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x124, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x124, 0xf, 0xf, false); }
    if (ctrl_sel == 2) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x128, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x128, 0xf, 0xf, false); }
    if (ctrl_sel == 3) { lo = __builtin_amdgcn_update_dpp(lo, lo, 0x12c, 0xf, 0xf, false); hi = __builtin_amdgcn_update_dpp(hi, hi, 0x12c, 0xf, 0xf, false); }
    return __hiloint2double(hi, lo);
}

struct Stamp { unsigned long long t0, t1, r0, r1; };

// MODE 0: 16x16x4 stream, NMF MFMAs then SLEEP x s_sleep 1 (64 clk each) per iteration
// MODE 1: the same work as 4 x 4x4x4_4b per 16x16x4 with DPP-rotated A
// MODE 2: pure 4x4x4_4b stream (no rotation)
// MODE 3: 16x16x4 + VALU/LDS "recursion" in the SAME wave between MFMA groups (NV fma pairs + stores)
// MODE 4: waves >= NWM run the recursion-like VALU + LDS-store loop, waves < NWM the 16x16x4 stream
// MODE 5: as 4 with the 4x4x4 form
template <int MODE, int NMF, int SLEEP>
__global__ __launch_bounds__(512) void k_duty(double *out, Stamp *st, int iters, int nwm, const double *__restrict__ src)
{
    __shared__ double lds[8][16 * 64];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = src[(threadIdx.x * 8 + u) & 4095]; b[u] = src[(threadIdx.x * 8 + u + 77) & 4095]; }
    double4_t c[4];
    double cq[16];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = (double4_t){0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < 16; ++u) cq[u] = 0.0;
    double v0 = a[0], v1 = a[1], p0 = a[2], p1 = a[3];
    const double x = 0.3 + 1e-3 * lane;
    Stamp s;
    s.t0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime();
    const bool mf_wave = (MODE < 4) || w < nwm;
    for (int i = 0; i < iters; ++i) {
        if (mf_wave) {
#pragma unroll
            for (int u = 0; u < NMF; ++u) {
                if (MODE == 0 || MODE == 3 || MODE == 4)
                    c[u & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u & 7], b[u & 7], c[u & 3], 0, 0, 0);
                else if (MODE == 1 || MODE == 5) {
                    const double a0 = a[u & 7];
                    const double a1 = rot_row(a0, 1), a2 = rot_row(a0, 2), a3 = rot_row(a0, 3);
                    cq[(u & 3) * 4 + 0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, b[u & 7], cq[(u & 3) * 4 + 0], 0, 0, 0);
                    cq[(u & 3) * 4 + 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, b[u & 7], cq[(u & 3) * 4 + 1], 0, 0, 0);
                    cq[(u & 3) * 4 + 2] = __builtin_amdgcn_mfma_f64_4x4x4f64(a2, b[u & 7], cq[(u & 3) * 4 + 2], 0, 0, 0);
                    cq[(u & 3) * 4 + 3] = __builtin_amdgcn_mfma_f64_4x4x4f64(a3, b[u & 7], cq[(u & 3) * 4 + 3], 0, 0, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        cq[(u & 3) * 4 + r] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u & 7], b[u & 7], cq[(u & 3) * 4 + r], 0, 0, 0);
                }
            }
            if (MODE == 3) {
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    lds[w][u * 64 + (lane ^ u)] = v0;
                    const double n0 = fma(fma(a[4], x, b[4]), v0, -p0);
                    p0 = v0; v0 = n0;
                }
            }
#pragma unroll
            for (int u = 0; u < SLEEP; ++u) __builtin_amdgcn_s_sleep(1);
        } else {
            // recursion-like: two chains per lane, 16 steps, tile stores
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                lds[w][u * 64 + (lane ^ u)] = v0 + v1;
                const double n0 = fma(fma(a[4], x, b[4]), v0, -p0), n1 = fma(fma(a[5], x, b[5]), v1, -p1);
                p0 = v0; v0 = n0; p1 = v1; v1 = n1;
            }
        }
    }
    s.t1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
    double r = v0 + v1 + p0 + p1 + lds[w][lane];
#pragma unroll
    for (int u = 0; u < 4; ++u) r += c[u][0] + c[u][1] + c[u][2] + c[u][3];
#pragma unroll
    for (int u = 0; u < 16; ++u) r += cq[u];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) st[blockIdx.x * 8 + w] = s;
}

// round 1's probe (hx_measure_peaks of that round, tools/ubench_fp64.hip): ONE operand pair for every MFMA, 4 accumulators,
// 4 MFMAs per loop iteration.  SAMEOP 1: as it was; 0: the operands rotate through 4 register pairs
template <int SAMEOP>
__global__ __launch_bounds__(256) void k_r1probe(double *out, Stamp *st, int iters, double seed)
{
    double4_t c[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) c[u] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double ma[4], mb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { ma[u] = seed * 1e-3 + threadIdx.x * 1e-6 + (SAMEOP ? 0 : u * 1e-4); mb[u] = 1.0 + threadIdx.x * 1e-7 + (SAMEOP ? 0 : u * 1e-5); }
    Stamp s;
    s.t0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) c[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma[SAMEOP ? 0 : u], mb[SAMEOP ? 0 : u], c[u], 0, 0, 0);
    }
    s.t1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
    if ((threadIdx.x & 63) == 0) st[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 2047] = s;
}

template <int SAMEOP>
int run_r1(const char *label, int bpc, double *d_out, Stamp *d_st, int cus)
{
    const int iters = 20000, blocks = cus * bpc;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f; double total = 0.0;
    for (int reps = 0; total < 400.0 && reps < 200; ++reps) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_r1probe<SAMEOP>), dim3(blocks), dim3(256), 0, 0, d_out, d_st, iters, 1.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    std::vector<Stamp> h(2048);
    CK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * 2048, hipMemcpyDeviceToHost));
    std::vector<double> ghz, ticks;
    for (auto &s : h) if (s.r1 > s.r0) { ghz.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1); ticks.push_back((double)(s.t1 - s.t0) / (4.0 * iters)); }
    std::sort(ghz.begin(), ghz.end()); std::sort(ticks.begin(), ticks.end());
    printf("%-44s waves/SIMD %d: %8.3f ms  %6.1f TF  %6.1f ticks per MFMA per wave  clock %.2f GHz\n", label, bpc, ms,
           (double)blocks * 4 * iters * 4.0 * 2048.0 / ms * 1e-9, ticks[ticks.size() / 2], ghz[ghz.size() / 2]);
    fflush(stdout);
    return 0;
}

// pure 16x16x4 stream on NACC accumulators in rotation (dependent-accumulate latency): 64 MFMAs per loop iteration
template <int NACC>
__global__ __launch_bounds__(256) void k_nacc(double *out, Stamp *st, int iters, const double *__restrict__ src)
{
    double4_t c[NACC];
#pragma unroll
    for (int u = 0; u < NACC; ++u) c[u] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = src[(threadIdx.x * 8 + u) & 4095]; b[u] = src[(threadIdx.x * 8 + u + 77) & 4095]; }
    Stamp s;
    s.t0 = __builtin_amdgcn_s_memtime(); s.r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) c[u % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u & 7], b[(u >> 3) & 7], c[u % NACC], 0, 0, 0);
    }
    s.t1 = __builtin_amdgcn_s_memtime(); s.r1 = __builtin_amdgcn_s_memrealtime();
    double r = 0;
#pragma unroll
    for (int u = 0; u < NACC; ++u) r += c[u][0] + c[u][1] + c[u][2] + c[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) st[(blockIdx.x * 4 + (threadIdx.x >> 6)) & 2047] = s;
}

template <int NACC>
int run_nacc(double *d_out, Stamp *d_st, const double *d_src, int cus)
{
    const int iters = 1000;
    for (int rep = 0; rep < 100; ++rep) hipLaunchKernelGGL((k_nacc<NACC>), dim3(cus), dim3(256), 0, 0, d_out, d_st, iters, d_src);
    CK(hipDeviceSynchronize());
    std::vector<Stamp> h(2048);
    CK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * 2048, hipMemcpyDeviceToHost));
    std::vector<double> ticks;
    for (int i = 0; i < cus * 4 && i < 2048; ++i) ticks.push_back((double)(h[i].t1 - h[i].t0) / (64.0 * iters));
    std::sort(ticks.begin(), ticks.end());
    printf("16x16x4 stream on %d accumulator(s) in rotation, one wave per SIMD: %6.1f ticks per MFMA\n", NACC, ticks[ticks.size() / 2]);
    fflush(stdout);
    return 0;
}

struct Res { double tf, ticks_per_mfma, ghz, ms; };

template <int MODE, int NMF, int SLEEP>
int run(const char *label, int wpb, int nwm, double *d_out, Stamp *d_st, const double *d_src, int cus)
{
    const int blocks = cus, iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // settle DVFS: ~0.4 s of launches
    float ms = 0.f;
    int reps = 0;
    double total = 0.0;
    while (total < 400.0 && reps < 2000) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k_duty<MODE, NMF, SLEEP>), dim3(blocks), dim3(wpb * 64), 0, 0, d_out, d_st, iters, nwm, d_src);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms; ++reps;
    }
    std::vector<Stamp> h(blocks * 8);
    CK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * blocks * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz, ticks;
    const int nmw = MODE < 4 ? wpb : nwm;
    for (int bI = 0; bI < blocks; ++bI) for (int w = 0; w < nmw; ++w) {
        const Stamp &s = h[bI * 8 + w];
        if (s.r1 > s.r0) { ghz.push_back((double)(s.t1 - s.t0) / (double)(s.r1 - s.r0) * 0.1); ticks.push_back((double)(s.t1 - s.t0) / ((double)iters * NMF)); }
    }
    std::sort(ghz.begin(), ghz.end()); std::sort(ticks.begin(), ticks.end());
    const double flops = (double)blocks * nmw * iters * NMF * 2048.0;
    printf("%-44s waves/CU %2d (mfma waves %2d): %8.3f ms  %6.1f TF  %6.1f ticks/16x16x4-equiv  clock %.2f GHz\n", label, wpb, nmw, ms,
           flops / ms * 1e-9, ticks.empty() ? 0.0 : ticks[ticks.size() / 2], ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
    fflush(stdout);
    return 0;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("device %s CUs %d\n", p.name, cus);
    double *d_out, *d_src; Stamp *d_st;
    CK(hipMalloc(&d_out, sizeof(double) * cus * 512)); CK(hipMalloc(&d_st, sizeof(Stamp) * (cus * 8 > 2048 ? cus * 8 : 2048))); CK(hipMalloc(&d_src, sizeof(double) * 4096));
    std::vector<double> src(4096);
    srand(50);
    for (auto &v : src) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
    CK(hipMemcpy(d_src, src.data(), sizeof(double) * 4096, hipMemcpyHostToDevice));
    if (run_nacc<1>(d_out, d_st, d_src, cus) || run_nacc<2>(d_out, d_st, d_src, cus) || run_nacc<3>(d_out, d_st, d_src, cus) ||
        run_nacc<4>(d_out, d_st, d_src, cus) || run_nacc<8>(d_out, d_st, d_src, cus)) return 1;
    for (int bpc : {1, 2, 4}) {
        if (run_r1<1>("round-1 probe: one operand pair", bpc, d_out, d_st, cus)) return 1;
        if (run_r1<0>("round-1 probe: rotating operand pairs", bpc, d_out, d_st, cus)) return 1;
    }
    for (int wpb : {4, 8}) {
        if (run<0, 16, 0>("16x16x4 stream", wpb, 0, d_out, d_st, d_src, cus)) return 1;
        if (run<1, 16, 0>("4x(4x4x4_4b) + DPP-rotated A", wpb, 0, d_out, d_st, d_src, cus)) return 1;
        if (run<2, 16, 0>("4x4x4_4b stream (no rotation)", wpb, 0, d_out, d_st, d_src, cus)) return 1;
    }
    // duty cycle: 16 MFMAs (1024 clk) + SLEEP x 64 clk idle, one wave per SIMD
    if (run<0, 16, 4>("16x16x4, 16 mfma + 4 sleeps (~80%)", 4, 0, d_out, d_st, d_src, cus)) return 1;
    if (run<0, 16, 8>("16x16x4, 16 mfma + 8 sleeps (~67%)", 4, 0, d_out, d_st, d_src, cus)) return 1;
    if (run<0, 16, 16>("16x16x4, 16 mfma + 16 sleeps (~50%)", 4, 0, d_out, d_st, d_src, cus)) return 1;
    if (run<1, 16, 4>("4x4x4 rot, 16 + 4 sleeps", 4, 0, d_out, d_st, d_src, cus)) return 1;
    if (run<1, 16, 8>("4x4x4 rot, 16 + 8 sleeps", 4, 0, d_out, d_st, d_src, cus)) return 1;
    if (run<1, 16, 16>("4x4x4 rot, 16 + 16 sleeps", 4, 0, d_out, d_st, d_src, cus)) return 1;
    // MFMA + recursion in the same wave
    if (run<3, 16, 0>("16x16x4 + recursion/LDS same wave", 4, 0, d_out, d_st, d_src, cus)) return 1;
    if (run<3, 16, 0>("16x16x4 + recursion/LDS same wave", 8, 0, d_out, d_st, d_src, cus)) return 1;
    // 4 MFMA waves + 4 recursion waves per CU
    if (run<4, 16, 0>("16x16x4 waves beside recursion waves", 8, 4, d_out, d_st, d_src, cus)) return 1;
    if (run<5, 16, 0>("4x4x4 rot waves beside recursion waves", 8, 4, d_out, d_st, d_src, cus)) return 1;
    return 0;
}
