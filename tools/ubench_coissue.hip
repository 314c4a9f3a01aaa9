// ubench_coissue.hip -- round 5: WHICH vector instructions of a second wave get through while the first wave of a SIMD streams FP64 matrix
// instructions?  (round 4 found FP64 vector FMAs starved completely: tools/ubench_duo.hip.)  Consumer: v_mfma_f64_16x16x4 back to back.
// Producer kinds: 0 FP64 FMA chain, 1 FP32 FMA chain, 2 packed FP32 FMA, 3 32-bit integer add / xor chain, 4 v_mov_b32_dpp quad_perm,
// 5 FP64 FMA independent x8, 6 FP32 FMA independent x8.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_coissue.hip -o tools/bin/ubench_coissue
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template <int KIND, int NOCONS, int PRIO>
__global__ __launch_bounds__(512, 1) void k_co(double *out, unsigned long long *cyc, int iters, const double *__restrict__ src)
{
    __shared__ volatile int stop;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    double r = 0.0;
    unsigned long long t0 = 0, t1 = 0, n = 0;
    if (w < 4) {
        double b[8];
        for (int i = 0; i < 8; ++i) b[i] = src[(lane * 7 + i) & 4095];
        double4_t acc[2] = {(double4_t){0, 0, 0, 0}, (double4_t){0, 0, 0, 0}};
        const double a = src[lane];
        t0 = __builtin_amdgcn_s_memtime();
        if (!NOCONS)
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[i], acc[i & 1], 0, 0, 0);
                n += 8;
            }
        t1 = __builtin_amdgcn_s_memtime();
        r = acc[0][0] + acc[0][1] + acc[1][2] + acc[1][3];
        if (lane == 0) stop = 1;
    } else {
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        double d0 = src[lane] * 1e-3, d1 = src[lane + 64] * 1e-3, dd[8];
        float f0 = (float)d0, f1 = (float)d1, ff[8];
        float2_t p0 = {f0, f1}, p1 = {f1, f0};
        unsigned u0 = lane * 2654435761u, u1 = lane + 17;
        for (int i = 0; i < 8; ++i) { dd[i] = d0 + i; ff[i] = f0 + i; }
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; NOCONS ? it < iters : !stop; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (KIND == 0) { const double v = fma(d0, 0.999, -d1); d1 = d0; d0 = v; }
                if (KIND == 1) { const float v = fmaf(f0, 0.999f, -f1); f1 = f0; f0 = v; }
                if (KIND == 2) { asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p0) : "v"(p1), "v"(p0)); }
                if (KIND == 3) { u0 = (u0 + u1) ^ (u0 >> 3); }
                if (KIND == 4) { asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u0)); }
                if (KIND == 5) { dd[i & 7] = fma(dd[i & 7], 0.999, 1e-3); }
                if (KIND == 6) { ff[i & 7] = fmaf(ff[i & 7], 0.999f, 1e-3f); }
            }
            __builtin_amdgcn_sched_barrier(0);
            n += 16;
            if ((it & 255) == 255) { d0 = d0 * 1e-3 + 1e-4; d1 = d1 * 1e-3 + 2e-4; f0 = f0 * 1e-3f + 1e-4f; f1 = f1 * 1e-3f + 2e-4f; }
        }
        t1 = __builtin_amdgcn_s_memtime();
        r = d0 + d1 + f0 + f1 + p0[0] + p0[1] + (double)u0;
        for (int i = 0; i < 8; ++i) r += dd[i] + ff[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) { cyc[(blockIdx.x * 8 + w) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + w) * 2 + 1] = n; }
}

template <int KIND, int NOCONS, int PRIO>
int run(const char *name, int iters)
{
    const int nb = 256;
    double *out, *src;
    unsigned long long *cyc;
    CK(hipMalloc(&out, nb * 512 * 8)); CK(hipMalloc(&src, 4096 * 8)); CK(hipMalloc(&cyc, nb * 16 * 8));
    std::vector<double> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (double)rand() / RAND_MAX;
    CK(hipMemcpy(src, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_co<KIND, NOCONS, PRIO>), dim3(nb), dim3(512), 0, 0, out, cyc, iters, src);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> c(nb * 16);
    CK(hipMemcpy(c.data(), cyc, nb * 16 * 8, hipMemcpyDeviceToHost));
    std::vector<double> cons, prod;
    for (int b = 0; b < nb; ++b)
        for (int w = 0; w < 8; ++w) {
            const double cy = (double)c[(b * 8 + w) * 2], n = (double)c[(b * 8 + w) * 2 + 1];
            if (n > 0) (w < 4 ? cons : prod).push_back(cy / n);
        }
    std::sort(cons.begin(), cons.end()); std::sort(prod.begin(), prod.end());
    printf("%-44s consumer %7.1f cycles per 16x16x4 | producer %8.1f cycles per instruction\n", name, cons.empty() ? 0.0 : cons[cons.size() / 2],
           prod.empty() ? 0.0 : prod[prod.size() / 2]);
    hipFree(out); hipFree(src); hipFree(cyc);
    return 0;
}

int main()
{
    const int it = 20000;
    run<0, 1, 0>("alone: FP64 FMA chain", it);
    run<1, 1, 0>("alone: FP32 FMA chain", it);
    run<2, 1, 0>("alone: packed FP32 FMA chain", it);
    run<3, 1, 0>("alone: integer add/xor chain", it);
    run<4, 1, 0>("alone: v_mov_b32_dpp chain", it);
    run<5, 1, 0>("alone: FP64 FMA x8 independent", it);
    run<6, 1, 0>("alone: FP32 FMA x8 independent", it);
    run<0, 0, 1>("under FP64 MFMA stream: FP64 FMA chain", it);
    run<1, 0, 1>("under FP64 MFMA stream: FP32 FMA chain", it);
    run<2, 0, 1>("under FP64 MFMA stream: packed FP32 FMA", it);
    run<3, 0, 1>("under FP64 MFMA stream: integer chain", it);
    run<4, 0, 1>("under FP64 MFMA stream: v_mov_b32_dpp", it);
    run<5, 0, 1>("under FP64 MFMA stream: FP64 FMA x8 indep", it);
    run<6, 0, 1>("under FP64 MFMA stream: FP32 FMA x8 indep", it);
    run<1, 0, 0>("under FP64 MFMA stream: FP32 chain, no prio", it);
    return 0;
}
