// ubench_accreg.hip -- does it matter where the accumulators of v_mfma_f64_16x16x4 live (VGPR / AGPR) and how many rotate?
// One wave per SIMD, 64 MFMAs per loop iteration, operands from 8 + 8 VGPR pairs.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_accreg.hip -o tools/bin/ubench_accreg
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template <int NACC, int AG>
__global__ __launch_bounds__(256, 1) void k(double *out, unsigned long long *cyc, int iters, const double *__restrict__ src)
{
    double4_t c[NACC];
#pragma unroll
    for (int u = 0; u < NACC; ++u) c[u] = (double4_t){0.0, 0.0, 0.0, 0.0};
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = src[(threadIdx.x * 8 + u) & 4095]; b[u] = src[(threadIdx.x * 8 + u + 77) & 4095]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (AG >= 2) asm volatile("" : "+v"(a[u]), "+a"(b[u]));
        else asm volatile("" : "+v"(a[u]), "+v"(b[u]));
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) {
            if (AG == 1) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c[u % NACC]) : "v"(a[u & 7]), "v"(b[(u >> 3) & 7]));
            else if (AG == 2) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c[u % NACC]) : "v"(a[u & 7]), "a"(b[(u >> 3) & 7]));  // B operand from an AGPR
            else if (AG == 3) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(c[u % NACC]) : "v"(a[u & 7]), "a"(b[(u >> 3) & 7]));
            else asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(c[u % NACC]) : "v"(a[u & 7]), "v"(b[(u >> 3) & 7]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0;
#pragma unroll
    for (int u = 0; u < NACC; ++u) r += c[u][0] + c[u][1] + c[u][2] + c[u][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, int AG>
int run(double *d_out, unsigned long long *d_cyc, const double *d_src, int cus)
{
    const int iters = 1000;
    for (int rep = 0; rep < 50; ++rep) hipLaunchKernelGGL((k<NACC, AG>), dim3(cus), dim3(256), 0, 0, d_out, d_cyc, iters, d_src);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(cus * 4);
    CK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * cus * 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%d accumulator(s) in %s: %6.1f cycles per v_mfma_f64_16x16x4\n", NACC, AG == 0 ? "VGPRs, B in VGPRs" : (AG == 1 ? "AGPRs, B in VGPRs" : (AG == 2 ? "VGPRs, B in AGPRs" : "AGPRs, B in AGPRs")), (double)h[h.size() / 2] / (64.0 * iters));
    return 0;
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    double *d_out, *d_src; unsigned long long *d_cyc;
    CK(hipMalloc(&d_out, sizeof(double) * cus * 256)); CK(hipMalloc(&d_src, sizeof(double) * 4096)); CK(hipMalloc(&d_cyc, sizeof(unsigned long long) * cus * 4));
    std::vector<double> src(4096);
    srand(5); for (auto &v : src) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
    CK(hipMemcpy(d_src, src.data(), sizeof(double) * 4096, hipMemcpyHostToDevice));
    if (run<1, 0>(d_out, d_cyc, d_src, cus) || run<2, 0>(d_out, d_cyc, d_src, cus) || run<4, 0>(d_out, d_cyc, d_src, cus) || run<8, 0>(d_out, d_cyc, d_src, cus)) return 1;
    if (run<1, 1>(d_out, d_cyc, d_src, cus) || run<2, 1>(d_out, d_cyc, d_src, cus) || run<4, 1>(d_out, d_cyc, d_src, cus) || run<8, 1>(d_out, d_cyc, d_src, cus)) return 1;
    if (run<2, 2>(d_out, d_cyc, d_src, cus) || run<4, 2>(d_out, d_cyc, d_src, cus) || run<8, 2>(d_out, d_cyc, d_src, cus)) return 1;
    if (run<2, 3>(d_out, d_cyc, d_src, cus) || run<4, 3>(d_out, d_cyc, d_src, cus) || run<8, 3>(d_out, d_cyc, d_src, cus)) return 1;
    return 0;
}
