// ubench_mix.hip -- order of v_mfma_f64_16x16x4 and v_mfma_f64_4x4x4_4b in the matrix block of k_legendre_pipe<.,1,1>:
// 8 + 8 instructions per block (ideal 8 x 64 + 8 x 16 = 640 cycles), one wave per SIMD.
//   V0 slot-major (the kernel): 16a 4a 16b 4b ...      V1 grouped: 8 x 16 (a b a b ...), 8 x 4 (a b a b ...)
//   V2 grouped, four 4x4x4 accumulators in rotation     V3 pairs: 16a 16b 4a 4b ...
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mix.hip -o tools/bin/ubench_mix
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template <int V>
__global__ __launch_bounds__(256, 1) void k_mix(double *out, unsigned long long *cyc, int iters, const double *__restrict__ src)
{
    double a[8], b[8], bx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = src[(threadIdx.x * 8 + u) & 4095]; b[u] = src[(threadIdx.x * 8 + u + 77) & 4095]; bx[u] = src[(threadIdx.x * 8 + u + 177) & 4095]; }
    double4_t c[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    double x[4] = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (V == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                c[u & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], c[u & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                x[u & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], bx[u], x[u & 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if (V == 1 || V == 2) {
#pragma unroll
            for (int u = 0; u < 8; ++u) { c[u & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], c[u & 1], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int u = 0; u < 8; ++u) { x[V == 2 ? (u & 3) : (u & 1)] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], bx[u], x[V == 2 ? (u & 3) : (u & 1)], 0, 0, 0); __builtin_amdgcn_sched_barrier(0); }
        } else {
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], c[0], 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
                c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u + 1], b[u + 1], c[1], 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
                x[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u], bx[u], x[0], 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
                x[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[u + 1], bx[u + 1], x[1], 0, 0, 0); __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c[0][0] + c[1][1] + x[0] + x[1] + x[2] + x[3];
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
int run(const char *label, double *d_out, unsigned long long *d_cyc, const double *d_src, int cus)
{
    const int iters = 2000;
    for (int rep = 0; rep < 50; ++rep) hipLaunchKernelGGL((k_mix<V>), dim3(cus), dim3(256), 0, 0, d_out, d_cyc, iters, d_src);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(cus * 4);
    CK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * cus * 4, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-60s %7.1f cycles per block of 8 + 8 (ideal 640)\n", label, (double)h[h.size() / 2] / iters);
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
    if (run<0>("V0 slot-major 16a 4a 16b 4b", d_out, d_cyc, d_src, cus)) return 1;
    if (run<1>("V1 grouped 8 x 16 then 8 x 4 (two accumulators each)", d_out, d_cyc, d_src, cus)) return 1;
    if (run<2>("V2 grouped, 4x4x4 on four accumulators", d_out, d_cyc, d_src, cus)) return 1;
    if (run<3>("V3 pairs 16a 16b 4a 4b", d_out, d_cyc, d_src, cus)) return 1;
    return 0;
}
