// Micro-benchmark: ceiling of no-return FP64 global atomics (global_atomic_add_f64) against plain stores, in the access pattern of the
// Legendre kernel's flush: a wave adds 4 rows of 16 consecutive doubles (128 B each); rows of one work-group walk an (l, column)
// span, work-groups own disjoint spans (one per m), every address is touched once per pass.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_atomic.hip -o gpurun_out/ubench_atomic && gpurun_out/ubench_atomic
// Prints GB/s of PAYLOAD (8 B per add) for: f64 atomic adds, plain stores, f64 atomic adds with ROWPAD doubles between rows (rows that
// straddle 128-byte lines), f32 atomic adds (the figure the microarchitecture guide quotes), and adds that all hit ONE line per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template <int MODE>  // 0 f64 atomic, 1 f64 store, 2 f32 atomic, 3 f64 atomic all lanes of a wave into one 128-byte line
__global__ __launch_bounds__(256) void k_add(double *out, long long rows_per_wg, int pcol, double v)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double *base = out + (long long)blockIdx.x * rows_per_wg * pcol;
    for (long long r = w * 4; r < rows_per_wg; r += 16) {
        double *p = base + (r + (lane >> 4)) * pcol + (lane & 15);
        if (MODE == 0) __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)p, v);
        else if (MODE == 1) *p = v;
        else if (MODE == 2) atomicAdd(reinterpret_cast<float *>(p), (float)v);
        else __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double *)(base + r * pcol + (lane & 15)), v);
    }
}

template <int MODE>
static int run(const char *name, double *d, long long rows_per_wg, int nwg, int pcol, double bytes_per_op)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int reps = 20;
    hipLaunchKernelGGL(k_add<MODE>, dim3(nwg), dim3(256), 0, 0, d, rows_per_wg, pcol, 1.0);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_add<MODE>, dim3(nwg), dim3(256), 0, 0, d, rows_per_wg, pcol, 1.0);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const double ops = (double)nwg * rows_per_wg * 16.0 * reps;
    printf("%-44s pcol %3d: %8.1f G ops/s, %7.1f GB/s payload\n", name, pcol, ops / (ms * 1e-3) / 1e9, ops * bytes_per_op / (ms * 1e-3) / 1e9);
    return 0;
}

int main()
{
    const int nwg = 6144;
    const long long rows = 4096;  // per work-group: 6144 x 4096 rows x 16 doubles = 3.2 GB per pass
    double *d;
    CK(hipMalloc(&d, sizeof(double) * (size_t)nwg * rows * 48));
    CK(hipMemset(d, 0, sizeof(double) * (size_t)nwg * rows * 48));
    if (run<0>("f64 atomic add, rows on 128-byte lines", d, rows, nwg, 16, 8)) return 1;
    if (run<0>("f64 atomic add, rows of a 48-double stride", d, rows, nwg, 48, 8)) return 1;
    if (run<0>("f64 atomic add, rows straddling lines", d, rows, nwg, 20, 8)) return 1;
    if (run<1>("f64 plain store, rows on 128-byte lines", d, rows, nwg, 16, 8)) return 1;
    if (run<1>("f64 plain store, rows straddling lines", d, rows, nwg, 20, 8)) return 1;
    if (run<2>("f32 atomic add (every other word)", d, rows, nwg, 16, 4)) return 1;
    if (run<3>("f64 atomic add, 4 lanes per address", d, rows, nwg, 16, 8)) return 1;
    return 0;
}
