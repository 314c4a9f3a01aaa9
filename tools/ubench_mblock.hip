// ubench_mblock.hip -- the matrix block of k_legendre_duo in isolation: 16 slot pairs x 2 positions x (NG 16x16x4 + NBX 4x4x4_4b)
// with the A operands read from an LDS tile and, optionally, the B operand of position 1 made by quad_perm DPP moves (HALFB).
// Output: cycles per slot pair against the matrix-pipe time of its instructions.  ORDER: 0 big then small per position,
// 1 all bigs of the pair then the smalls, 2 smalls first.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_mblock.hip -o tools/bin/ubench_mblock
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
__device__ __forceinline__ double quad_rev(double v)
{
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), 0x1B, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), 0x1B, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double opaque_d(double v) { asm("; opaque" : "+v"(v)); return v; }
__device__ __forceinline__ double quad_rev_swz(double v)
{
    const int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x801B);  // quad-perm mode, lanes [3,2,1,0]
    const int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x801B);
    return __hiloint2double(hi, lo);
}
template <int MODE> __device__ __forceinline__ double rev(double v)
{
    if (MODE == 2) return quad_rev_swz(v);
    if (MODE == 3) return quad_rev(v);
    return quad_rev(opaque_d(v));
}

template <int NG, int NBX, int DPP, int ORDER, int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k_mb(double *out, unsigned long long *cyc, int iters, const double *__restrict__ src)
{
    __shared__ double tile[WAVES][2048];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, ai = lane & 15, ak = lane >> 4;
    for (int i = threadIdx.x; i < WAVES * 2048; i += 64 * WAVES) (&tile[0][0])[i] = src[i & 4095];
    __syncthreads();
    constexpr int NGA = NG > 0 ? NG : 1, NXA = NBX > 0 ? NBX : 1, NPB = DPP ? 1 : 2;
    double fr[16][NPB][NGA], frx[16][NPB][NXA];
#pragma unroll
    for (int sp = 0; sp < 16; ++sp)
#pragma unroll
        for (int pos = 0; pos < NPB; ++pos) {
#pragma unroll
            for (int g = 0; g < NGA; ++g) fr[sp][pos][g] = src[(lane * 37 + sp * 5 + pos + g * 3) & 4095];
#pragma unroll
            for (int g = 0; g < NXA; ++g) frx[sp][pos][g] = src[(lane * 41 + sp * 7 + pos + g) & 4095];
        }
    double4_t acc[NGA][2];
    double accx[NXA][2];
#pragma unroll
    for (int g = 0; g < NGA; ++g) acc[g][0] = acc[g][1] = (double4_t){0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < NXA; ++g) accx[g][0] = accx[g][1] = 0;
    const double *tw = &tile[w][0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        constexpr int PF = 3;
        if (DPP >= 2) {  // the operands pass an empty asm statement once per block: what is derived from them stays inside the loop
#pragma unroll
            for (int sp = 0; sp < 16; ++sp) {
#pragma unroll
                for (int g = 0; g < NGA; ++g) asm volatile("" : "+v"(fr[sp][0][g]));
#pragma unroll
                for (int g = 0; g < NXA; ++g) asm volatile("" : "+v"(frx[sp][0][g]));
            }
        }
        auto a_fetch = [&](int sp) __attribute__((always_inline)) {
            const int c = (sp & 1) * 32 + 4 * (sp >> 1) + ak;
            return *reinterpret_cast<const double2 *>(tw + c * 32 + ((ai ^ (c & 7)) * 2));
        };
        double2 aq[PF + 1];
#pragma unroll
        for (int j = 0; j < PF; ++j) aq[j] = a_fetch(j);
#pragma unroll
        for (int sp = 0; sp < 16; ++sp) {
            if (sp + PF < 16) aq[(sp + PF) % (PF + 1)] = a_fetch(sp + PF);
            double bg[2][NGA], bx[2][NXA];
            if (ORDER < 3)
#pragma unroll
            for (int pos = 0; pos < 2; ++pos) {
#pragma unroll
                for (int g = 0; g < NGA; ++g) bg[pos][g] = (DPP && pos) ? quad_rev(opaque_d(fr[sp][0][g])) : fr[sp][DPP ? 0 : pos][g];
#pragma unroll
                for (int g = 0; g < NXA; ++g) bx[pos][g] = (DPP && pos) ? quad_rev(opaque_d(frx[sp][0][g])) : frx[sp][DPP ? 0 : pos][g];
            }
            const double a0 = aq[sp % (PF + 1)].x, a1 = aq[sp % (PF + 1)].y;
            if (ORDER == 3 || ORDER == 4) {
                // first matrix instruction of the pair, then the DPP moves of ITS OWN position-1 operands in its shadow
                if (NG > 0) acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, fr[sp][0][0], acc[0][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                double cg[NGA], cx[NXA];
#pragma unroll
                for (int g = 0; g < NGA; ++g) cg[g] = rev<DPP>(fr[sp][0][g]);
#pragma unroll
                for (int g = 0; g < NXA; ++g) cx[g] = rev<DPP>(frx[sp][0][g]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 1; g < NG; ++g) acc[g][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, fr[sp][0][g], acc[g][0], 0, 0, 0);
                if (ORDER == 3) {
#pragma unroll
                    for (int g = 0; g < NBX; ++g) accx[g][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, frx[sp][0][g], accx[g][0], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) acc[g][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, cg[g], acc[g][1], 0, 0, 0);
                if (ORDER == 4) {
#pragma unroll
                    for (int g = 0; g < NBX; ++g) accx[g][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(a0, frx[sp][0][g], accx[g][0], 0, 0, 0);
                }
#pragma unroll
                for (int g = 0; g < NBX; ++g) accx[g][1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a1, cx[g], accx[g][1], 0, 0, 0);
            } else if (ORDER == 0) {
#pragma unroll
                for (int pos = 0; pos < 2; ++pos) {
                    const double a = pos ? a1 : a0;
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bg[pos][g], acc[g][pos], 0, 0, 0);
#pragma unroll
                    for (int g = 0; g < NBX; ++g) accx[g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bx[pos][g], accx[g][pos], 0, 0, 0);
                }
            } else if (ORDER == 1) {
#pragma unroll
                for (int pos = 0; pos < 2; ++pos)
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(pos ? a1 : a0, bg[pos][g], acc[g][pos], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pos = 0; pos < 2; ++pos)
#pragma unroll
                    for (int g = 0; g < NBX; ++g) accx[g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(pos ? a1 : a0, bx[pos][g], accx[g][pos], 0, 0, 0);
            } else {
#pragma unroll
                for (int pos = 0; pos < 2; ++pos)
#pragma unroll
                    for (int g = 0; g < NBX; ++g) accx[g][pos] = __builtin_amdgcn_mfma_f64_4x4x4f64(pos ? a1 : a0, bx[pos][g], accx[g][pos], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int pos = 0; pos < 2; ++pos)
#pragma unroll
                    for (int g = 0; g < NG; ++g) acc[g][pos] = __builtin_amdgcn_mfma_f64_16x16x4f64(pos ? a1 : a0, bg[pos][g], acc[g][pos], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double r = 0;
#pragma unroll
    for (int g = 0; g < NGA; ++g) for (int q = 0; q < 2; ++q) r += acc[g][q][0] + acc[g][q][1] + acc[g][q][2] + acc[g][q][3];
#pragma unroll
    for (int g = 0; g < NXA; ++g) r += accx[g][0] + accx[g][1];
    out[blockIdx.x * 64 * WAVES + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * WAVES + w] = t1 - t0;
}

template <int NG, int NBX, int DPP, int ORDER, int WAVES>
int run(const char *name)
{
    const int nb = 256, iters = 2000;
    double *out, *src;
    unsigned long long *cyc;
    CK(hipMalloc(&out, nb * 64 * WAVES * 8)); CK(hipMalloc(&src, 4096 * 8)); CK(hipMalloc(&cyc, nb * WAVES * 8));
    std::vector<double> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (double)rand() / RAND_MAX;
    CK(hipMemcpy(src, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_mb<NG, NBX, DPP, ORDER, WAVES>), dim3(nb), dim3(64 * WAVES), 0, 0, out, cyc, iters, src);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> c(nb * WAVES);
    CK(hipMemcpy(c.data(), cyc, nb * WAVES * 8, hipMemcpyDeviceToHost));
    std::sort(c.begin(), c.end());
    const double per = (double)c[c.size() / 2] / iters / 16.0, ideal = 2.0 * (NG * 64 + NBX * 16) * (WAVES / 4);
    printf("%-60s %7.1f cycles per slot pair (matrix pipe %4.0f)  %5.1f %%\n", name, per, ideal, 100.0 * ideal / per);
    (void)hipFree(out); (void)hipFree(src); (void)hipFree(cyc);
    return 0;
}

int main()
{
    run<1, 0, 0, 0, 4>("NG 1 NBX 0, operands in registers");
    run<1, 0, 1, 0, 4>("NG 1 NBX 0, DPP");
    run<1, 1, 0, 0, 4>("NG 1 NBX 1, operands in registers, big-small per position");
    run<1, 1, 1, 0, 4>("NG 1 NBX 1, DPP, big-small per position");
    run<1, 1, 0, 1, 4>("NG 1 NBX 1, operands in registers, bigs then smalls");
    run<1, 1, 1, 1, 4>("NG 1 NBX 1, DPP, bigs then smalls");
    run<1, 1, 1, 2, 4>("NG 1 NBX 1, DPP, smalls then bigs");
    run<2, 1, 0, 0, 4>("NG 2 NBX 1, operands in registers, big-small per position");
    run<2, 1, 1, 0, 4>("NG 2 NBX 1, DPP, big-small per position");
    run<2, 1, 1, 1, 4>("NG 2 NBX 1, DPP, bigs then smalls");
    run<2, 2, 1, 0, 4>("NG 2 NBX 2, DPP, big-small per position");
    run<2, 2, 1, 1, 4>("NG 2 NBX 2, DPP, bigs then smalls");
    run<2, 0, 1, 0, 4>("NG 2 NBX 0, DPP");
    run<1, 0, 1, 3, 4>("NG 1 NBX 0, DPP in the shadow of the first big");
    run<1, 1, 1, 3, 4>("NG 1 NBX 1, DPP in the shadow of the first big");
    run<1, 1, 1, 4, 4>("NG 1 NBX 1, DPP in the shadow, bigs then smalls");
    run<2, 1, 1, 3, 4>("NG 2 NBX 1, DPP in the shadow of the first big");
    run<2, 2, 1, 3, 4>("NG 2 NBX 2, DPP in the shadow of the first big");
    run<2, 2, 1, 4, 4>("NG 2 NBX 2, DPP in the shadow, bigs then smalls");
    run<2, 2, 1, 3, 8>("two waves per SIMD: NG 2 NBX 2, DPP in the shadow");
    run<1, 0, 3, 3, 4>("NG 1 NBX 0, DPP without the copy, in the shadow");
    run<1, 1, 3, 3, 4>("NG 1 NBX 1, DPP without the copy, in the shadow");
    run<2, 2, 3, 3, 4>("NG 2 NBX 2, DPP without the copy, in the shadow");
    run<1, 0, 2, 3, 4>("NG 1 NBX 0, ds_swizzle in the shadow");
    run<1, 1, 2, 3, 4>("NG 1 NBX 1, ds_swizzle in the shadow");
    run<2, 1, 2, 3, 4>("NG 2 NBX 1, ds_swizzle in the shadow");
    run<2, 2, 2, 3, 4>("NG 2 NBX 2, ds_swizzle in the shadow");
    run<2, 2, 2, 3, 8>("two waves per SIMD: NG 2 NBX 2, ds_swizzle in the shadow");
    run<1, 1, 2, 3, 8>("two waves per SIMD: NG 1 NBX 1, ds_swizzle in the shadow");
    run<0, 1, 0, 0, 4>("NG 0 NBX 1, operands in registers");
    run<0, 2, 0, 0, 4>("NG 0 NBX 2, operands in registers");
    run<1, 1, 1, 0, 8>("two waves per SIMD: NG 1 NBX 1, DPP, big-small");
    run<2, 2, 1, 0, 8>("two waves per SIMD: NG 2 NBX 2, DPP, big-small");
    run<1, 0, 1, 0, 8>("two waves per SIMD: NG 1 NBX 0, DPP");
    return 0;
}
