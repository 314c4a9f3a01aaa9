// ubench_rec.hip -- the recursion block of k_legendre_duo (32 steps of v' = (p' x + q') v - v_prev, wave-uniform (p', q') per step) for a wave that is
// alone on its SIMD: how the coefficient pair reaches the lanes decides the cost.  Output: cycles per 32-step block.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_rec.hip -o tools/bin/ubench_rec
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
template <int K> __device__ __forceinline__ double row_bcast(double v)
{
    double d;
    asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(d) : "v"(v), "n"(K));
    return d;
}
template <int K> __device__ __forceinline__ double row_bcast_fmac(double t, double p, double x)
{
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(t) : "v"(p), "v"(x), "n"(K));
    return t;
}
template <int K> __device__ __forceinline__ double lane_of(double v)  // value of lane K as a wave-uniform (scalar) double
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), K), hi = __builtin_amdgcn_readlane(__double2hiint(v), K);
    return __hiloint2double(hi, lo);
}

// MODE 0: coefficients already in registers (lower bound: 2 FMAs per step)   1: DPP row broadcast + fmac_dpp (the kernel)   2: LDS broadcast reads
//      3: v_readlane to scalars, three-op form (x v, q v - vp, p y + z)         4: as 0, two-step form (chain depth 1 per 2 steps)   5: as 1, two chains per lane
template <int MODE>
__global__ __launch_bounds__(256, 2) void k_rec(double *out, unsigned long long *cyc, int iters, const double2 *__restrict__ coef)
{
    __shared__ double2 cl_s[32];
    __shared__ double tile[4][2048];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x < 32) cl_s[threadIdx.x] = coef[threadIdx.x];
    __syncthreads();
    double vc = 1e-3 * (lane + 1), vp = 0.5e-3 * (lane + 1), xx = 0.3 + 1e-3 * lane;
    double vc2 = vc * 0.7, vp2 = vp * 0.7, xx2 = xx + 0.1;
    const double2 c0 = coef[lane & 15], c1 = coef[16 + (lane & 15)];
    double *tw = &tile[w][0];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            double cur[8], cur2[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                constexpr int dummy = 0; (void)dummy;
                const int kk = 8 * h + k;
                double tq = 0.0;
                if (MODE == 0 || MODE == 4) {
                    const double2 cc = cl_s[0];  // hoisted by the compiler: registers
                    tq = fma(cc.x + kk, xx, cc.y);
                } else if (MODE == 1 || MODE == 5) {
                    const double2 cs = kk < 16 ? c0 : c1;
                    switch (kk & 15) {
#define BC(K) case K: tq = row_bcast_fmac<K>(row_bcast<K>(cs.y), cs.x, xx); break;
                        BC(0) BC(1) BC(2) BC(3) BC(4) BC(5) BC(6) BC(7) BC(8) BC(9) BC(10) BC(11) BC(12) BC(13) BC(14) BC(15)
#undef BC
                    }
                } else if (MODE == 2) {
                    int idx = kk;
                    asm volatile("" : "+v"(idx));  // (not hoistable out of the block loop)
                    const double2 cc = cl_s[idx];
                    tq = fma(cc.x, xx, cc.y);
                }
                if (MODE == 3) {
                    const double2 cs = kk < 16 ? c0 : c1;
                    double sp = 0, sq = 0;
                    switch (kk & 15) {
#define RL(K) case K: sp = lane_of<K>(cs.x); sq = lane_of<K>(cs.y); break;
                        RL(0) RL(1) RL(2) RL(3) RL(4) RL(5) RL(6) RL(7) RL(8) RL(9) RL(10) RL(11) RL(12) RL(13) RL(14) RL(15)
#undef RL
                    }
                    cur[k] = vc;
                    const double y = xx * vc, z = fma(sq, vc, -vp);
                    const double vn = fma(sp, y, z);
                    vp = vc; vc = vn;
                } else if (MODE == 4 && (k & 1)) {
                    // two steps at once from (vc, vp): handled at the odd step
                    cur[k] = 0;  // filled below
                } else if (MODE == 4) {
                    const double2 cc = cl_s[0];
                    const double a0 = tq, a1 = fma(cc.x + kk + 1, xx, cc.y);
                    const double p = fma(a1, a0, -1.0), wq = a1 * vp;
                    const double v1 = fma(a0, vc, -vp), v2 = fma(p, vc, -wq);
                    cur[k] = vc; cur[k + 1] = v1;
                    vp = v1; vc = v2;
                } else {
                    cur[k] = vc;
                    const double vn = fma(tq, vc, -vp);
                    vp = vc; vc = vn;
                    if (MODE == 5) {
                        cur2[k] = vc2;
                        const double tq2 = tq + 0.1 * 1e-3;  // (the second ring of the lane has its own x: one more fma in the real thing)
                        const double vn2 = fma(fma(tq2, 1.0, 0.0), vc2, -vp2);
                        vp2 = vc2; vc2 = vn2;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                *reinterpret_cast<double2 *>(tw + lane * 32 + (((h * 4 + j) ^ (lane & 7)) * 2)) = make_double2(cur[2 * j], cur[2 * j + 1]);
                if (MODE == 5) *reinterpret_cast<double2 *>(tw + ((lane + 7) & 63) * 32 + (((h * 4 + j) ^ (lane & 7)) * 2)) = make_double2(cur2[2 * j], cur2[2 * j + 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        vc = vc * 1e-3 + 1e-4; vp = vp * 1e-3 + 2e-4; vc2 = vc2 * 1e-3 + 1e-4; vp2 = vp2 * 1e-3 + 3e-4;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = vc + vp + vc2 + vp2 + xx2 + tw[lane];
    if (lane == 0) cyc[blockIdx.x * 4 + w] = t1 - t0;
}

template <int MODE> int run(const char *name, int threads)
{
    const int nb = 256, iters = 4000;
    double *out; double2 *coef; unsigned long long *cyc;
    CK(hipMalloc(&out, nb * 256 * 8)); CK(hipMalloc(&coef, 64 * 16)); CK(hipMalloc(&cyc, nb * 4 * 8));
    std::vector<double2> h(64);
    for (int i = 0; i < 64; ++i) h[i] = make_double2(1e-3 * (i + 1), 0.9 + 1e-3 * i);
    CK(hipMemcpy(coef, h.data(), 64 * 16, hipMemcpyHostToDevice));
    CK(hipMemset(cyc, 0, nb * 4 * 8));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_rec<MODE>), dim3(nb), dim3(threads), 0, 0, out, cyc, iters, coef);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> c(nb * 4);
    CK(hipMemcpy(c.data(), cyc, nb * 4 * 8, hipMemcpyDeviceToHost));
    std::vector<double> v;
    for (auto x : c) if (x) v.push_back((double)x / iters);
    std::sort(v.begin(), v.end());
    printf("%-78s %7.0f cycles per 32-step block\n", name, v[v.size() / 2]);
    (void)hipFree(out); (void)hipFree(coef); (void)hipFree(cyc);
    return 0;
}

int main()
{
    run<0>("coefficients in registers (lower bound), one wave per SIMD", 256);
    run<1>("DPP row broadcast + fmac_dpp (k_legendre_duo), one wave per SIMD", 256);
    run<2>("LDS broadcast read per step, one wave per SIMD", 256);
    run<3>("v_readlane to scalars, three-op form, one wave per SIMD", 256);
    run<4>("two-step form (coefficients in registers), one wave per SIMD", 256);
    run<5>("DPP broadcast shared by two chains per lane (per chain-block: half), one wave per SIMD", 256);
    return 0;
}
