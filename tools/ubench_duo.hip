// ubench_duo.hip -- two waves on one SIMD: a CONSUMER that streams FP64 matrix instructions and a PRODUCER that runs FP64 vector
// FMAs in different arrangements (dependent chain / bursts of independent ones / with s_setprio).  What does the producer get per
// matrix instruction, and what does it cost the consumer?  (round 4: design input for k_legendre_duo / wave specialisation)
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_duo.hip -o tools/bin/ubench_duo
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

// 512 threads = 8 waves: waves 0..3 consumers (one per SIMD), waves 4..7 producers (one per SIMD).
// MODE: 0 dependent pairs (t = fma(c, x, d); v = fma(t, v, -p)), 1 bursts: 8 independent t's, then the 8 chain steps, 2 all independent,
//       3 two-step form (chain depth 1 per 2 steps)
// SMALL: consumer alternates 16x16x4 and 4x4x4 (the 20-column shape).  PRIO: producer s_setprio 3.  NOCONS: consumer idle.
template <int MODE, int SMALL, int PRIO, int NOCONS, int LDSOPS, int GAP = 0>
__global__ __launch_bounds__(512, 1) void k_duo(double *out, unsigned long long *cyc, int iters, const double *__restrict__ src)
{
    __shared__ double tile[8][2048];
    __shared__ volatile int stop;
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8 * 2048; i += 512) (&tile[0][0])[i] = src[i & 4095];
    if (threadIdx.x == 0) stop = 0;
    __syncthreads();
    double r = 0.0;
    unsigned long long t0 = 0, t1 = 0, n = 0;
    if (w < 4) {
        // consumer
        double b[8], bx[8];
        for (int i = 0; i < 8; ++i) { b[i] = src[(lane * 7 + i) & 4095]; bx[i] = src[(lane * 11 + i) & 4095]; }
        double4_t acc[2] = {(double4_t){0, 0, 0, 0}, (double4_t){0, 0, 0, 0}};
        double accx[2] = {0, 0};
        double a = src[lane];
        t0 = __builtin_amdgcn_s_memtime();
        if (!NOCONS)
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[i], acc[i & 1], 0, 0, 0);
                if (SMALL) accx[i & 1] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, bx[i], accx[i & 1], 0, 0, 0);
                if (LDSOPS & 1) a += tile[w][(lane * 2 + i * 128 + it) & 2047] * 1e-30;
                // GAP: the consumer leaves the issue port alone for a moment after every matrix instruction (pair): does the producer
                // of the other wave get an FP64 vector instruction through?  (1..15: s_nop n - 1; 16: s_sleep 1; 17: s_setprio 0 / 3 toggling)
                if (GAP >= 1 && GAP <= 15) asm volatile("s_nop %0" ::"n"(GAP >= 1 && GAP <= 15 ? GAP - 1 : 0));
                if (GAP == 16) __builtin_amdgcn_s_sleep(1);
                if (GAP >= 20 && (i & 1)) asm volatile("s_nop %0" ::"n"(GAP >= 20 ? GAP - 20 : 0));   // after every second pair
            }
            n += 8;
        }
        t1 = __builtin_amdgcn_s_memtime();
        r = acc[0][0] + acc[0][1] + acc[1][2] + acc[1][3] + accx[0] + accx[1];
        if (lane == 0) stop = 1;
    } else {
        // producer: runs until the consumers are done (or iters rounds if there is no consumer)
        if (PRIO) __builtin_amdgcn_s_setprio(3);
        double vc = src[lane] * 1e-3, vp = src[lane + 64] * 1e-3, xx = 0.3 + 1e-3 * lane;
        double vc2 = vc * 0.5, vp2 = vp * 0.5;
        double c[8], d[8];
        for (int i = 0; i < 8; ++i) { c[i] = 1e-3 * src[(lane + i) & 4095]; d[i] = 0.5 + 1e-3 * i; }
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; NOCONS ? it < iters : !stop; ++it) {
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const double t = fma(c[i], xx, d[i]);
                    const double vn = fma(t, vc, -vp);
                    vp = vc; vc = vn;
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (MODE == 1) {
                double t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = fma(c[i], xx, d[i]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) { const double vn = fma(t[i], vc, -vp); vp = vc; vc = vn; }
                __builtin_amdgcn_sched_barrier(0);
            } else if (MODE == 2) {
                double t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = fma(c[i], xx, d[i]);
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = fma(t[i], 1e-3, c[i] * 0.5);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                // two chains (even / odd) that advance two steps per dependent FMA: v[k+2] = p v[k] - a' v[k-1]
                double t[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) t[i] = fma(c[i], xx, d[i]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const double p = fma(t[i + 1], t[i], -1.0);
                    const double wq = t[i + 1] * vp;
                    const double v1 = fma(t[i], vc, -vp);
                    const double v2 = fma(p, vc, -wq);
                    vp = v1; vc = v2;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (LDSOPS & 2) {
                *reinterpret_cast<double2 *>(&tile[w][(lane * 32 + (it & 15) * 2) & 2047]) = make_double2(vc, vp);
                *reinterpret_cast<double2 *>(&tile[w][(lane * 32 + ((it + 5) & 15) * 2) & 2047]) = make_double2(vp, vc);
            }
            n += 16;  // FMAs of the plain recursion this round stands for (8 steps x 2)
            if ((it & 15) == 15) { vc = vc * 1e-3 + 1e-4; vp = vp * 1e-3 + 2e-4; }
        }
        t1 = __builtin_amdgcn_s_memtime();
        r = vc + vp + vc2 + vp2 + c[0] + c[3];
    }
    out[blockIdx.x * 512 + threadIdx.x] = r;
    if (lane == 0) { cyc[(blockIdx.x * 8 + w) * 2] = t1 - t0; cyc[(blockIdx.x * 8 + w) * 2 + 1] = n; }
}

template <int MODE, int SMALL, int PRIO, int NOCONS, int LDSOPS, int GAP = 0>
int run(const char *name, int iters)
{
    const int nb = 256;
    double *out, *src;
    unsigned long long *cyc;
    CK(hipMalloc(&out, nb * 512 * 8)); CK(hipMalloc(&src, 4096 * 8)); CK(hipMalloc(&cyc, nb * 16 * 8));
    std::vector<double> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = (double)rand() / RAND_MAX;
    CK(hipMemcpy(src, h.data(), 4096 * 8, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k_duo<MODE, SMALL, PRIO, NOCONS, LDSOPS, GAP>), dim3(nb), dim3(512), 0, 0, out, cyc, iters, src);
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
    printf("%-58s consumer %7.1f cycles per 16x16x4%s | producer %6.1f cycles per FMA of the plain recursion\n", name,
           cons.empty() ? 0.0 : cons[cons.size() / 2], SMALL ? " (+4x4x4)" : "", prod.empty() ? 0.0 : prod[prod.size() / 2]);
    hipFree(out); hipFree(src); hipFree(cyc);
    return 0;
}

int main()
{
    const int it = 20000;
    run<0, 0, 0, 1, 0>("producer alone, dependent pairs", it);
    run<1, 0, 0, 1, 0>("producer alone, bursts (8 t, 8 chain)", it);
    run<2, 0, 0, 1, 0>("producer alone, independent", it);
    run<3, 0, 0, 1, 0>("producer alone, two-step", it);
    run<0, 0, 0, 0, 0>("16x16x4 stream | dependent pairs", it);
    run<0, 0, 1, 0, 0>("16x16x4 stream | dependent pairs, prio", it);
    run<1, 0, 0, 0, 0>("16x16x4 stream | bursts", it);
    run<1, 0, 1, 0, 0>("16x16x4 stream | bursts, prio", it);
    run<2, 0, 0, 0, 0>("16x16x4 stream | independent", it);
    run<2, 0, 1, 0, 0>("16x16x4 stream | independent, prio", it);
    run<3, 0, 0, 0, 0>("16x16x4 stream | two-step", it);
    run<3, 0, 1, 0, 0>("16x16x4 stream | two-step, prio", it);
    run<0, 1, 0, 0, 0>("16x16x4 + 4x4x4 stream | dependent pairs", it);
    run<1, 1, 0, 0, 0>("16x16x4 + 4x4x4 stream | bursts", it);
    run<1, 1, 1, 0, 0>("16x16x4 + 4x4x4 stream | bursts, prio", it);
    run<3, 1, 0, 0, 0>("16x16x4 + 4x4x4 stream | two-step", it);
    run<3, 1, 1, 0, 0>("16x16x4 + 4x4x4 stream | two-step, prio", it);
    run<1, 1, 0, 0, 3>("16x16x4 + 4x4x4 stream + A reads | bursts + tile stores", it);
    run<1, 1, 1, 0, 3>("16x16x4 + 4x4x4 stream + A reads | bursts + tile stores, prio", it);
    run<3, 1, 1, 0, 3>("16x16x4 + 4x4x4 stream + A reads | two-step + tile stores, prio", it);
    // round 4, second session: gaps in the consumer's stream
    run<1, 1, 0, 0, 0, 1>("16x16x4 + 4x4x4 stream, s_nop 0 per pair | bursts", it);
    run<1, 1, 0, 0, 0, 2>("16x16x4 + 4x4x4 stream, s_nop 1 per pair | bursts", it);
    run<1, 1, 0, 0, 0, 4>("16x16x4 + 4x4x4 stream, s_nop 3 per pair | bursts", it);
    run<1, 1, 0, 0, 0, 8>("16x16x4 + 4x4x4 stream, s_nop 7 per pair | bursts", it);
    run<1, 1, 0, 0, 0, 15>("16x16x4 + 4x4x4 stream, s_nop 14 per pair | bursts", it);
    run<1, 1, 0, 0, 0, 16>("16x16x4 + 4x4x4 stream, s_sleep 1 per pair | bursts", it);
    run<1, 1, 1, 0, 0, 4>("16x16x4 + 4x4x4 stream, s_nop 3 per pair | bursts, prio", it);
    run<1, 1, 1, 0, 0, 8>("16x16x4 + 4x4x4 stream, s_nop 7 per pair | bursts, prio", it);
    run<0, 1, 1, 0, 0, 8>("16x16x4 + 4x4x4 stream, s_nop 7 per pair | dependent pairs, prio", it);
    run<1, 1, 1, 0, 0, 27>("16x16x4 + 4x4x4 stream, s_nop 7 per 2 pairs | bursts, prio", it);
    run<1, 0, 1, 0, 0, 8>("16x16x4 stream, s_nop 7 per instruction | bursts, prio", it);
    run<1, 0, 1, 0, 0, 15>("16x16x4 stream, s_nop 14 per instruction | bursts, prio", it);
    return 0;
}
