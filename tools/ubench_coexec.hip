// Does an FP64 VALU dependency chain make progress while another wave on the SAME SIMD streams
// FP64 MFMAs?  And when the MFMA waves sit on other SIMDs of the CU?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

__device__ inline int simd_id() {
  unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(v)); return (v >> 4) & 3;  // SIMD_ID bits [5:4]
}

// role: 0 = VALU chain (2 independent chains, like the recursion), 1 = MFMA stream, 2 = idle
// mode 0: all waves VALU; 1: all MFMA; 2: waves w<half VALU, others MFMA (mixed per SIMD);
// mode 3: role by SIMD id: SIMD 0 -> VALU, SIMDs 1..3 -> MFMA; mode 4: SIMD0 VALU others idle; mode 5: SIMD0 idle others MFMA
__global__ void k(int mode, int iters, double* out, unsigned long long* cyc, int* simds)
{
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int sid = simd_id();
  int role;
  if (mode == 0) role = 0; else if (mode == 1) role = 1;
  else if (mode == 2) role = (w < nw / 2) ? 0 : 1;
  else if (mode == 3) role = sid == 0 ? 0 : 1;
  else if (mode == 4) role = sid == 0 ? 0 : 2;
  else role = sid == 0 ? 2 : 1;
  double a0 = 1.0 + threadIdx.x * 1e-9, a1 = 0.5, x = 0.9999999, y = 1e-9;
  double4_t c0 = {0,0,0,0}, c1 = {0,0,0,0};
  double ma = 1e-3 + threadIdx.x * 1e-6, mb = 1.0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 0) {
#ifdef PRIO
    __builtin_amdgcn_s_setprio(3);  // does the arbiter let the VALU chain in ahead of the MFMA stream?
#endif
#ifdef NCHAIN
    // NCHAIN independent dependency chains per lane: latency- or throughput-starved beside MFMA?
    double ch[NCHAIN];
#pragma unroll
    for (int u = 0; u < NCHAIN; ++u) ch[u] = a0 + u;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < NCHAIN; ++u) ch[u] = __builtin_fma(ch[u], x, y);
    }
#pragma unroll
    for (int u = 0; u < NCHAIN; ++u) a1 += ch[u];
#else
    for (int i = 0; i < iters; ++i) { a0 = __builtin_fma(a0, x, y); a1 = __builtin_fma(a1, x, y); }
#endif
  } else if (role == 1) {
#ifdef QUAD
    // same flops per iteration as two 16x16x4: eight 4x4x4_4b on independent accumulators
    double q[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = __builtin_amdgcn_mfma_f64_4x4x4f64(ma, mb, q[u], 0, 0, 0);
    }
    c0[0] = q[0] + q[1] + q[2] + q[3]; c1[1] = q[4] + q[5] + q[6] + q[7];
#elif defined(NACC)
    // NACC independent 16x16x4 accumulators per wave, 2 MFMAs per accumulator per iteration / NACC
    double4_t cc[NACC];
#pragma unroll
    for (int u = 0; u < NACC; ++u) cc[u] = c0;
    for (int i = 0; i < iters; i += NACC / 2) {
#pragma unroll
      for (int u = 0; u < NACC; ++u) cc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, cc[u], 0,0,0);
    }
#pragma unroll
    for (int u = 0; u < NACC; ++u) c0 += cc[u];
#else
    for (int i = 0; i < iters; ++i) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c0, 0,0,0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c1, 0,0,0);
    }
#endif
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + c0[0] + c1[1];
  if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * nw + w] = t1 - t0; simds[blockIdx.x * nw + w] = sid * 4 + role; }
}

int main() {
  double* d; unsigned long long* c; int* s;
  CK(hipMalloc(&d, 1 << 24)); CK(hipMalloc(&c, 1 << 20)); CK(hipMalloc(&s, 1 << 20));
  const int iters = 20000, blocks = 256;
  for (int nw : {4, 8, 16}) for (int mode = 0; mode < 6; ++mode) {
    k<<<blocks, nw * 64>>>(mode, iters, d, c, s); CK(hipDeviceSynchronize());
    static unsigned long long hc[1 << 14]; static int hs[1 << 14];
    CK(hipMemcpy(hc, c, sizeof(unsigned long long) * blocks * nw, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hs, s, sizeof(int) * blocks * nw, hipMemcpyDeviceToHost));
    double sum[3] = {0,0,0}; int cnt[3] = {0,0,0}; int persimd[4] = {0,0,0,0};
    for (int i = 0; i < blocks * nw; ++i) { int role = hs[i] & 3; sum[role] += hc[i]; cnt[role]++; if (i < nw) persimd[hs[i] >> 2]++; }
    printf("waves/blk %2d mode %d: VALU waves %4d cyc/iter %.1f | MFMA waves %4d cyc/iter %.1f | simd occupancy of blk0: %d %d %d %d\n", nw, mode,
      cnt[0], cnt[0] ? sum[0] / cnt[0] / iters : 0.0, cnt[1], cnt[1] ? sum[1] / cnt[1] / iters : 0.0, persimd[0], persimd[1], persimd[2], persimd[3]);
  }
  return 0;
}
