// layout + rate probe of v_mfma_f64_4x4x4_4b_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
__global__ void k_layout(const double* a, const double* b, double* out) {
  int lane = threadIdx.x;
  double c = __builtin_amdgcn_mfma_f64_4x4x4f64(a[lane], b[lane], 0.0, 0, 0, 0);
  out[lane] = c;
}
__global__ void k_rate(double* out, int iters) {
  double a = 1e-3 + threadIdx.x * 1e-6, b = 1.0 + threadIdx.x * 1e-7;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = c0 + c1 + c2 + c3;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[1 << 20] = (double)(t1 - t0) / (4.0 * iters);
}
int main() {
  double *da, *db, *dout; CK(hipMalloc(&da, 512)); CK(hipMalloc(&db, 512)); CK(hipMalloc(&dout, (1 << 21) * 8));
  // probe A layout: set a[lane] = lane+1, b = indicator of one lane at a time -> see which outputs light up
  std::vector<double> ha(64), hb(64), ho(64);
  printf("probe: for each (la, lb) pair with a[la]=1, b[lb]=1, which output lanes are 1\n");
  for (int la : {0, 1, 4, 5, 16, 21, 63}) for (int lb : {0, 1, 4, 5, 16, 21, 63}) {
    for (int i = 0; i < 64; ++i) { ha[i] = (i == la); hb[i] = (i == lb); }
    CK(hipMemcpy(da, ha.data(), 512, hipMemcpyHostToDevice)); CK(hipMemcpy(db, hb.data(), 512, hipMemcpyHostToDevice));
    k_layout<<<1, 64>>>(da, db, dout); CK(hipMemcpy(ho.data(), dout, 512, hipMemcpyDeviceToHost));
    printf("a@%2d b@%2d ->", la, lb); for (int i = 0; i < 64; ++i) if (ho[i] != 0) printf(" %d", i); printf("\n");
  }
  for (int wpb : {4, 8, 16}) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 20000, blocks = 256;
    CK(hipEventRecord(e0)); k_rate<<<blocks, wpb * 64>>>(dout, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); double cyc; CK(hipMemcpy(&cyc, dout + (1 << 20), 8, hipMemcpyDeviceToHost));
    double flops = 4.0 * iters * 512.0 * blocks * wpb;
    printf("4x4x4: waves/CU %2d: %.3f ms, %.1f TF, %.1f cycles per MFMA per wave\n", wpb, ms, flops / ms * 1e-9, cyc);
  }
  return 0;
}
