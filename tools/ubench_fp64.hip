// Micro-benchmark: FP64 VALU FMA rate, FP64 MFMA 16x16x4 rate, and co-execution, plus HBM copy.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_fp64.hip -o gpurun_out/ubench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));

#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template<int MODE> // 0: VALU only, 1: MFMA only, 2: both interleaved in the same wave
__global__ __launch_bounds__(256) void k_flops(double* out, int iters, double seed)
{
  double a0=seed+threadIdx.x*1e-9, a1=a0+1e-3, a2=a0+2e-3, a3=a0+3e-3, a4=a0+4e-3,a5=a0+5e-3,a6=a0+6e-3,a7=a0+7e-3;
  double x = 0.999999 + seed*1e-12, y = 1e-7;
  double4_t c0 = {0,0,0,0}, c1={0,0,0,0}, c2={0,0,0,0}, c3={0,0,0,0};
  double ma = seed*1e-3+threadIdx.x*1e-6, mb = 1.0+threadIdx.x*1e-7;
  for (int i=0;i<iters;++i) {
    if (MODE==0 || MODE==2) {
      #pragma unroll
      for (int u=0;u<2;++u){
      a0 = __builtin_fma(a0,x,y); a1 = __builtin_fma(a1,x,y); a2 = __builtin_fma(a2,x,y); a3 = __builtin_fma(a3,x,y);
      a4 = __builtin_fma(a4,x,y); a5 = __builtin_fma(a5,x,y); a6 = __builtin_fma(a6,x,y); a7 = __builtin_fma(a7,x,y);}
    }
    if (MODE==1 || MODE==2) {
      c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c0, 0,0,0);
      c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c1, 0,0,0);
      if (MODE==1){
      c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c2, 0,0,0);
      c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(ma, mb, c3, 0,0,0);}
    }
  }
  double r = a0+a1+a2+a3+a4+a5+a6+a7 + c0[0]+c0[1]+c0[2]+c0[3]+c1[0]+c1[1]+c1[2]+c1[3]+c2[0]+c2[3]+c3[1]+c3[2];
  out[blockIdx.x*blockDim.x+threadIdx.x] = r;
}

__global__ void k_copy(const double2* __restrict__ in, double2* __restrict__ out, size_t n)
{
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, st=(size_t)gridDim.x*blockDim.x;
  for (; i<n; i+=st) out[i]=in[i];
}
__global__ void k_read(const double2* __restrict__ in, double* out, size_t n)
{
  size_t i = blockIdx.x*(size_t)blockDim.x+threadIdx.x, st=(size_t)gridDim.x*blockDim.x;
  double s=0; for (; i<n; i+=st) { double2 v=in[i]; s+=v.x+v.y; }
  if (s==1.2345e-300) out[0]=s;
}

// check the f64 MFMA layout: A[i][k] (lane = i + 16k), B[k][j] (lane = j + 16k), D[i][j]: lane=j+16*(i%4)?...
__global__ void k_layout(double* out)
{
  int lane = threadIdx.x;
  int i = lane & 15, k = lane >> 4;
  double a = (double)(i*10 + k);        // A[i][k] = 10 i + k
  double b = (double)(1000*k + (lane&15)); // B[k][j] = 1000k + j
  double4_t c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0,0,0);
  for (int r=0;r<4;++r) out[lane*4+r] = c[r];
}

int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  double* d; CK(hipMalloc(&d, 1<<26));
  hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000;
  for (int wpb : {4, 8}) for (int bpc : {1, 2, 4}) {
    int blocks = p.multiProcessorCount*bpc, thr = wpb*64;
    for (int mode=0; mode<3; ++mode) {
      float best=1e30f;
      for (int rep=0; rep<3; ++rep){
        CK(hipEventRecord(e0));
        if (mode==0) k_flops<0><<<blocks,thr>>>(d,iters,1.0);
        if (mode==1) k_flops<1><<<blocks,thr>>>(d,iters,1.0);
        if (mode==2) k_flops<2><<<blocks,thr>>>(d,iters,1.0);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms,e0,e1)); if(ms<best)best=ms;
      }
      double valu = (mode!=1)? 16.0*2*64 : 0;                 // flops per wave per iter
      double mfma = (mode==1)? 4.0*2048 : (mode==2? 2.0*2048:0);
      double waves = (double)blocks*wpb;
      printf("waves/blk %d blk/CU %d mode %d: %.3f ms  VALU %.1f TF  MFMA %.1f TF  total %.1f TF\n", wpb,bpc,mode,best,
        valu*waves*iters/best*1e-9, mfma*waves*iters/best*1e-9, (valu+mfma)*waves*iters/best*1e-9);
    }
  }
  // layout probe
  k_layout<<<1,64>>>(d); std::vector<double> h(256); CK(hipMemcpy(h.data(), d, 256*8, hipMemcpyDeviceToHost));
  // expected D[i][j] = sum_k (10i+k)(1000k+j)
  int okA=0, okB=0;
  for (int lane=0; lane<64; ++lane) for (int r=0;r<4;++r){
    int j = lane&15; int iA = (lane>>4) + 4*r; int iB = 4*(lane>>4) + r;
    double eA=0,eB=0; for(int k=0;k<4;++k){ eA += (10.0*iA+k)*(1000.0*k+j); eB += (10.0*iB+k)*(1000.0*k+j);} 
    okA += (h[lane*4+r]==eA); okB += (h[lane*4+r]==eB);
  }
  printf("layout: rowmap (lane>>4)+4*reg matches %d/256 ; 4*(lane>>4)+reg matches %d/256\n", okA, okB);
  // HBM
  size_t n = (size_t)1<<27; // 2 GiB of double2
  double2 *a,*b; CK(hipMalloc(&a,n*16)); CK(hipMalloc(&b,n*16)); CK(hipMemset(a,1,n*16));
  for (int rep=0;rep<3;++rep){
    CK(hipEventRecord(e0)); k_copy<<<2048*4,256>>>(a,b,n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms,e0,e1)); printf("copy 2GiB+2GiB: %.3f ms  %.2f TB/s\n", ms, 2.0*n*16/ms*1e-9);
    CK(hipEventRecord(e0)); k_read<<<2048*4,256>>>(a,d,n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms,e0,e1)); printf("read 2GiB: %.3f ms  %.2f TB/s\n", ms, 1.0*n*16/ms*1e-9);
  }
  return 0;
}
