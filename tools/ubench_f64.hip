// Microbenchmarks that set the fp64 roofline denominator for the kriging predict kernel on gfx950:
//   (1) v_mfma_f64_16x16x4_f64 issue rate, (2) v_fma_f64 rate, (3) both concurrently (separate waves),
//   (4) fp64 exp / sqrt throughput.  Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_f64.hip -o tools/ubench_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)

template<int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters, double a0, double b0){
  d4 acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=(d4){0,0,0,0};
  double a=a0+threadIdx.x*1e-9, b=b0;
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<int NACC>
__global__ void __launch_bounds__(256) k_fma(double* out, int iters, double a0, double b0){
  double acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=i;
  double a=a0+threadIdx.x*1e-9, b=b0;
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_fma(a,acc[i],b);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
// waves with (wave id & 1)==0 do MFMA, others do VALU FMA: tests whether the two fp64 pipes overlap
__global__ void __launch_bounds__(512) k_mixed(double* out, int iters, double a0, double b0, int fma_per_mfma){
  int wave=threadIdx.x>>6;
  double a=a0+threadIdx.x*1e-9, b=b0; double s=0;
  if((wave&1)==0){
    d4 acc[8]; for(int i=0;i<8;i++) acc[i]=(d4){0,0,0,0};
    for(int it=0;it<iters;it++){
#pragma unroll
      for(int i=0;i<8;i++) acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
    }
    for(int i=0;i<8;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  } else {
    double acc[16]; for(int i=0;i<16;i++) acc[i]=i;
    int n=iters*fma_per_mfma/2;
    for(int it=0;it<n;it++){
#pragma unroll
      for(int i=0;i<16;i++) acc[i]=__builtin_fma(a,acc[i],b);
    }
    for(int i=0;i<16;i++) s+=acc[i];
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
// v_mfma_f64_4x4x4_4b_f64: 4 blocks of 4x4x4, one accumulator double per lane (512 flop per instruction)
template<int NACC>
__global__ void __launch_bounds__(256) k_mfma4(double* out, int iters, double a0, double b0){
  double acc[NACC];
  for(int i=0;i<NACC;i++) acc[i]=0.0;
  double a=a0+threadIdx.x*1e-9, b=b0;
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<NACC;i++) acc[i]=__builtin_amdgcn_mfma_f64_4x4x4f64(a,b,acc[i],0,0,0);
  }
  double s=0; for(int i=0;i<NACC;i++) s+=acc[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
// the SAME wave issues 1 MFMA 16x16x4 then R independent v_fma_f64: do the two fp64 pipes overlap inside one wave?
template<int R>
__global__ void __launch_bounds__(256) k_inwave(double* out, int iters, double a0, double b0){
  d4 acc[4]; for(int i=0;i<4;i++) acc[i]=(d4){0,0,0,0};
  double v[16]; for(int i=0;i<16;i++) v[i]=i;
  double a=a0+threadIdx.x*1e-9, b=b0;
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<4;i++){
      acc[i]=__builtin_amdgcn_mfma_f64_16x16x4f64(a,b,acc[i],0,0,0);
#pragma unroll
      for(int r=0;r<R;r++) v[(i*R+r)&15]=__builtin_fma(a,v[(i*R+r)&15],b);
    }
  }
  double s=0; for(int i=0;i<4;i++) s+=acc[i][0]+acc[i][1]+acc[i][2]+acc[i][3];
  for(int i=0;i<16;i++) s+=v[i];
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void __launch_bounds__(256) k_exp(double* out, int iters, double x0){
  double x=x0+threadIdx.x*1e-3; double s=0;
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<8;i++){ s+=exp(-x); x+=1e-7; }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
__global__ void __launch_bounds__(256) k_sqrt(double* out, int iters, double x0){
  double x=x0+threadIdx.x*1e-3; double s=0;
  for(int it=0;it<iters;it++){
#pragma unroll
    for(int i=0;i<8;i++){ s+=sqrt(x); x+=1e-7; }
  }
  out[blockIdx.x*blockDim.x+threadIdx.x]=s;
}
template<class F> float timeit(F f){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms;
}
int main(){
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p,0));
  printf("device %s CUs=%d clock=%d kHz\n",p.name,p.multiProcessorCount,p.clockRate);
  double* out; CK(hipMalloc(&out,sizeof(double)*4096*512));
  int iters=200000;
  // warm-up to get out of the low-power state
  for(int w=0;w<3;w++){hipLaunchKernelGGL(k_fma<16>,dim3(1024),dim3(256),0,0,out,iters,1.0000001,1e-9);} hipDeviceSynchronize();
  for(int bpc: {1,2,4,8}){ // blocks per CU (256 thr = 1 wave/SIMD each)
    int grid=256*bpc;
    float ms=timeit([&]{hipLaunchKernelGGL(k_mfma<8>,dim3(grid),dim3(256),0,0,out,iters,1.0,1.0);});
    double fl=(double)grid*4*iters*8*2048.0;
    printf("mfma_f64_16x16x4 nacc=8 waves/SIMD=%d: %.3f ms  %.2f TFLOP/s  cyc/mfma/SIMD@2.4GHz=%.1f\n",bpc,ms,fl/ms*1e-9, ms*1e-3*2.4e9/(iters*8.0*bpc));
    ms=timeit([&]{hipLaunchKernelGGL(k_mfma<2>,dim3(grid),dim3(256),0,0,out,iters,1.0,1.0);});
    fl=(double)grid*4*iters*2*2048.0;
    printf("mfma_f64_16x16x4 nacc=2 waves/SIMD=%d: %.3f ms  %.2f TFLOP/s\n",bpc,ms,fl/ms*1e-9);
    ms=timeit([&]{hipLaunchKernelGGL(k_mfma<1>,dim3(grid),dim3(256),0,0,out,iters,1.0,1.0);});
    fl=(double)grid*4*iters*1*2048.0;
    printf("mfma_f64_16x16x4 nacc=1 (dependent) waves/SIMD=%d: %.3f ms  %.2f TFLOP/s cyc=%.1f\n",bpc,ms,fl/ms*1e-9, ms*1e-3*2.4e9/(iters*1.0*bpc));
  }
  for(int bpc: {1,2,4,8}){
    int grid=256*bpc;
    float ms=timeit([&]{hipLaunchKernelGGL(k_fma<16>,dim3(grid),dim3(256),0,0,out,iters,1.0000001,1e-9);});
    double fl=(double)grid*256*(double)iters*16*2.0;
    printf("v_fma_f64 nacc=16 waves/SIMD=%d: %.3f ms  %.2f TFLOP/s\n",bpc,ms,fl/ms*1e-9);
  }
  for(int r: {0,4,8,16}){
    int grid=256;
    float ms=timeit([&]{hipLaunchKernelGGL(k_mixed,dim3(grid),dim3(512),0,0,out,iters,1.0,1.0,r);});
    double flm=(double)grid*4*iters*8*2048.0; double flv=(double)grid*4*64*(double)(iters*r/2)*16*2.0;
    printf("mixed (1 mfma wave + 1 fma wave per SIMD) fma_per_mfma=%d: %.3f ms  mfma %.2f TF + valu %.2f TF = %.2f TF\n",r,ms,flm/ms*1e-9,flv/ms*1e-9,(flm+flv)/ms*1e-9);
  }
  for(int bpc: {1,2,4,8}){
    int grid=256*bpc;
    float ms=timeit([&]{hipLaunchKernelGGL(k_mfma4<8>,dim3(grid),dim3(256),0,0,out,iters,1.0,1.0);});
    double fl=(double)grid*4*iters*8*512.0;
    printf("mfma_f64_4x4x4_4b nacc=8 waves/SIMD=%d: %.3f ms  %.2f TFLOP/s  cyc/mfma/SIMD@2.4GHz=%.1f\n",bpc,ms,fl/ms*1e-9, ms*1e-3*2.4e9/(iters*8.0*bpc));
    if(bpc<=2){
      ms=timeit([&]{hipLaunchKernelGGL(k_mfma4<1>,dim3(grid),dim3(256),0,0,out,iters,1.0,1.0);});
      printf("mfma_f64_4x4x4_4b nacc=1 (dependent) waves/SIMD=%d: %.2f TFLOP/s cyc/mfma=%.1f\n",bpc,(double)grid*4*iters*1*512.0/ms*1e-9, ms*1e-3*2.4e9/(iters*1.0*bpc));
      ms=timeit([&]{hipLaunchKernelGGL(k_mfma4<2>,dim3(grid),dim3(256),0,0,out,iters,1.0,1.0);});
      printf("mfma_f64_4x4x4_4b nacc=2 waves/SIMD=%d: %.2f TFLOP/s cyc/mfma=%.1f\n",bpc,(double)grid*4*iters*2*512.0/ms*1e-9, ms*1e-3*2.4e9/(iters*2.0*bpc));
      ms=timeit([&]{hipLaunchKernelGGL(k_mfma4<4>,dim3(grid),dim3(256),0,0,out,iters,1.0,1.0);});
      printf("mfma_f64_4x4x4_4b nacc=4 waves/SIMD=%d: %.2f TFLOP/s cyc/mfma=%.1f\n",bpc,(double)grid*4*iters*4*512.0/ms*1e-9, ms*1e-3*2.4e9/(iters*4.0*bpc));
    }
  }
  for(int bpc: {1,2}){
    int grid=256*bpc;
    float ms=timeit([&]{hipLaunchKernelGGL(k_inwave<4>,dim3(grid),dim3(256),0,0,out,iters,1.0,1.0);});
    double flm=(double)grid*4*iters*4*2048.0, flv=(double)grid*256*(double)iters*16*2.0;
    printf("in-wave 1 mfma + 4 fma  waves/SIMD=%d: %.3f ms  mfma %.2f + valu %.2f = %.2f TF\n",bpc,ms,flm/ms*1e-9,flv/ms*1e-9,(flm+flv)/ms*1e-9);
    ms=timeit([&]{hipLaunchKernelGGL(k_inwave<8>,dim3(grid),dim3(256),0,0,out,iters,1.0,1.0);});
    flv*=2;
    printf("in-wave 1 mfma + 8 fma  waves/SIMD=%d: %.3f ms  mfma %.2f + valu %.2f = %.2f TF\n",bpc,ms,flm/ms*1e-9,flv/ms*1e-9,(flm+flv)/ms*1e-9);
    ms=timeit([&]{hipLaunchKernelGGL(k_inwave<16>,dim3(grid),dim3(256),0,0,out,iters,1.0,1.0);});
    flv*=2;
    printf("in-wave 1 mfma + 16 fma waves/SIMD=%d: %.3f ms  mfma %.2f + valu %.2f = %.2f TF\n",bpc,ms,flm/ms*1e-9,flv/ms*1e-9,(flm+flv)/ms*1e-9);
  }
  for(int bpc: {1,4}){
    int grid=256*bpc;
    float ms=timeit([&]{hipLaunchKernelGGL(k_exp,dim3(grid),dim3(256),0,0,out,20000,0.5);});
    double n=(double)grid*256*20000*8;
    printf("exp(f64) waves/SIMD=%d: %.3f ms  %.2f Gexp/s  (%.1f cyc/wave-exp/SIMD)\n",bpc,ms,n/ms*1e-6, ms*1e-3*2.4e9/(20000.0*8*bpc));
    ms=timeit([&]{hipLaunchKernelGGL(k_sqrt,dim3(grid),dim3(256),0,0,out,20000,0.5);});
    printf("sqrt(f64) waves/SIMD=%d: %.3f ms  %.2f Gsqrt/s (%.1f cyc/wave-sqrt/SIMD)\n",bpc,ms,n/ms*1e-6, ms*1e-3*2.4e9/(20000.0*8*bpc));
  }
  return 0;
}
