// Discover the lane mapping of v_mfma_f64_4x4x4_4b_f64 empirically: A = one-hot at lane la, B = one-hot at lane lb.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double* out){ // grid 64*64 blocks of 64 threads
  int la = blockIdx.x / 64, lb = blockIdx.x % 64, l = threadIdx.x;
  double a = (l==la)?1.0:0.0, b = (l==lb)?1.0:0.0;
  double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a,b,0.0,0,0,0);
  out[(size_t)blockIdx.x*64 + l] = d;
}
int main(){
  double* d; hipMalloc(&d, sizeof(double)*64*64*64);
  hipLaunchKernelGGL(probe, dim3(64*64), dim3(64), 0, 0, d);
  std::vector<double> h(64*64*64); hipMemcpy(h.data(), d, h.size()*8, hipMemcpyDeviceToHost);
  // for each la: list (lb -> output lanes)
  for(int la=0; la<64; ++la){
    printf("la=%2d:", la);
    for(int lb=0; lb<64; ++lb){
      for(int l=0;l<64;++l) if(h[((size_t)la*64+lb)*64+l]!=0.0) printf(" (lb=%d->lo=%d)", lb, l);
    }
    printf("\n");
  }
  return 0;
}
