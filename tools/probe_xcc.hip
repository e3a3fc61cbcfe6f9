// Which XCD does block b run on?  Compares HW_REG_XCC_ID with blockIdx % 8.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(unsigned* out){ unsigned x; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x)); if(threadIdx.x==0) out[blockIdx.x]=x; }
int main(){ const int n=2048; unsigned* d; hipMalloc(&d,4*n); hipLaunchKernelGGL(k,dim3(n),dim3(512),0,0,d); std::vector<unsigned> h(n); hipMemcpy(h.data(),d,4*n,hipMemcpyDeviceToHost);
  int match=0; int hist[16]={0}; for(int b=0;b<n;++b){ if((h[b]&7)==(unsigned)(b%8)) ++match; hist[h[b]&15]++; }
  printf("raw first 16:"); for(int b=0;b<16;++b) printf(" %08x",h[b]); printf("\nmatch (xcc&7)==b%%8: %d / %d\nhist of xcc&15:",match,n); for(int i=0;i<16;++i) printf(" %d",hist[i]); printf("\n"); return 0; }
