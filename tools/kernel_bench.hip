// Standalone timing of the library's kernels on BASELINE config-2 shapes (M=5001 -> Mp=5120) with
// HIP events.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pykrige_amd/csrc tools/kernel_bench.hip -o tools/kernel_bench
#include "mik_kernels.h"
#include "mik_k_experiments.h"  // kernels the library no longer instantiates (round 6)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace mik;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
template<class F> float timeit(F f, int reps){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for(int i=0;i<reps;i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms/reps;
}
int main(int argc,char**argv){
  const int Mp = argc>1? atoi(argv[1]) : 5120, P = argc>2? atoi(argv[2]) : 32768;
  const int nblk = Mp/128, kend = argc>3? atoi(argv[3]) : Mp;
  double *T,*Bt,*part,*Dinv,*DinvT,*Cold,*Cnew,*Rt; int* flag; unsigned long long* queue; CK(hipMalloc(&queue,64));
  const unsigned pgrid = 512;
  CK(hipMalloc(&T,sizeof(double)*(size_t)Mp*Mp)); CK(hipMalloc(&Bt,sizeof(double)*(size_t)P*Mp));
  CK(hipMalloc(&part,sizeof(double)*(size_t)P*nblk)); CK(hipMalloc(&Dinv,131072)); CK(hipMalloc(&DinvT,131072));
  CK(hipMalloc(&Cold,sizeof(double)*(size_t)Mp*128)); CK(hipMalloc(&Cnew,sizeof(double)*(size_t)Mp*128)); CK(hipMalloc(&Rt,sizeof(double)*(size_t)Mp*128));
  CK(hipMalloc(&flag,sizeof(int)*MIK_F_INTS)); CK(hipMemset(flag,0,sizeof(int)*MIK_F_INTS));
  { std::vector<double> h((size_t)Mp*Mp); srand(1);
    for(size_t i=0;i<h.size();++i) h[i]=(rand()/(double)RAND_MAX-0.5)*0.01;
    for(int i=0;i<Mp;++i) h[(size_t)i*Mp+i]=1.0+0.1*(i%7);
    for(int i=0;i<Mp;++i) for(int j=0;j<i;++j) h[(size_t)i*Mp+j]=h[(size_t)j*Mp+i];
    CK(hipMemcpy(T,h.data(),h.size()*8,hipMemcpyHostToDevice));
    std::vector<double> b((size_t)P*Mp); for(size_t i=0;i<b.size();++i) b[i]=rand()/(double)RAND_MAX-0.5;
    CK(hipMemcpy(Bt,b.data(),b.size()*8,hipMemcpyHostToDevice));
    CK(hipMemcpy(Cold,b.data(),(size_t)Mp*128*8,hipMemcpyHostToDevice)); CK(hipMemcpy(Cnew,b.data()+Mp*128,(size_t)Mp*128*8,hipMemcpyHostToDevice));
    CK(hipMemcpy(Rt,b.data()+2*Mp*128,(size_t)Mp*128*8,hipMemcpyHostToDevice)); }
  const long tiles=(long)nblk*(P/128); const unsigned vgrid=(unsigned)(8*((tiles+7)/8));
  const unsigned grid=(unsigned)super_grid(nblk,P/128);
  double kext_sym=0; for(int ib=0;ib<nblk;++ib) kext_sym+= kend-ib*128;
  if(argc>3){ kext_sym=0; for(int ib=0;ib<nblk;++ib) kext_sym+= (kend-ib*128>0? kend-ib*128:0); }
  const double fl_full=2.0*128*128*(double)kend*nblk*(P/128), fl_sym=2.0*128*128*kext_sym*(P/128);
  // warm the clocks
  for(int w=0;w<2;++w) hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<true,4>),dim3(pgrid),dim3(256),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);
  hipDeviceSynchronize();
  float ms;
  if(argc>4 && !strcmp(argv[4],"tri")){  // round 3: triangular diagonal blocks (gemm_core TRI) against the whole-block form, alternating
    std::vector<double> a((size_t)P*nblk), b((size_t)P*nblk);
    double kext_tri=0; for(int ib=0;ib<nblk;++ib){ int ext=kend-ib*128; if(ext<0) ext=0; int nt=(ext<128?ext:128)/16; kext_tri+=(ext-16*nt)+16.0*(nt*(nt+1)/2)/8.0; }
    const double fl_tri=2.0*128*128*kext_tri*(P/128);
    for(int rep=0;rep<3;++rep){
      ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<true,2>),dim3(pgrid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},3);
      printf("A/B sym 8-wave, whole diagonal blocks     : %.3f ms  executed %.2f TF/s  useful-equivalent %.2f TF/s\n",ms,fl_sym/ms*1e-9,fl_full/2/ms*1e-9);
      if(rep==0) CK(hipMemcpy(a.data(),part,a.size()*8,hipMemcpyDeviceToHost));
      ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<true,2,true,false,true>),dim3(pgrid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},3);
      printf("A/B sym 8-wave, triangular diagonal blocks: %.3f ms  executed %.2f TF/s  useful-equivalent %.2f TF/s\n",ms,fl_tri/ms*1e-9,fl_full/2/ms*1e-9);
      if(rep==0) CK(hipMemcpy(b.data(),part,b.size()*8,hipMemcpyDeviceToHost));
    }
    double md=0,mx=0; for(size_t i=0;i<a.size();++i){ md=fmax(md,fabs(a[i]-b[i])); mx=fmax(mx,fabs(a[i])); }
    printf("   partials, triangular vs whole diagonal blocks: max|diff| %.3e (max|partial| %.3e)\n",md,mx);
    return 0;
  }
  ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<true,4>),dim3(pgrid),dim3(256),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},3);
  printf("k_contract<sym>  mfma : %.3f ms  executed %.2f TF/s  effective(2M^2) %.2f TF/s\n",ms,fl_sym/ms*1e-9,fl_full/ms*1e-9);
  std::vector<double> refsym((size_t)P*nblk); CK(hipMemcpy(refsym.data(),part,refsym.size()*8,hipMemcpyDeviceToHost));  // symmetric-form partials (4-wave)
  ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<true,2,true,true>),dim3(pgrid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},3);
  printf("A/B sym 8-wave persistent, PAIR units : %.3f ms  executed %.2f TF/s\n",ms,fl_sym/ms*1e-9);
  for(int rep=0;rep<2;++rep){
    ms=timeit([&]{hipLaunchKernelGGL((k_contract<true,2,false>),dim3(grid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},3);
    printf("A/B sym 8-wave one block per tile : %.3f ms  executed %.2f TF/s\n",ms,fl_sym/ms*1e-9);
    ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<true,2,true>),dim3(pgrid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},3);
    printf("A/B sym 8-wave persistent (512 blks): %.3f ms  executed %.2f TF/s\n",ms,fl_sym/ms*1e-9);
    ms=timeit([&]{hipLaunchKernelGGL((k_contract<false,2,false>),dim3(grid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},2);
    printf("A/B full 8-wave one block per tile: %.3f ms  executed %.2f TF/s\n",ms,fl_full/ms*1e-9);
    ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<false,2,true>),dim3(pgrid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},2);
    printf("A/B full 8-wave persistent         : %.3f ms  executed %.2f TF/s\n",ms,fl_full/ms*1e-9);
  }
  std::vector<double> ref((size_t)P*nblk), got((size_t)P*nblk);
  CK(hipMemcpy(ref.data(),part,ref.size()*8,hipMemcpyDeviceToHost));
  ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<false,4>),dim3(pgrid),dim3(256),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},2);
  printf("k_contract<full> mfma : %.3f ms  executed %.2f TF/s\n",ms,fl_full/ms*1e-9);
  ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<true,2>),dim3(pgrid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},3);
  printf("k_contract<sym>  mfma 8 waves/block (wave tile 32x64): %.3f ms  executed %.2f TF/s  effective %.2f TF/s\n",ms,fl_sym/ms*1e-9,fl_full/ms*1e-9);
  CK(hipMemcpy(got.data(),part,got.size()*8,hipMemcpyDeviceToHost));
  { double md=0; for(size_t i=0;i<refsym.size();++i) md=fmax(md,fabs(refsym[i]-got[i])); printf("   symmetric form, 8-wave vs 4-wave partials: max|diff| %.3e\n",md); }
  { // the symmetric and the full form split b^T A_inv b differently over the row blocks; their sums over the row blocks agree
    double md=0, mx=0; for(int t=0;t<P;++t){ double a=0,b=0; for(int ib=0;ib<nblk;++ib){ a+=ref[(size_t)ib*P+t]; b+=got[(size_t)ib*P+t]; } md=fmax(md,fabs(a-b)); mx=fmax(mx,fabs(a)); }
    printf("   symmetric vs full form, sum over row blocks: max|diff| %.3e (max|sum| %.3e)\n",md,mx); }
  ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<false,2>),dim3(pgrid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},2);
  printf("k_contract<full> mfma 8 waves/block: %.3f ms  executed %.2f TF/s\n",ms,fl_full/ms*1e-9);
  // round 3: the tile-shape experiment -- 256 x 128 block tile (16 waves, one 1024-thread block per CU, 96 KB LDS), same MFMA loop
  if (Mp % 256 == 0) {
    const int nb256 = Mp / 256;
    const size_t lds256 = sizeof(GemmSmemT<256>);
    CK(hipFuncSetAttribute((const void*)k_contract256<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256));
    CK(hipFuncSetAttribute((const void*)k_contract256<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds256));
    double kext256 = 0; for (int ib = 0; ib < nb256; ++ib) kext256 += (kend - ib * 256 > 0 ? kend - ib * 256 : 0);
    const double fl_sym256 = 2.0 * 256 * 128 * kext256 * (P / 128);
    for (int rep = 0; rep < 2; ++rep) {
      ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<true,2>),dim3(pgrid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},3);
      printf("TILE A/B sym 128x128 (8 waves, 2 blocks/CU, 512 blocks): %.3f ms  executed %.2f TF/s  useful-rate %.2f TF/s\n",ms,fl_sym/ms*1e-9,fl_sym/ms*1e-9*(double)Mp*Mp/2/(128.0*kext_sym));
      std::vector<double> p128((size_t)P*nblk); CK(hipMemcpy(p128.data(),part,p128.size()*8,hipMemcpyDeviceToHost));
      ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract256<true>),dim3(256),dim3(1024),lds256,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nb256,kend,queue);},3);
      printf("TILE A/B sym 256x128 (16 waves, 1 block/CU, 256 blocks): %.3f ms  executed %.2f TF/s  useful-rate %.2f TF/s\n",ms,fl_sym256/ms*1e-9,fl_sym256/ms*1e-9*(double)Mp*Mp/2/(256.0*kext256));
      std::vector<double> p256((size_t)P*nb256); CK(hipMemcpy(p256.data(),part,p256.size()*8,hipMemcpyDeviceToHost));
      if (rep == 0) { double md=0, mx=0; for(int t=0;t<P;++t){ double a=0,b=0; for(int ib=0;ib<nblk;++ib) a+=p128[(size_t)ib*P+t]; for(int ib=0;ib<nb256;++ib) b+=p256[(size_t)ib*P+t]; md=fmax(md,fabs(a-b)); mx=fmax(mx,fabs(a)); }
        printf("   256x128 vs 128x128, sum over row blocks: max|diff| %.3e (max|sum| %.3e)\n",md,mx); }
      ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract256<false>),dim3(256),dim3(1024),lds256,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nb256,kend,queue);},2);
      printf("TILE A/B full 256x128 (16 waves, 1 block/CU)            : %.3f ms  executed %.2f TF/s\n",ms,fl_full/ms*1e-9);
    }
  }
  ms=timeit([&]{hipMemsetAsync(queue,0,64,0); hipLaunchKernelGGL((k_contract<false,4>),dim3(pgrid),dim3(256),40960,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend,queue);},2);
  printf("k_contract<full> mfma, ONE block per CU (40 KB dummy dynamic LDS): %.3f ms  executed %.2f TF/s\n",ms,fl_full/ms*1e-9);
  ms=timeit([&]{hipLaunchKernelGGL(k_contract_valu<true>,dim3(vgrid),dim3(256),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend);},3);
  printf("k_contract<sym>  valu : %.3f ms  executed %.2f TF/s  effective(2M^2) %.2f TF/s\n",ms,fl_sym/ms*1e-9,fl_full/ms*1e-9);
  CK(hipMemcpy(got.data(),part,got.size()*8,hipMemcpyDeviceToHost));
  { double md=0, mx=0; for(size_t i=0;i<refsym.size();++i){ md=fmax(md,fabs(refsym[i]-got[i])); mx=fmax(mx,fabs(refsym[i])); } printf("   symmetric form, valu vs mfma partials: max|diff| %.3e (max|ref| %.3e)\n",md,mx); }
  ms=timeit([&]{hipLaunchKernelGGL(k_contract_valu<false>,dim3(vgrid),dim3(256),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend);},2);
  printf("k_contract<full> valu : %.3f ms  executed %.2f TF/s\n",ms,fl_full/ms*1e-9);
  // ablations of the main loop (full form): what each component costs
#define ABLATE(NAI, ABL, NT, label) ms=timeit([&]{hipLaunchKernelGGL((k_contract_ablate<NAI,ABL>),dim3(grid),dim3(NT),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend);},2); \
  printf("ablate %-44s: %.3f ms  %.2f TF/s\n",label,ms,fl_full/ms*1e-9);
  ABLATE(4,0,256,"4-wave, nothing removed");
  ABLATE(4,1,256,"4-wave, no LDS-DMA");
  ABLATE(4,2,256,"4-wave, no fragment ds_reads");
  ABLATE(4,4,256,"4-wave, no barrier");
  ABLATE(4,3,256,"4-wave, no DMA + no ds_reads");
  ABLATE(4,7,256,"4-wave, MFMA only");
  { ms=timeit([&]{hipLaunchKernelGGL((k_contract_ablate<2,16>),dim3(grid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend);},3);
    printf("ablate %-44s: %.3f ms  %.2f TF/s (executed, ragged flop count)\n","8-wave, ragged K (k >= i0), no doubling, no epilogue",ms,fl_sym/ms*1e-9); }
  { ms=timeit([&]{hipLaunchKernelGGL((k_contract_ablate<2,17>),dim3(grid),dim3(512),0,0,(const double*)T,(long)Mp,(const double*)Bt,(long)Mp,part,P,nblk,kend);},3);
    printf("ablate %-44s: %.3f ms  %.2f TF/s (executed, ragged flop count)\n","8-wave, ragged K, no LDS-DMA",ms,fl_sym/ms*1e-9); }
  ABLATE(4,8,256,"4-wave, DMA source always k-tile 0 (cached)");
  ABLATE(2,8,512,"8-wave, DMA source always k-tile 0 (cached)");
  ABLATE(2,0,512,"8-wave, nothing removed");
  ABLATE(2,1,512,"8-wave, no LDS-DMA");
  ABLATE(2,2,512,"8-wave, no fragment ds_reads");
  ABLATE(2,4,512,"8-wave, no barrier");
  ABLATE(2,7,512,"8-wave, MFMA only");
  ABLATE(2,32,512,"8-wave, B tile generated on the VALU (exp)");
  ABLATE(4,32,256,"4-wave, B tile generated on the VALU (exp)");
  // inverse pieces
  ms=timeit([&]{hipLaunchKernelGGL(k_diag_inv,dim3(1),dim3(1024),0,0,(const double*)T,(long)Mp,0,0,Dinv,DinvT,flag);},10);
  printf("k_diag_inv (1024 thr) back-to-back: %.1f us\n",ms*1e3);
  ms=timeit([&]{hipLaunchKernelGGL((k_diag_inv_t<16,16>),dim3(1),dim3(256),0,0,(const double*)T,(long)Mp,0,0,Dinv,DinvT,flag);},10);
  printf("k_diag_inv_t<16,16> (scalar pivots) back-to-back: %.1f us\n",ms*1e3);

  const long ut=(long)nblk*nblk; const unsigned ug=(unsigned)(8*((ut+7)/8));
  ms=timeit([&]{hipLaunchKernelGGL((k_update<false,4>),dim3(ug),dim3(256),0,0,T,(long)Mp,nblk,1,(const double*)Cold,(const double*)Cnew,(const double*)Rt,(const double*)Dinv,0,0,(double*)nullptr);},5);
  printf("k_update: %.1f us  %.2f TF/s\n",ms*1e3, 2.0*Mp*(double)Mp*128/ms*1e-9);
  ms=timeit([&]{hipLaunchKernelGGL((k_update<false,2>),dim3(ug),dim3(512),0,0,T,(long)Mp,nblk,1,(const double*)Cold,(const double*)Cnew,(const double*)Rt,(const double*)Dinv,0,0,(double*)nullptr,(double*)nullptr);},5);
  printf("k_update, 8 waves per tile (round 3): %.1f us  %.2f TF/s\n",ms*1e3, 2.0*Mp*(double)Mp*128/ms*1e-9);
  ms=timeit([&]{hipLaunchKernelGGL(k_diag_inv,dim3(1),dim3(1024),0,0,(const double*)T,(long)Mp,0,0,Dinv,DinvT,flag);
               hipLaunchKernelGGL((k_update<false,4>),dim3(ug),dim3(256),0,0,T,(long)Mp,nblk,1,(const double*)Cold,(const double*)Cnew,(const double*)Rt,(const double*)Dinv,0,0,(double*)nullptr);},5);
  printf("k_diag_inv + k_update interleaved: %.1f us per pair\n",ms*1e3);
  ms=timeit([&]{hipLaunchKernelGGL(k_panel<4>,dim3(nblk),dim3(256),0,0,(const double*)Cold,128L,(const double*)DinvT,-1.0,Cnew);},5);
  printf("k_panel: %.1f us\n",ms*1e3);
  return 0;
}
