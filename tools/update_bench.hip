// Standalone timing of the block sweep's trailing update (K2): k_update (8 waves per tile, two blocks per CU) against
// k_update_deep (round 5) and its ablations, on a half sweep's upper block triangle; one profiled launch of k_update (MIK_UPD_PROF hooks: s_memtime at kernel
// entry, K loop start / end, last store; profiles/r05_update_kernel_phases.txt).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pykrige_amd/csrc tools/update_bench.hip -o tools/update_bench
// Run:   tools/update_bench [Mp = 8064] [tpb ...]
#define MIK_UPD_PROF 1
#include "mik_k_inverse.h"
#include "mik_k_experiments.h"  // k_update_deep: left the library in round 6
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mik;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
template<class F> float timeit(F f, int reps){
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); f(); hipDeviceSynchronize();
  hipEventRecord(e0); for(int i=0;i<reps;i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms,e0,e1); return ms/reps;
}
template <int ABL> float run_deep(double* T, int Mp, int nblk, int kb, const double* Cold, const double* Cnew, const double* Rt, const double* Dinv, int tpb) {
  const long lt = (long)nblk * (nblk + 1) / 2, per = (lt + 7) / 8;
  const int gdeep = (int)(8 * ((per + tpb - 1) / tpb));
  (void)hipFuncSetAttribute((const void*)k_update_deep<true, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, MIK_UD_LDS_BYTES);
  int step = 0;
  return timeit([&]{ hipLaunchKernelGGL((k_update_deep<true, ABL>), dim3(gdeep + nblk), dim3(1024), MIK_UD_LDS_BYTES, 0, T, (long)Mp, nblk, kb, Cold, Cnew, Rt, Dinv, 0, -2, tpb, (step++) & 1, gdeep); }, 10);
}
int main(int argc,char**argv){
  const int Mp = argc>1? atoi(argv[1]) : 8064, nblk = Mp/128, kb = nblk/2;
  double *T,*Dinv,*Cold,*Cnew,*Rt;
  CK(hipMalloc(&T,sizeof(double)*(size_t)Mp*Mp)); CK(hipMalloc(&Dinv,131072));
  CK(hipMalloc(&Cold,sizeof(double)*(size_t)Mp*128)); CK(hipMalloc(&Cnew,sizeof(double)*(size_t)Mp*128)); CK(hipMalloc(&Rt,sizeof(double)*(size_t)Mp*128));
  { std::vector<double> b((size_t)Mp*128); srand(1);
    for(size_t i=0;i<b.size();++i) b[i]=(rand()/(double)RAND_MAX-0.5)*1e-3;
    CK(hipMemcpy(Cold,b.data(),b.size()*8,hipMemcpyHostToDevice)); CK(hipMemcpy(Cnew,b.data(),b.size()*8,hipMemcpyHostToDevice));
    CK(hipMemcpy(Rt,b.data(),b.size()*8,hipMemcpyHostToDevice)); CK(hipMemcpy(Dinv,b.data(),131072,hipMemcpyHostToDevice));
    CK(hipMemset(T,0,sizeof(double)*(size_t)Mp*Mp)); }
  const long lt=(long)nblk*(nblk+1)/2; const unsigned ug=(unsigned)(8*((lt+7)/8));
  const double fl = 2.0*128*128*128*(double)(lt - nblk);
  printf("Mp = %d, %d block columns, %ld upper tiles (%ld take a rank-128 update: %.2f GFLOP, %.0f MB read + written of T)\n", Mp, nblk, lt, lt - nblk, fl*1e-9, (lt - nblk)*0.262144);
  int step = 0;
  float ms=timeit([&]{hipLaunchKernelGGL((k_update<true,2>),dim3(ug),dim3(512),0,0,T,(long)Mp,nblk,kb,(const double*)Cold,(const double*)Cnew,(const double*)Rt,(const double*)Dinv,0,-2,(double*)nullptr,(double*)nullptr,(step++)&1);},10);
  printf("k_update<true,2> (two 8-wave blocks per CU)        : %7.1f us  %5.1f TFLOP/s\n",ms*1e3, fl/ms*1e-9);
  {  // one profiled launch: when each block's K loop and read-modify-write began and ended
    unsigned long long* pb; CK(hipMalloc(&pb, sizeof(unsigned long long) * (4 * ug + 4))); CK(hipMemset(pb, 0, sizeof(unsigned long long) * (4 * ug + 4)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(mik_upd_prof), &pb, sizeof(pb)));
    hipLaunchKernelGGL((k_update<true,2>),dim3(ug),dim3(512),0,0,T,(long)Mp,nblk,kb,(const double*)Cold,(const double*)Cnew,(const double*)Rt,(const double*)Dinv,0,-2,(double*)nullptr,(double*)nullptr,0);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hp(4 * (size_t)ug + 4); CK(hipMemcpy(hp.data(), pb, hp.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long* none = nullptr; CK(hipMemcpyToSymbol(HIP_SYMBOL(mik_upd_prof), &none, sizeof(none)));
    unsigned long long tmin = ~0ULL, tmax = 0; double kl = 0, rm = 0; long nb = 0;
    std::vector<double> kls, rms;
    for (unsigned b = 0; b < ug; ++b) { if (!hp[4*b]) continue; tmin = std::min(tmin, hp[4*b]); tmax = std::max(tmax, hp[4*b+2]); kl += (double)(hp[4*b+1]-hp[4*b]); rm += (double)(hp[4*b+2]-hp[4*b+1]); kls.push_back((double)(hp[4*b+1]-hp[4*b])); rms.push_back((double)(hp[4*b+2]-hp[4*b+1])); ++nb; }
    std::sort(kls.begin(), kls.end()); std::sort(rms.begin(), rms.end());
    (void)tmin; (void)tmax;
    printf("   profile of one launch (s_memtime ticks; %ld tiles with a product; the counters of different CUs are not aligned: sums only)\n", nb);
    printf("     K loop per tile: mean %.0f (median %.0f, 10 %% %.0f, 90 %% %.0f)   read-modify-write: mean %.0f (median %.0f, 10 %% %.0f, 90 %% %.0f)\n", kl/nb, kls[nb/2], kls[nb/10], kls[nb*9/10], rm/nb, rms[nb/2], rms[nb/10], rms[nb*9/10]);
    printf("     sum over tiles / 512 resident blocks = %.0f ticks of K loop + %.0f ticks of read-modify-write per slot\n", kl/512, rm/512);
    { double pro = 0; for (unsigned b = 0; b < ug; ++b) if (hp[4*b]) pro += (double)(hp[4*b+3] >> 8);
      printf("     kernel entry -> K loop: mean %.0f ticks per block\n", pro/nb); }
    printf("     calibration (block 0): %llu s_memtime ticks in %llu ticks of the 100 MHz clock = %.3f ns per tick\n", hp[4*(size_t)ug], hp[4*(size_t)ug+1], 10.0 * (double)hp[4*(size_t)ug+1] / (double)hp[4*(size_t)ug]);
  }
  std::vector<int> tpbs; for(int a=2;a<argc;++a) tpbs.push_back(atoi(argv[a])); if(tpbs.empty()) tpbs={1,2,4,8};
  for(int tpb: tpbs){
    printf("k_update_deep, %d tiles per block:\n", tpb);
#define RUN(ABL, WHAT) ms = run_deep<ABL>(T, Mp, nblk, kb, Cold, Cnew, Rt, Dinv, tpb); printf("   %-50s: %7.1f us  %5.1f TFLOP/s\n", WHAT, ms*1e3, fl/ms*1e-9);
    RUN(0, "as in the library");
    RUN(8, "no stagger");
    RUN(1, "no T loads");
    RUN(33, "no T loads, no stores");
    RUN(2, "no LDS-DMA");
    RUN(35, "no T loads, no stores, no LDS-DMA");
    RUN(12, "no stagger, no MFMAs");
    RUN(28, "no stagger, no MFMAs, no fragment reads");
    RUN(63, "nothing but barriers and waits");
  }
  return 0;
}
