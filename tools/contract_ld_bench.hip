// Does the leading dimension of A_inv / the RHS panel matter to k_contract?  Mp = 5120 gives rows 40960 B apart
// (2^13 x 5): every row of a 128-row K slice then falls into the same few L2 sets unless the cache hashes its index.
// Times the library's contraction kernel with lda = ldb = Mp + pad for several pads.  Run it under
//   rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum          (and a second pass: --pmc FETCH_SIZE)
// to see the L2 hit rate / fabric traffic per dispatch (dispatch order = the order printed here).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pykrige_amd/csrc -I include tools/contract_ld_bench.hip -o tools/contract_ld_bench
#include "mik_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mik;
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__);return 1;}}while(0)
__global__ void k_fill(double* p, size_t n, unsigned seed) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    unsigned x = (unsigned)(i * 2654435761u) ^ seed;
    x ^= x >> 13; x *= 0x5bd1e995u; x ^= x >> 15;
    p[i] = ((double)(x & 0xffffff) / 16777216.0 - 0.5) * 0.02;
  }
}
int main(int argc, char** argv) {
  const int Mp = argc > 1 ? atoi(argv[1]) : 5120, P = argc > 2 ? atoi(argv[2]) : 65536;
  const int nblk = Mp / 128, kend = Mp;
  const int pads[] = {0, 16};
  unsigned long long* queue; CK(hipMalloc(&queue, 64));
  double* part; CK(hipMalloc(&part, sizeof(double) * (size_t)P * nblk));
  double kext_sym = 0; for (int ib = 0; ib < nblk; ++ib) kext_sym += kend - ib * 128;
  const double fl_full = 2.0 * 128 * 128 * (double)kend * nblk * (P / 128), fl_sym = 2.0 * 128 * 128 * kext_sym * (P / 128);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int pad : pads) {
    const long ld = Mp + pad;
    double *T, *Bt;
    CK(hipMalloc(&T, sizeof(double) * (size_t)Mp * ld)); CK(hipMalloc(&Bt, sizeof(double) * (size_t)P * ld));
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, T, (size_t)Mp * ld, 1u);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, Bt, (size_t)P * ld, 2u);
    CK(hipDeviceSynchronize());
    std::vector<double> ref((size_t)P * nblk), got((size_t)P * nblk);
    for (int form = 0; form < 3; ++form) {
      float best = 1e30f;
      for (int rep = 0; rep < 3; ++rep) {
        hipMemsetAsync(queue, 0, 64, 0);
        hipEventRecord(e0);
        if (form == 0) hipLaunchKernelGGL((k_contract<true, 2, true>), dim3(512), dim3(512), 0, 0, (const double*)T, ld, (const double*)Bt, ld, part, P, nblk, kend, queue);
        else if (form >= 2) hipLaunchKernelGGL((k_contract<true, 2, true, true>), dim3(512), dim3(512), 0, 0, (const double*)T, ld, (const double*)Bt, ld, part, P, nblk, kend, queue);
        else hipLaunchKernelGGL((k_contract<false, 2, true>), dim3(512), dim3(512), 0, 0, (const double*)T, ld, (const double*)Bt, ld, part, P, nblk, kend, queue);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      printf("ld = Mp + %3d  %s : %.3f ms  executed %.2f TF/s\n", pad, form == 0 ? "sym tiles" : form == 2 ? "sym pairs" : form == 3 ? "pairs +0.26ms" : form == 4 ? "pairs +0.52ms" : form == 5 ? "pairs +0.13ms" : "full     ", best, (form != 1 ? fl_sym : fl_full) / best * 1e-9);
      if (form == 0) CK(hipMemcpy(ref.data(), part, ref.size() * 8, hipMemcpyDeviceToHost));
      if (form >= 2) {
        CK(hipMemcpy(got.data(), part, got.size() * 8, hipMemcpyDeviceToHost));
        double md = 0; for (size_t i = 0; i < ref.size(); ++i) md = fmax(md, fabs(ref[i] - got[i]));
        printf("                 sym pairs vs sym tiles partials: max|diff| %.3e\n", md);
      }
    }
    CK(hipFree(T)); CK(hipFree(Bt));
  }
  return 0;
}
