// Which CUs does a stream created with hipExtStreamCreateWithCUMask run on?  Every block records (XCC_ID, HW_ID) while it spins briefly;
// the host prints, per mask, the number of distinct (SE, CU) slots seen on each XCD.  Build: hipcc --offload-arch=gfx950 -O3 tools/probe_cumask.hip -o tools/probe_cumask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <set>
#include <vector>
__global__ void k(unsigned* out) {
  unsigned x, hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 20000) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = x; out[2 * blockIdx.x + 1] = hw; }
}
static void run(const char* what, const std::vector<uint32_t>& mask) {
  hipStream_t st;
  if (mask.empty()) hipStreamCreate(&st);
  else if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed\n", what); return; }
  const int nb = 4096;
  unsigned* d; hipMalloc(&d, 8 * nb);
  hipLaunchKernelGGL(k, dim3(nb), dim3(1024), 0, st, d);
  hipStreamSynchronize(st);
  std::vector<unsigned> h(2 * nb); hipMemcpy(h.data(), d, 8 * nb, hipMemcpyDeviceToHost);
  std::set<unsigned> cus[16];
  for (int b = 0; b < nb; ++b) cus[h[2 * b] & 15].insert(h[2 * b + 1] & 0x0000ff00u | ((h[2 * b + 1] >> 13) & 7) << 16);  // CU_ID bits 11:8, SH 12, SE 15:13
  printf("%-44s:", what); int tot = 0;
  for (int x = 0; x < 8; ++x) { printf(" xcd%d=%zu", x, cus[x].size()); tot += (int)cus[x].size(); }
  printf("  total %d\n", tot);
  hipFree(d); hipStreamDestroy(st);
}
int main() {
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount; printf("%d CUs\n", ncu);
  run("no mask", {});
  std::vector<uint32_t> all((ncu + 31) / 32, 0xffffffffu);
  run("all bits set", all);
  { auto m = all; for (int i = ncu - 16; i < ncu; ++i) m[i / 32] &= ~(1u << (i % 32)); run("top 16 bits cleared", m); }
  { auto m = all; for (int i = 0; i < ncu; ++i) if (i % 32 >= 30) m[i / 32] &= ~(1u << (i % 32)); run("bits 30, 31 of every 32 cleared", m); }
  { auto m = all; for (int i = 0; i < 16; ++i) m[i / 32] &= ~(1u << (i % 32)); run("low 16 bits cleared", m); }
  return 0;
}
