// ea_probe.hip -- calibration kernels for splitting a kernel's fabric reads (FETCH_SIZE: L2 misses, Infinity-Cache hits included) into
// Infinity-Cache hits and HBM reads by their average L2-miss latency (TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ, rocprofv3 --pmc):
//   k_probe_hbm   streams a 6 GiB buffer once      -> every L2 miss goes to HBM          (buffer >> 256 MiB Infinity Cache)
//   k_probe_mall  re-reads a 96 MiB buffer 40 x     -> L2 misses (96 MiB > 32 MiB of L2) hit the Infinity Cache after the first pass
// Both read with the access shape of the contraction's operand stream (16-byte loads, 128-byte rows) from 512 resident blocks.
// scripts/gpu_hbm_split.sh runs this and one bench step under the same counters; scripts/hbm_split.py does the arithmetic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void __launch_bounds__(512) k_probe_hbm(const double2* __restrict__ src, size_t n, double* __restrict__ out) {
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const double2 v = src[i];
    acc += v.x + v.y;
  }
  if (acc == 1.2345e300) out[0] = acc;
}
__global__ void __launch_bounds__(512) k_probe_mall(const double2* __restrict__ src, size_t n, int passes, double* __restrict__ out) {
  double acc = 0.0;
  for (int p = 0; p < passes; ++p)
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
      const double2 v = src[i];
      acc += v.x + v.y;
    }
  if (acc == 1.2345e300) out[0] = acc;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const size_t big = (size_t)6 << 30, small = (size_t)96 << 20;
  double2 *a = nullptr, *b = nullptr;
  double* out = nullptr;
  CK(hipMalloc(&a, big));
  CK(hipMalloc(&b, small));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(a, 0, big));
  CK(hipMemset(b, 0, small));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    float ms = 0.f;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_probe_hbm, dim3(512), dim3(512), 0, 0, a, big / sizeof(double2), out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_probe_hbm : %.1f GiB in %.2f ms = %.2f TB/s\n", big / 1073741824.0, ms, big / ms / 1e9);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_probe_mall, dim3(512), dim3(512), 0, 0, b, small / sizeof(double2), 40, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_probe_mall: 40 x %.0f MiB in %.2f ms = %.2f TB/s\n", small / 1048576.0, ms, 40.0 * small / ms / 1e9);
  }
  return 0;
}
