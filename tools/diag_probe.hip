// Where do the ~1600 cycles of a pivot step of the diagonal-block inverse go?  The pipelined kernel (k_diag_inv_p) with pieces
// switched off (results are then wrong; only the clock is read), plus bare loops of its synchronisation pattern.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I pykrige_amd/csrc tools/diag_probe.hip -o tools/diag_probe
#include "mik_kernels.h"
#include "mik_k_experiments.h"  // k_diag_inv_t: left the library in round 6
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace mik;
enum { NO_BARRIER = 1, NO_READS = 2, NO_REST = 4, NO_RECIP = 8, NO_WRITES = 16, NO_FIX = 32 };

template <int MODE, int KB, int KBN, bool LAST>
__device__ __forceinline__ void pstep(double (&a)[8][8], const double (&rkraw)[8], const double (&ck)[8], double piv, double (&rkn)[8],
                                      double (&ckn)[8], double& pivn, int kr, int krn, int ty, int tx, double* rowk, double* colk, int nb) {
  const double pinv = (MODE & NO_RECIP) ? piv : pivot_recip(piv);
  double rk[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) rk[j] = rkraw[j] * pinv;
  const bool prow = (ty == kr), pcol = (tx == kr);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    double v = a[KB][j] - ck[KB] * rk[j];
    if (!(MODE & NO_FIX)) {
      if (j == KB) v = pcol ? -ck[KB] * pinv : v;
      v = prow ? ((j == KB && pcol) ? pinv : rk[j]) : v;
    }
    a[KB][j] = v;
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i == KB) continue;
    double v = a[i][KB] - ck[i] * rk[KB];
    if (!(MODE & NO_FIX)) v = pcol ? -ck[i] * pinv : v;
    a[i][KB] = v;
  }
  if (KBN != KB) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j != KB) a[KBN][j] -= ck[KBN] * rk[j];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (i != KB && i != KBN) a[i][KBN] -= ck[i] * rk[KBN];
  }
  if (!LAST) {
    if (!(MODE & NO_WRITES)) {
      if (ty == krn) {
#pragma unroll
        for (int j = 0; j < 8; ++j) rowk[nb * 128 + tx * 8 + j] = a[KBN][j];
      }
      if (tx == krn) {
#pragma unroll
        for (int i = 0; i < 8; ++i) colk[nb * 128 + ty * 8 + i] = a[i][KBN];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE & NO_BARRIER) __builtin_amdgcn_wave_barrier(); else __syncthreads();
    if (MODE & NO_READS) {
      pivn = a[KBN][KBN] + 1.0;
#pragma unroll
      for (int j = 0; j < 8; ++j) rkn[j] = a[KBN][j];
#pragma unroll
      for (int i = 0; i < 8; ++i) ckn[i] = a[i][KBN];
    } else {
      pivn = rowk[nb * 128 + krn * 8 + KBN];
#pragma unroll
      for (int j = 0; j < 8; ++j) rkn[j] = rowk[nb * 128 + tx * 8 + j];
#pragma unroll
      for (int i = 0; i < 8; ++i) ckn[i] = colk[nb * 128 + ty * 8 + i];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (!(MODE & NO_REST)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (i == KB || i == KBN) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if (j == KB || j == KBN) continue;
        a[i][j] -= ck[i] * rk[j];
      }
    }
  }
}
template <int MODE, int KB>
__device__ __forceinline__ void pgroup(double (&a)[8][8], double (&rA)[8], double (&cA)[8], double& pA, double (&rB)[8], double (&cB)[8],
                                       double& pB, int ty, int tx, double* rowk, double* colk) {
#pragma unroll 1
  for (int kr = 0; kr < 14; kr += 2) {
    pstep<MODE, KB, KB, false>(a, rA, cA, pA, rB, cB, pB, kr, kr + 1, ty, tx, rowk, colk, 1);
    pstep<MODE, KB, KB, false>(a, rB, cB, pB, rA, cA, pA, kr + 1, kr + 2, ty, tx, rowk, colk, 0);
  }
  pstep<MODE, KB, KB, false>(a, rA, cA, pA, rB, cB, pB, 14, 15, ty, tx, rowk, colk, 1);
  pstep<MODE, KB, (KB < 7 ? KB + 1 : KB), KB == 7>(a, rB, cB, pB, rA, cA, pA, 15, 0, ty, tx, rowk, colk, 0);
}
template <int MODE>
__global__ void __launch_bounds__(256) k_probe(const double* __restrict__ T, long ld, double* __restrict__ Dinv) {
  __shared__ double rowk[2 * 128], colk[2 * 128];
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  double a[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a[i][j] = T[(long)(ty + 16 * i) * ld + tx + 16 * j];
  if (ty == 0) {
#pragma unroll
    for (int j = 0; j < 8; ++j) rowk[tx * 8 + j] = a[0][j];
  }
  if (tx == 0) {
#pragma unroll
    for (int i = 0; i < 8; ++i) colk[ty * 8 + i] = a[i][0];
  }
  __syncthreads();
  double rA[8], cA[8], pA, rB[8], cB[8], pB = 0.0;
  pA = rowk[0];
#pragma unroll
  for (int j = 0; j < 8; ++j) rA[j] = rowk[tx * 8 + j];
#pragma unroll
  for (int i = 0; i < 8; ++i) cA[i] = colk[ty * 8 + i];
  pgroup<MODE, 0>(a, rA, cA, pA, rB, cB, pB, ty, tx, rowk, colk);
  pgroup<MODE, 1>(a, rA, cA, pA, rB, cB, pB, ty, tx, rowk, colk);
  pgroup<MODE, 2>(a, rA, cA, pA, rB, cB, pB, ty, tx, rowk, colk);
  pgroup<MODE, 3>(a, rA, cA, pA, rB, cB, pB, ty, tx, rowk, colk);
  pgroup<MODE, 4>(a, rA, cA, pA, rB, cB, pB, ty, tx, rowk, colk);
  pgroup<MODE, 5>(a, rA, cA, pA, rB, cB, pB, ty, tx, rowk, colk);
  pgroup<MODE, 6>(a, rA, cA, pA, rB, cB, pB, ty, tx, rowk, colk);
  pgroup<MODE, 7>(a, rA, cA, pA, rB, cB, pB, ty, tx, rowk, colk);
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) Dinv[(ty + 16 * i) * 128 + tx + 16 * j] = a[i][j];
}

// bare patterns, 128 rounds each, NT threads
template <int NT, int WHAT>  // 0: barrier only; 1: write 64 B by 16 threads + barrier + every thread reads 2 x 64 B; 2: (1) + dependent rcp/4 fma
__global__ void __launch_bounds__(NT) k_sync(double* out, int rounds) {
  __shared__ double buf[2][256];
  double acc = threadIdx.x * 1e-3 + 1.0, v[16];
  for (int r = 0; r < rounds; ++r) {
    const int nb = r & 1;
    if (WHAT >= 1) {
      if ((threadIdx.x >> 4) == (r & 15))
        for (int j = 0; j < 8; ++j) buf[nb][(threadIdx.x & 15) * 8 + j] = acc + j;
      if ((threadIdx.x & 15) == (r & 15))
        for (int j = 0; j < 8; ++j) buf[nb][128 + ((threadIdx.x >> 4) & 15) * 8 + j] = acc - j;
    }
    __syncthreads();
    if (WHAT >= 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = buf[nb][(threadIdx.x & 15) * 8 + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[8 + j] = buf[nb][128 + ((threadIdx.x >> 4) & 15) * 8 + j];
      double s = 0;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += v[j];
      acc = (WHAT >= 2) ? pivot_recip(s + 3.0) : s * 1e-3 + 1.0;
    }
  }
  out[threadIdx.x] = acc;
}
// WHAT 3: 64 (or nf) independent FMAs per round, no sync at all: the issue rate of one wave per SIMD
template <int NF>
__global__ void __launch_bounds__(256) k_fma(double* out, int rounds) {
  double a[NF];
#pragma unroll
  for (int i = 0; i < NF; ++i) a[i] = threadIdx.x + i;
  double x = 1.0000001, y = 1e-9;
  for (int r = 0; r < rounds; ++r) {
#pragma unroll
    for (int i = 0; i < NF; ++i) a[i] = __builtin_fma(a[i], x, y);
    __builtin_amdgcn_sched_barrier(0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NF; ++i) s += a[i];
  out[threadIdx.x] = s;
}

template <class F> float timeit(F f, int reps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  f(); hipDeviceSynchronize();
  hipEventRecord(e0); for (int i = 0; i < reps; i++) f(); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms / reps;
}
#define PROBE(MODE, NAME) { float ms = timeit([&]{ hipLaunchKernelGGL((k_probe<MODE>), dim3(1), dim3(256), 0, 0, (const double*)T, (long)128, D); }, 20); \
    printf("%-58s %7.1f us  %6.0f cycles/step\n", NAME, ms * 1e3, ms * 1e-3 * 2.4e9 / 128); }
int main() {
  double *T, *D; hipMalloc(&T, 128 * 128 * 8); hipMalloc(&D, 128 * 128 * 8);
  std::vector<double> h(128 * 128); srand(1);
  for (auto& x : h) x = (rand() / (double)RAND_MAX - 0.5) * 0.01;
  for (int i = 0; i < 128; ++i) h[i * 128 + i] = 1.0 + 0.1 * (i % 7);
  hipMemcpy(T, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  { float ms = timeit([&]{ hipLaunchKernelGGL((k_sync<64, 0>), dim3(1), dim3(64), 0, 0, D, 0); }, 50); printf("empty kernel launch-to-launch: %.1f us\n", ms * 1e3); }
  PROBE(0, "pipelined kernel, complete");
  PROBE(NO_BARRIER, "  wave barrier instead of workgroup barrier");
  PROBE(NO_READS, "  no LDS reads");
  PROBE(NO_WRITES, "  no LDS writes");
  PROBE(NO_REST, "  without the 49 plain FMAs");
  PROBE(NO_RECIP, "  without the reciprocal chain");
  PROBE(NO_FIX, "  without the pivot row / column selects");
  PROBE(NO_READS | NO_WRITES | NO_BARRIER, "  no LDS traffic, no barrier (arithmetic only)");
  PROBE(NO_REST | NO_RECIP | NO_FIX, "  synchronisation + 15 FMAs only");
  PROBE(NO_READS | NO_WRITES | NO_BARRIER | NO_RECIP | NO_FIX, "  64 FMAs + 8 muls per step, nothing else");
#define SYNC(NT, W, NAME) { float ms = timeit([&]{ hipLaunchKernelGGL((k_sync<NT, W>), dim3(1), dim3(NT), 0, 0, D, 128); }, 20); \
    printf("%-58s %7.1f us  %6.0f cycles/round\n", NAME, ms * 1e3, ms * 1e-3 * 2.4e9 / 128); }
  SYNC(256, 0, "128 x workgroup barrier, 4 waves");
  SYNC(1024, 0, "128 x workgroup barrier, 16 waves");
  SYNC(256, 1, "128 x (publish + barrier + 2 x 64 B reads), 4 waves");
  SYNC(256, 2, "128 x (publish + barrier + reads + reciprocal), 4 waves");
  SYNC(64, 1, "128 x (publish + barrier + reads), 1 wave");
  { float ms = timeit([&]{ hipLaunchKernelGGL((k_fma<64>), dim3(1), dim3(256), 0, 0, D, 128); }, 20);
    printf("%-58s %7.1f us  %6.0f cycles/round\n", "128 x 64 independent v_fma_f64, 1 wave per SIMD", ms * 1e3, ms * 1e-3 * 2.4e9 / 128); }
  { float ms = timeit([&]{ hipLaunchKernelGGL((k_fma<64>), dim3(1), dim3(256), 0, 0, D, 1280); }, 20);
    printf("%-58s %7.1f us  %6.0f cycles/round\n", "1280 x 64 independent v_fma_f64, 1 wave per SIMD", ms * 1e3, ms * 1e-3 * 2.4e9 / 1280); }
  // round 3: the blocked diagonal inverse (k_diag_inv_b), complete and with pieces switched off
  {
    double *Dv, *DvT; int* flag;
    hipMalloc(&Dv, 128 * 128 * 8); hipMalloc(&DvT, 128 * 128 * 8); hipMalloc(&flag, MIK_F_INTS * sizeof(int)); hipMemset(flag, 0, MIK_F_INTS * sizeof(int));
    const int lds = (int)(sizeof(double) * MIK_DIAGB_LDS_DOUBLES);
#define PROBEB(ABL, NAME) { (void)hipFuncSetAttribute((const void*)k_diag_inv_b<ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, lds); \
    float ms = timeit([&]{ hipLaunchKernelGGL((k_diag_inv_b<ABL>), dim3(1), dim3(256), lds, 0, (const double*)T, 128L, 0, 128, Dv, DvT, flag); }, 20); \
    printf("%-58s %7.1f us  %6.0f cycles/sub-step\n", NAME, ms * 1e3, ms * 1e-3 * 2.4e9 / 8); }
    PROBEB(0, "blocked inverse (8 x 16 pivots), complete");
    PROBEB(1, "  without the 16 wave-level pivot steps");
    PROBEB(2, "  without the rank-16 update (256 MFMAs per wave)");
    PROBEB(4, "  without the panel products");
    PROBEB(8, "  without publish / overwrite");
    PROBEB(16, "  without the barriers");
    PROBEB(1 | 2, "  without pivots and update");
    PROBEB(1 | 2 | 4 | 8, "  barriers only");
    { float ms = timeit([&]{ hipLaunchKernelGGL((k_diag_inv_t<16, 16>), dim3(1), dim3(256), 0, 0, (const double*)T, 128L, 0, 128, Dv, DvT, flag); }, 20);
      printf("%-58s %7.1f us\n", "k_diag_inv_t<16,16> (128 barrier-separated pivots)", ms * 1e3); }
  }
  return 0;
}
