// tools/mw_ldl_bench.hip -- round 6 GATE of the review's item 2: the moving window's per-point elimination on the matrix cores.
//
// One wavefront per point eliminates the (k + 3) x (k + 3) system [C R; R^T 0] (C = the k x k covariance block of the window, R = the three
// right-hand-side columns: c0, 1, v) by SYMMETRIC BLOCK GAUSSIAN ELIMINATION WITH 4 x 4 PIVOT BLOCKS on v_mfma_f64_4x4x4_4b (four independent
// 4 x 4 x 4 products per instruction, one accumulator double per lane); the last block ends as -R^T C^-1 R: the five inner products z and
// sigma^2 are formed from (what k_mw_chol's LDL^T leaves, mik_k_mw_chol.h).  The whole upper block triangle lives in accumulator registers:
//   * block (J, I), J <= I, is kept in the instruction's D layout (lane 16 i + 4 b + j = entry (i, j) of the block in position b); a SLOT is the
//     four blocks (J, 4 Ig + b), b = 0 .. 3.  In that layout a block IS a B operand (lane 16 k + 4 b + j) and, read as its own transpose, an A
//     operand (lane 16 k + 4 b + i): no layout moves between the steps.
//   * pivot block p:  S = A_pp^-1 (4 x 4 SPD inverse, every lane redundantly, from readlane'd entries; 2 x 2 block form, two reciprocals);
//     Y[Ig] = S . A_p,4Ig..  (one instruction per column group);  for every row J > p:  A_J,4Ig.. -= A_pJ^T . Y[Ig]  -- the A operand is the
//     block (p, J) broadcast to the four block positions (ds_bpermute: the LDS crossbar, no LDS memory).
// Counted for k = 100 (26 blocks): 1 120 matrix instructions (17.9 k cycles at 16 each), 650 ds_bpermute, 25 x ~130 vector instructions for the
// pivot inverses; k = 50 (14 blocks): 228 matrix instructions; k = 200 (51 blocks) needs 363 slots = 726 registers: it does not fit one wavefront.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mw_ldl_bench.hip -o tools/mw_ldl_bench      Run: tools/mw_ldl_bench [points]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__host__ __device__ inline double frac(double x) { return x - floor(x); }
// the synthetic window of point `pt`: k stations on a line at x_r, exponential covariance + a nugget on the diagonal (SPD), three right-hand sides
__host__ __device__ inline double entry(int k, int n4, int r, int c, int pt) {
  const int nc4 = n4 - 4;  // rows of the covariance part, padded to a multiple of 4
  if (r < nc4 && c < nc4) {
    if (r >= k || c >= k) return r == c ? 1.0 : 0.0;  // identity padding
    const double xr = frac(r * 0.6180339887 + pt * 1e-3), xc = frac(c * 0.6180339887 + pt * 1e-3);
    return exp(-3.0 * fabs(xr - xc)) + (r == c ? 0.1 : 0.0);
  }
  if (r >= nc4 && c >= nc4) return 0.0;
  const int s = (r < nc4 ? r : c), q = (r < nc4 ? c : r) - nc4;  // station s, right-hand side q
  if (s >= k || q == 3) return 0.0;
  const double xs = frac(s * 0.6180339887 + pt * 1e-3);
  return q == 0 ? exp(-3.0 * fabs(xs - 0.5)) : q == 1 ? 1.0 : sin(7.0 * xs);
}

template <int NB> struct Lay {
  static constexpr int NG = (NB + 3) / 4;
  // slots before row J: sum_{j < J} (NG - j / 4), in closed form (q = J / 4 whole groups of four rows, then J % 4 rows of the next)
  static constexpr int base(int J) { return 4 * (J / 4) * NG - 2 * (J / 4) * (J / 4 - 1) + (J % 4) * (NG - J / 4); }
  static constexpr int idx(int J, int Ig) { return base(J) + Ig - J / 4; }
  static constexpr int NSLOT = base(NB);
};
static_assert(Lay<26>::NSLOT == 110 && Lay<14>::NSLOT == 38 && Lay<26>::base(5) == 4 * 7 + 6, "slot layout");

__device__ __forceinline__ double bperm(int addr, double v) {
  const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(v)), hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double rdlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// MODE 0 = set-up + store only (the baseline to subtract), 1 = the elimination, 2 = the elimination without the pivot inverses (S = I: wrong numbers,
// the matrix pipe's share), 3 = without the matrix instructions (the vector / crossbar share)
template <int NB, int MODE, int WPS>
__global__ void __launch_bounds__(256, WPS) k_ldl(int k, int npts, double* __restrict__ out) {
  using L = Lay<NB>;
  constexpr int NG = L::NG;
  const int lane = threadIdx.x & 63, pt = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= npts) return;
  const int li = lane >> 4, lb = (lane >> 2) & 3, lj = lane & 3;
  double s[L::NSLOT];
#pragma unroll
  for (int J = 0; J < NB; ++J)
#pragma unroll
    for (int Ig = J / 4; Ig < NG; ++Ig) {
      const int I = 4 * Ig + lb;
      s[L::idx(J, Ig)] = (I >= J && I < NB) ? entry(k, 4 * NB, 4 * J + li, 4 * I + lj, pt) : 0.0;
    }
  if (MODE != 0) {
    int addr[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) addr[b] = ((lane & ~12) | (b << 2)) * 4;
#pragma unroll
    for (int p = 0; p < NB - 1; ++p) {
      // ---- S = A_pp^-1: the block sits in slot (p, p / 4), position p % 4, lanes 16 i + 4 (p % 4) + j
      const double d = s[L::idx(p, p / 4)];
      const int b0 = 4 * (p & 3);
      double S00, S01, S02, S03, S11, S12, S13, S22, S23, S33;
      if (MODE == 2) {
        S00 = S11 = S22 = S33 = 1.0;
        S01 = S02 = S03 = S12 = S13 = S23 = 0.0;
      } else {
        const double a00 = rdlane(d, b0), a01 = rdlane(d, b0 + 1), a02 = rdlane(d, b0 + 2), a03 = rdlane(d, b0 + 3);
        const double a11 = rdlane(d, 16 + b0 + 1), a12 = rdlane(d, 16 + b0 + 2), a13 = rdlane(d, 16 + b0 + 3);
        const double a22 = rdlane(d, 32 + b0 + 2), a23 = rdlane(d, 32 + b0 + 3), a33 = rdlane(d, 48 + b0 + 3);
        // [P Q; Q^T R]^-1 with P = [a00 a01; a01 a11], Q = [a02 a03; a12 a13], R = [a22 a23; a23 a33]
        const double ip = 1.0 / (a00 * a11 - a01 * a01);
        const double p00 = a11 * ip, p01 = -a01 * ip, p11 = a00 * ip;                                  // P^-1
        const double w00 = p00 * a02 + p01 * a12, w01 = p00 * a03 + p01 * a13, w10 = p01 * a02 + p11 * a12, w11 = p01 * a03 + p11 * a13;  // W = P^-1 Q
        const double t00 = a22 - (a02 * w00 + a12 * w10), t01 = a23 - (a02 * w01 + a12 * w11), t11 = a33 - (a03 * w01 + a13 * w11);      // T = R - Q^T W
        const double it = 1.0 / (t00 * t11 - t01 * t01);
        S22 = t11 * it, S23 = -t01 * it, S33 = t00 * it;                                                // T^-1
        S02 = -(w00 * S22 + w01 * S23), S03 = -(w00 * S23 + w01 * S33), S12 = -(w10 * S22 + w11 * S23), S13 = -(w10 * S23 + w11 * S33);  // -W T^-1
        S00 = p00 - (S02 * w00 + S03 * w01), S01 = p01 - (S02 * w10 + S03 * w11), S11 = p11 - (S12 * w10 + S13 * w11);                   // P^-1 + W T^-1 W^T
      }
      // A operand of S: lane (k = li, i = lj) takes S[lj][li] (symmetric)
      const double r0 = lj == 0 ? S00 : lj == 1 ? S01 : lj == 2 ? S02 : S03, r1 = lj == 0 ? S01 : lj == 1 ? S11 : lj == 2 ? S12 : S13;
      const double r2 = lj == 0 ? S02 : lj == 1 ? S12 : lj == 2 ? S22 : S23, r3 = lj == 0 ? S03 : lj == 1 ? S13 : lj == 2 ? S23 : S33;
      const double Sa = li == 0 ? r0 : li == 1 ? r1 : li == 2 ? r2 : r3;
      // ---- Y[Ig] = -(S . A_p,4Ig..) for the column groups that hold a block beyond p
      double Y[NG];
#pragma unroll
      for (int Ig = (p + 1) / 4; Ig < NG; ++Ig) {
        const double y = MODE == 3 ? s[L::idx(p, Ig)] * Sa : __builtin_amdgcn_mfma_f64_4x4x4f64(Sa, s[L::idx(p, Ig)], 0.0, 0, 0, 0);
        Y[Ig] = -y;
      }
      // ---- rows beyond p: A_J,4Ig.. += A_pJ^T . Y[Ig], A_pJ broadcast to the four block positions
#pragma unroll
      for (int J = p + 1; J < NB; ++J) {
        const double a = bperm(addr[J & 3], s[L::idx(p, J / 4)]);
#pragma unroll
        for (int Ig = J / 4; Ig < NG; ++Ig) {
          if (MODE == 3) s[L::idx(J, Ig)] += a * Y[Ig];
          else s[L::idx(J, Ig)] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, Y[Ig], s[L::idx(J, Ig)], 0, 0, 0);
        }
      }
    }
  }
  // the last block (position (NB - 1) % 4 of the last slot): -R^T C^-1 R
  double acc = s[L::idx(NB - 1, NG - 1)];
  if (MODE == 0) {  // keep every slot alive
#pragma unroll
    for (int q = 0; q < L::NSLOT; ++q) acc += s[q];
  }
  if (lb == ((NB - 1) & 3)) out[(long)pt * 16 + 4 * li + lj] = acc;
}

template <class F> static float timeit(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  f();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

// the same Schur complement on the host (Cholesky of C, forward substitution of R), point `pt`
static void host_ref(int k, int n4, int pt, double* q9) {
  const int nc = n4 - 4;
  std::vector<double> c((size_t)nc * nc), r((size_t)nc * 3);
  for (int i = 0; i < nc; ++i) {
    for (int j = 0; j < nc; ++j) c[(size_t)i * nc + j] = entry(k, n4, i, j, pt);
    for (int q = 0; q < 3; ++q) r[(size_t)i * 3 + q] = entry(k, n4, i, nc + q, pt);
  }
  for (int j = 0; j < nc; ++j) {  // Cholesky, lower
    for (int t = 0; t < j; ++t)
      for (int i = j; i < nc; ++i) c[(size_t)i * nc + j] -= c[(size_t)i * nc + t] * c[(size_t)j * nc + t];
    const double dd = sqrt(c[(size_t)j * nc + j]);
    for (int i = j; i < nc; ++i) c[(size_t)i * nc + j] /= dd;
  }
  for (int q = 0; q < 3; ++q)
    for (int i = 0; i < nc; ++i) {
      double v = r[(size_t)i * 3 + q];
      for (int t = 0; t < i; ++t) v -= c[(size_t)i * nc + t] * r[(size_t)t * 3 + q];
      r[(size_t)i * 3 + q] = v / c[(size_t)i * nc + i];
    }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      double v = 0.0;
      for (int i = 0; i < nc; ++i) v += r[(size_t)i * 3 + a] * r[(size_t)i * 3 + b];
      q9[3 * a + b] = -v;
    }
}

template <int NB, int WPS> static int run(int k, int npts, double* out) {
  const unsigned grid = (unsigned)((npts + 3) / 4);
  printf("k = %d: order %d = %d blocks of 4, %d accumulator slots (%d VGPRs), %d wavefronts per SIMD asked for\n", k, 4 * NB, NB, Lay<NB>::NSLOT,
         2 * Lay<NB>::NSLOT, WPS);
  hipFuncAttributes fa;
  CK(hipFuncGetAttributes(&fa, (const void*)k_ldl<NB, 1, WPS>));
  printf("   elimination kernel: %d registers, %zu B scratch\n", fa.numRegs, (size_t)fa.localSizeBytes);
  hipLaunchKernelGGL((k_ldl<NB, 1, WPS>), dim3(grid), dim3(256), 0, 0, k, npts, out);
  CK(hipDeviceSynchronize());
  std::vector<double> h(16 * 3);
  double worst = 0.0, scale = 0.0;
  for (int pt : {0, npts / 2, npts - 1}) {
    CK(hipMemcpy(h.data(), out + (long)pt * 16, 16 * sizeof(double), hipMemcpyDeviceToHost));
    double q9[9];
    host_ref(k, 4 * NB, pt, q9);
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        worst = std::max(worst, fabs(h[4 * a + b] - q9[3 * a + b]));
        scale = std::max(scale, fabs(q9[3 * a + b]));
      }
  }
  printf("   -R^T C^-1 R against the host's Cholesky, three points: max|diff| %.3e (max|value| %.3e)\n", worst, scale);
  const float t0 = timeit([&] { hipLaunchKernelGGL((k_ldl<NB, 0, WPS>), dim3(grid), dim3(256), 0, 0, k, npts, out); }, 5);
  const float t1 = timeit([&] { hipLaunchKernelGGL((k_ldl<NB, 1, WPS>), dim3(grid), dim3(256), 0, 0, k, npts, out); }, 5);
  const float t2 = timeit([&] { hipLaunchKernelGGL((k_ldl<NB, 2, WPS>), dim3(grid), dim3(256), 0, 0, k, npts, out); }, 5);
  const float t3 = timeit([&] { hipLaunchKernelGGL((k_ldl<NB, 3, WPS>), dim3(grid), dim3(256), 0, 0, k, npts, out); }, 5);
  const double per = 1e6 / npts;
  printf("   per 1e6 points: set-up + store %.2f ms; with the elimination %.2f ms => elimination %.2f ms; without the pivot inverses %.2f ms; "
         "vector instructions in place of the matrix ones %.2f ms\n", t0 * per, t1 * per, (t1 - t0) * per, (t2 - t0) * per, (t3 - t0) * per);
  return 0;
}

int main(int argc, char** argv) {
  const int npts = argc > 1 ? atoi(argv[1]) : 200000;
  double* out;
  CK(hipMalloc(&out, sizeof(double) * 16 * (size_t)npts));
  // k + 3 rows (c0, 1, v) in a last block of their own: k = 50 -> 13 + 1 blocks, k = 100 -> 25 + 1 (k = 200 -> 51 blocks: 363 slots, no fit)
  if (run<14, 4>(50, npts, out)) return 1;
  if (run<26, 2>(100, npts, out)) return 1;
  if (run<20, 2>(72, npts, out)) return 1;
  printf("current library (profiles/r05_moving_window_lean_setup.txt, r05_mw_chol_phase_profile.txt): whole solve k = 50 6.9 ms, k = 100 40.1 ms per 1e6 points, of which "
         "the elimination 47 - 50 %% (k = 72 .. 96) / 66 %% (k = 100) = ~3.4 / ~26.5 ms\n");
  return 0;
}
