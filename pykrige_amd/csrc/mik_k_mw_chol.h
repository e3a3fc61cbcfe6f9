// mik_k_mw_chol.h -- moving-window kriging: the per-point argument block, the local-matrix entry, and k_mw_chol, the LDL^T solver in
// registers.  A template only: its 21 thread-grid / register-tile classes x 5 variogram forms are instantiated by mik_mw_chol.hip,
// which pykrige_amd/build.py compiles as four translation units (they are three quarters of the library's build time).
#pragma once
#include "mik_dev.h"

namespace mik {

struct MwArgs {
  const double *sx, *sy, *sz;  // station coordinates (adjusted); geographic: lon, lat in degrees
  int mode;                    // 2 / 3 = Euclidean dimension, 1 = geographic (great-circle degrees)
  const double* gtab;          // custom variogram: gamma of the K x K station pairs of every point (host-mapped), else NULL
  int K, npt;
  const int* idx;
  const double* dist;
  const double* Z;
  Vario v;
  int exact;
  double eps;
  double* z;
  double* ss;
  int* flag;
};

// variogram selected at run time (a wave-uniform switch; the moving-window kernels are not instantiated per model)
__device__ __forceinline__ double vario_dyn(const Vario& v, double d, double d2) {
  switch (v.model) {
    case 0: return vario<0, false>(v, d, d2);
    case 1: return vario<1, false>(v, d, d2);
    case 2: return vario<2, false>(v, d, d2);
    case 3: return vario<3, false>(v, d, d2);
    case 4: return vario<4, false>(v, d, d2);
    default: return vario<5, false>(v, d, d2);
  }
}
// entry (r, c), r != c, of a point's local kriging matrix: -gamma(distance between two selected stations), the value
// a_all[sel[r], sel[c]] of the reference (ok.py:626-648 then cok.pyx:138-147) computed from the coordinates, so that the
// moving window needs no N x N matrix.  (x, y, z) = adjusted coordinates, or (lon, cos lat, sin lat) when geographic.
__device__ __forceinline__ double mw_entry(const Vario& v, int mode, double x1, double y1, double z1, double x2, double y2,
                                           double z2) {
  double d, d2;
  if (mode == 1) {
    d = gc_dist(x1, y1, z1, x2, y2, z2);
    d2 = d * d;
  } else {
    const double dx = x1 - x2, dy = y1 - y2, dz = z1 - z2;  // z = 0 in 2-D
    d2 = dx * dx + dy * dy + dz * dz;
    d = sqrt(d2);
  }
  return -vario_dyn(v, d, d2);
}

// The same with the variogram model a COMPILE-TIME constant (MODEL >= 0; Euclidean coordinates): round 4.  mw_entry inlines the
// great-circle distance and all six models -- ~960 instructions per call site -- and k_mw_chol calls it once per register-tile
// element: its {8,13} class was 195 000 instructions (1.26 MB) of straight-line set-up code in front of a 4 000-instruction
// elimination loop, every point streaming it through a 64 KB instruction cache.  With the model fixed an entry is ~40 instructions.
template <int MODEL>
__device__ __forceinline__ double mw_entry_t(const Vario& v, int mode, double x1, double y1, double z1, double x2, double y2, double z2,
                                             const ExpTab* tab = nullptr) {
  if (MODEL < 0) return mw_entry(v, mode, x1, y1, z1, x2, y2, z2);
  const double dx = x1 - x2, dy = y1 - y2, dz = z1 - z2;  // z = 0 in 2-D
  const double d2 = dx * dx + dy * dy + dz * dz;
  // (round 5: reciprocal-multiply form and the 19-instruction exp -- <= 2 ulp on an entry, 1e-16 relative, against a 1e-8 bar; the
  // entry was ~124 instructions with the library's exp and the reference's divisions: half of a point's instructions at k = 100)
  return -vario<(MODEL < 0 ? 0 : MODEL), true, true>(v, sqrt(d2), d2, tab);
}

// Per-point solve WITHOUT pivot search, default of the moving window: LDL^T of the SPD-shifted station block in registers.
// The shifted system (k_mw_solve above) reads  C lam + mu 1 = bt,  1.lam = 1  with C = s 11^T - Gamma (covariances, SPD) and
// bt = b + s 1.  With C = L D L^T and the three forward-substituted vectors y_q = L^-1 {bt, 1, Z} everything the reference
// returns is a D^-1-weighted inner product G_pq = y_p . D^-1 y_q  (= B_p^T C^-1 B_q):
//     mu = (G_01 - 1) / G_11,   z = Z.lam = G_02 - mu G_12,   sigma^2 = -lam.b - mu = -(G_00 - mu G_01) + s - mu
// -- no back substitution, no solution vector.  A point is worked on by a G x G thread grid; thread (ty, tx) keeps the
// LOWER-triangle elements (ty + G i, tx + G j), j <= i < RI, in registers (cyclic: balanced while the trailing matrix
// shrinks): RI (RI + 1) / 2 FMAs per thread and step on a matrix that loses a row and a column per step -- about a sixth of
// the multiply-adds of the Gauss-Jordan form.  The three right-hand sides ride along as extra ROWS (threads ty = 0, 1, 2):
// the elimination forward-substitutes them.  The step loop is unrolled over the local tile index, so every register index
// is a compile-time constant (the Gauss-Jordan kernel selects its pivot row / column out of the tile with v_cndmask chains,
// which cost more than its FMAs).  Per step the G owners of column c publish it through double-buffered LDS; with at most 64
// threads per point the point lives inside one wavefront and no workgroup barrier is needed at all.
// A non-positive pivot raises flag bit 1 (the host reruns the call with the pivoted kernel).
// (second launch-bound = wavefronts per SIMD the register allocation must allow: the one-wavefront-per-point classes beyond
// RI = 12 otherwise take 256 VGPRs + a few AGPRs, which halves the occupancy -- measured 2 x slower)
#ifndef MIK_MWC_WAVES
#define MIK_MWC_WAVES(G, RI)                                                                                                        \
  (((G) == 4 && (RI) >= 11) ? 2 : ((G) == 4 && (RI) >= 9) ? 3 : ((G) == 4 && (RI) >= 7) ? 4 : ((G) == 4 && (RI) >= 5) ? 5 :          \
   ((G) == 8 && (RI) >= 13) ? 2 : ((G) == 8 && (RI) == 10) ? 3 : ((G) == 8 && (RI) == 8) ? 4 : ((G) == 8 && (RI) == 6) ? 5 :         \
   ((G) == 16 && (RI) == 8) ? 4 : ((G) == 16 && ((RI) == 9 || (RI) == 10)) ? 3 : ((G) == 16 && (RI) >= 11) ? 2 : 1)
// lean update (row factors read from LDS as they are used instead of held: RI fewer live doubles) where it buys a wavefront per
// SIMD; elsewhere it costs 1-2 % (profiles/r03_mw_classes_after_kernel_changes.txt)
#define MIK_MWC_LEAN(G, RI)                                                                                                         \
  (((G) == 4 && (RI) >= 6) || ((G) == 8 && ((RI) >= 13 || (RI) == 10 || (RI) == 8 || (RI) == 6)) || ((G) == 16 && ((RI) == 8 || (RI) == 10 || (RI) >= 13)) ||  \
   ((G) == 32 && (RI) == 8))
#endif
template <int G, int RI, int MODEL = -1>
__global__ void __launch_bounds__((G * G < 256) ? 256 : G * G, MIK_MWC_WAVES(G, RI)) k_mw_chol(MwArgs a) {
  extern __shared__ double mw_lds[];
  constexpr int T = G * G, NT = T < 256 ? 256 : T, NB = G * RI, ACOL = NB + 4;
  const int K = a.K;
  const int g = threadIdx.x / T, lt = threadIdx.x % T, ty = lt / G, tx = lt % G;
  constexpr int PER = 2 * ACOL + 9 * NB;
  double* acol = mw_lds + (long)g * PER;  // [2][ACOL]: column c of the trailing matrix by global row, right-hand-side rows at NB..NB+2
  double* csx = acol + 2 * ACOL;
  double* csy = csx + NB;
  double* csz = csy + NB;
  double* bvec = csz + NB;
  double* zsel = bvec + NB;
  double* ylog = zsel + NB;  // [4][NB]: per step c the three eliminated right-hand-side entries y_q(c) and 1 / d(c)
  const long pt = (long)blockIdx.x * (NT / T) + g;
  const bool live = pt < a.npt;
  auto sync = [&]() {
    if (T <= 64) {  // the point's threads are lanes of one wavefront: LDS operations of a wave complete in order
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
      __syncthreads();
    }
  };
  for (int r = lt; r < NB; r += T) {
    double x = 0.0, y = 0.0, z = 0.0, b = 0.0, zv = 0.0;
    if (live && r < K) {
      const int st = a.idx[pt * K + r];
      x = a.sx[st];
      y = a.sy[st];
      z = (a.mode == 3) ? a.sz[st] : 0.0;
      if (a.mode == 1) {
        const double lat = y * MIK_PI / 180.0;
        y = cos(lat);
        z = sin(lat);
      }
      b = a.dist[pt * K + r];  // dist holds b = -gamma(d), 0 on an exact hit (k_mw_rhs)
      zv = a.Z[st];
    }
    csx[r] = x, csy[r] = y, csz[r] = z, bvec[r] = b, zsel[r] = zv;
  }
  sync();
  double shift;
  if (a.v.model >= 2) {
    shift = a.v.p0 + a.v.p2;
  } else {
    double gmax = 0.0;
    for (int r = 0; r < K; ++r) gmax = fmax(gmax, -bvec[r]);
    shift = 4.0 * gmax;
  }
  if (!(shift > 0.0)) shift = 1.0;
  ExpTab etab;  // the lean exp's constants in scalar registers for the whole set-up -- only where the model takes an exponential (in the
                // dynamic form and the others the sixteen register pairs cost more than they save: k = 100 dynamic 54 -> 64 ms)
  constexpr bool USE_TAB = MODEL == 2 || MODEL == 4;
  if (USE_TAB) etab = exp_tab_load();
  double m[RI][RI], rhs[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int row = ty + G * i;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      const int col = tx + G * j;
      double v = (row == col) ? 1.0 : 0.0;  // padding rows / columns: identity
      if (row < K && col < K)
        v = (row == col) ? shift : shift + mw_entry_t<MODEL>(a.v, a.mode, csx[row], csy[row], csz[row], csx[col], csy[col], csz[col], USE_TAB ? &etab : nullptr);
      m[i][j] = v;
    }
  }
#pragma unroll
  for (int j = 0; j < RI; ++j) {
    const int col = tx + G * j;
    double v = 0.0;
    if (col < K) v = (ty == 0) ? bvec[col] + shift : (ty == 1) ? 1.0 : (ty == 2) ? zsel[col] : 0.0;
    rhs[j] = v;
  }
  int bad = 0;
#pragma unroll
  for (int cc = 0; cc < RI; ++cc) {
    for (int cx = 0; cx < G; ++cx) {
      const int c = cc * G + cx;
      if (c >= K) break;  // uniform over the block
      double* ab = acol + (c & 1) * ACOL;
      if (tx == cx) {
        // rows <= c of the diagonal local tile are finished: their column entries are published as zeros, so that neither the row factors
        // nor the column factors read back need a select of their own (round 6: one select at the G publishers instead of two in every thread)
#pragma unroll
        for (int i = cc; i < RI; ++i) ab[ty + G * i] = (i == cc && ty <= cx) ? 0.0 : m[i][cc];
        if (ty < 3) ab[NB + ty] = rhs[cc];
        // the pivot's owner (thread (cx, cx), local tile element (cc, cc)) publishes its reciprocal as well: one wavefront
        // per step pays for it instead of every one (this kernel is instruction-issue bound: round 3)
        if (ty == cx) ab[NB + 3] = pivot_recip(m[cc][cc]);
      }
      sync();
      const double inv = ab[NB + 3];
      if (!(inv > 0.0) || !(inv < 1e300)) bad = 2;  // a non-positive (or vanished) pivot
      double u[RI], w[RI];
#pragma unroll
      for (int i = cc; i < RI; ++i) {
        if (!MIK_MWC_LEAN(G, RI)) u[i] = ab[ty + G * i] * inv;
        w[i] = ab[tx + G * i];
      }
      const double ur = (ty < 3 ? ab[NB + ty] : 0.0) * inv;
      // the five inner products z and sigma^2 are made of, sum_c y_p(c) y_q(c) / d(c), are formed ONCE at the end from this log
      // (every thread used to accumulate all five in every step)
      if (lt < 4) ylog[lt * NB + c] = (lt < 3) ? ab[NB + lt] : inv;
      if (MIK_MWC_LEAN(G, RI)) {  // the largest one-wavefront tiles: the row factors are read as they are used (RI fewer live doubles)
#pragma unroll
        for (int i = cc; i < RI; ++i) {
          const double ui = ab[ty + G * i] * inv;
#pragma unroll
          for (int j = cc; j <= i; ++j) m[i][j] -= ui * w[j];
        }
      } else {
#pragma unroll
        for (int i = cc; i < RI; ++i)
#pragma unroll
          for (int j = cc; j <= i; ++j) m[i][j] -= u[i] * w[j];
      }
#pragma unroll
      for (int j = cc; j < RI; ++j) rhs[j] -= ur * w[j];
    }
  }
  sync();
  double g00 = 0.0, g01 = 0.0, g11 = 0.0, g02 = 0.0, g12 = 0.0;
  for (int c = lt; c < K; c += T) {
    const double y0 = ylog[c], y1 = ylog[NB + c], y2 = ylog[2 * NB + c], inv = ylog[3 * NB + c];
    g00 += y0 * y0 * inv;
    g01 += y0 * y1 * inv;
    g11 += y1 * y1 * inv;
    g02 += y0 * y2 * inv;
    g12 += y1 * y2 * inv;
  }
  constexpr int W = T < 64 ? T : 64;  // lanes of one wavefront that belong to this point
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) {
    g00 += __shfl_xor(g00, o);
    g01 += __shfl_xor(g01, o);
    g11 += __shfl_xor(g11, o);
    g02 += __shfl_xor(g02, o);
    g12 += __shfl_xor(g12, o);
  }
  if (T > 64) {  // several wavefronts per point: their partial sums meet in LDS (the column buffers are free now)
    const int wv = lt >> 6;
    if ((lt & 63) == 0) {
      acol[5 * wv + 0] = g00, acol[5 * wv + 1] = g01, acol[5 * wv + 2] = g11, acol[5 * wv + 3] = g02, acol[5 * wv + 4] = g12;
    }
    __syncthreads();
    if (lt == 0) {
      g00 = g01 = g11 = g02 = g12 = 0.0;
      for (int q = 0; q < T / 64; ++q) {
        g00 += acol[5 * q], g01 += acol[5 * q + 1], g11 += acol[5 * q + 2], g02 += acol[5 * q + 3], g12 += acol[5 * q + 4];
      }
    }
  }
  if (live && lt == 0) {
    const double mu = (g01 - 1.0) / g11;
    a.z[pt] = g02 - mu * g12;
    a.ss[pt] = -(g00 - mu * g01) + shift - mu;
    if (bad || !(g11 > 0.0)) atomicOr(a.flag, 2);
  }
}

}  // namespace mik
