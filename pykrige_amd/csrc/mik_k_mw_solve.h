// mik_k_mw_solve.h -- moving-window kriging: k_mw_solve, the per-point Gauss-Jordan solver with implicit partial pivoting (the fallback of the LDL^T
// kernels: hole-effect and custom variograms, non-positive pivots).  A template only; its twelve instantiations are mik_mw_solve.hip (round 6: they were
// two thirds of mik_mw.hip's 33 s of compile time, the longest unit of the parallel build).
#pragma once
#include "mik_dev.h"
#include "mik_k_mw_chol.h"

namespace mik {

// Per-point solve, register tiled.  A point is worked on by a GY x GX thread grid; thread (ty, tx) keeps the elements
// (ty + GY i, tx + GX j), i < RI, j < CJ, of the augmented (k+1) x (k+2) system in registers (cyclic distribution: the work
// stays balanced while the elimination shrinks).  Gauss-Jordan with implicit partial pivoting: at step c the pivot is
// the largest |a[r][c]| over the rows not used yet (the rows dgesv would look at), the pivot row and the multiplier
// column go through LDS once (RI + CJ reads per thread for RI x CJ FMAs), rows are never moved, columns <= c are left
// alone.  Two barriers per step.  x[c] = rhs[perm[c]] / pivot[c] at the end; z = x.Z[sel], ss = -x.b.
template <int GY, int GX, int RI, int CJ, bool PIV>
__global__ void __launch_bounds__(256) k_mw_solve(MwArgs a) {
  extern __shared__ double mw_lds[];
  constexpr int T = GY * GX, PPB = 256 / T, W = T < 64 ? T : 64, NW = T / W, CJP = (CJ + 1) & ~1;
  static_assert(RI % 2 == 0 && GY * RI <= 255 && GY <= 16, "row tile");
  const int K = a.K, nb = K + 1;
  const int g = threadIdx.x / T, lt = threadIdx.x % T, ty = lt / GX, tx = lt % GX;
  const int per = (2 * (GX * CJP + GY * RI) + 16 + 5 * nb + (2 * nb + 1) / 2 + 1) & ~1;
  // LDS of this point's thread grid.  prow / pcol are stored per owner thread ([tx][j], [ty][i]) so that a thread's
  // RI + CJ reads per step are contiguous: LDS bandwidth is shared by every wave of the CU and is what bounds this kernel.
  double* prow = mw_lds + (long)g * per;
  double* pcol = prow + 2 * GX * CJP;  // two buffers each (the unpivoted form alternates them: one barrier per step)
  unsigned long long* cand = reinterpret_cast<unsigned long long*>(pcol + 2 * GY * RI);
  double* pivv = reinterpret_cast<double*>(cand + 16);
  double* bvec = pivv + nb;
  double* csx = bvec + nb;  // coordinates of the selected stations
  double* csy = csx + nb;
  double* csz = csy + nb;
  int* perm = reinterpret_cast<int*>(csz + nb);
  int* sel = perm + nb;
  const long pt = (long)blockIdx.x * PPB + g;
  const bool live = pt < a.npt;
  if (live) {
    for (int r = lt; r < K; r += T) {
      const int st = a.idx[pt * K + r];
      sel[r] = st;
      double y = a.sy[st], z = (a.mode == 3) ? a.sz[st] : 0.0;
      if (a.mode == 1) {
        const double lat = y * MIK_PI / 180.0;
        y = cos(lat);
        z = sin(lat);
      }
      csx[r] = a.sx[st];
      csy[r] = y;
      csz[r] = z;
    }
    for (int r = lt; r < nb; r += T) bvec[r] = (r < K) ? a.dist[pt * K + r] : 1.0;  // dist holds b (k_mw_rhs)
  }
  __syncthreads();
  // PIV = false: the SPD-shifted system (A + s u u^T) x = b + s u, u = [1_K; 0] -- the same x because u.x = sum of the
  // weights = 1 -- whose station block s - gamma is a covariance matrix: eliminated in natural order without a pivot
  // search (quasi-definite, as in the dense path).  s = sill for the bounded models, 4 max gamma(d_i) >= gamma(2 d_K) for
  // linear / power.  A non-positive station pivot raises flag bit 1 and the host reruns the call with PIV = true.
  double shift = 0.0;
  if (!PIV) {
    if (a.v.model >= 2) {
      shift = a.v.p0 + a.v.p2;
    } else {
      double gmax = 0.0;
      for (int r = 0; r < K; ++r) gmax = fmax(gmax, -bvec[r]);
      shift = 4.0 * gmax;
    }
    if (!(shift > 0.0)) shift = 1.0;
  }
  double m[RI][CJ];
  unsigned used = 0;
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int row = ty + GY * i;
    if (row >= nb) used |= 1u << i;  // padding rows never pivot
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      const int col = tx + GX * j;
      double v = 0.0;
      if (live && row < nb && col <= nb) {
        if (col == nb) v = (row < K) ? bvec[row] + shift : bvec[row];
        else if (row < K && col < K)
          v = (row == col) ? shift
              : a.gtab ? -a.gtab[(pt * K + row) * K + col]
                       : shift + mw_entry(a.v, a.mode, csx[row], csy[row], csz[row], csx[col], csy[col], csz[col]);
        else v = (row == K && col == K) ? 0.0 : 1.0;
      }
      m[i][j] = v;
    }
  }
  int bad = 0;
  if (PIV)
  for (int c = 0; c < nb; ++c) {
    const int jj = c / GX, cx = c - jj * GX;  // block-uniform
    if (tx == cx) {
      // pivot candidates of this thread's part of column c: one 64-bit key = |value| (low 8 mantissa bits dropped) with
      // 255 - row in the low byte, so that the maximum key is the largest magnitude and, among equals, the first row
      unsigned long long best = 0ull;
#pragma unroll
      for (int j = 0; j < CJ; ++j)
        if (j == jj) {
#pragma unroll
          for (int i = 0; i < RI; ++i) {
            const unsigned long long key = ((unsigned long long)__double_as_longlong(fabs(m[i][j])) & ~0xFFull) |
                                           (unsigned long long)(255 - (ty + GY * i));
            if (!((used >> i) & 1u) && key > best) best = key;
          }
        }
      cand[ty] = best;
    }
    __syncthreads();
    unsigned long long kb = cand[0];
#pragma unroll
    for (int q = 1; q < GY; ++q) {
      const unsigned long long k2 = cand[q];
      if (k2 > kb) kb = k2;
    }
    if (live && (kb >> 8) == 0ull) bad = 1;
    const int p = 255 - (int)(kb & 0xFFull);
    const int ii = p / GY, py = p - ii * GY;
    if (ty == py) {
#pragma unroll
      for (int i = 0; i < RI; ++i)
        if (i == ii) {
#pragma unroll
          for (int j = 0; j < CJ; ++j) prow[tx * CJP + j] = m[i][j];
        }
    }
    if (tx == cx) {
#pragma unroll
      for (int j = 0; j < CJ; ++j)
        if (j == jj) {
#pragma unroll
          for (int i = 0; i < RI; ++i) pcol[ty * RI + i] = m[i][j];
        }
    }
    if (lt == 0) perm[c] = p;
    __syncthreads();
    double pr[CJ], pc[RI];
#pragma unroll
    for (int j = 0; j < CJ; ++j) pr[j] = prow[tx * CJP + j];
#pragma unroll
    for (int i = 0; i < RI; ++i) pc[i] = pcol[ty * RI + i];
    const double pv = prow[cx * CJP + jj], inv = 1.0 / pv;
    if (lt == 0) pivv[c] = pv;
    double mul[RI];
#pragma unroll
    for (int i = 0; i < RI; ++i) mul[i] = (ty + GY * i == p) ? 0.0 : pc[i] * inv;
    // columns <= c are done: whole tiles j < jj (block-uniform branch per tile), and in tile jj the threads with tx <= cx
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      if (j > jj) {
#pragma unroll
        for (int i = 0; i < RI; ++i) m[i][j] -= mul[i] * pr[j];
      } else if (j == jj) {
        const double prj = (tx > cx) ? pr[j] : 0.0;
#pragma unroll
        for (int i = 0; i < RI; ++i) m[i][j] -= mul[i] * prj;
      }
    }
    if (ty == py) used |= 1u << ii;
  }
  else
  for (int c = 0; c < nb; ++c) {
    const int jj = c / GX, cx = c - jj * GX, ii = c / GY, py = c - ii * GY;  // block-uniform
    double* prb = prow + (c & 1) * GX * CJP;
    double* pcb = pcol + (c & 1) * GY * RI;
    if (ty == py) {
#pragma unroll
      for (int i = 0; i < RI; ++i)
        if (i == ii) {
#pragma unroll
          for (int j = 0; j < CJ; ++j) prb[tx * CJP + j] = m[i][j];
        }
    }
    if (tx == cx) {
#pragma unroll
      for (int j = 0; j < CJ; ++j)
        if (j == jj) {
#pragma unroll
          for (int i = 0; i < RI; ++i) pcb[ty * RI + i] = m[i][j];
        }
    }
    __syncthreads();
    double pr[CJ], pc[RI];
#pragma unroll
    for (int j = 0; j < CJ; ++j) pr[j] = prb[tx * CJP + j];
#pragma unroll
    for (int i = 0; i < RI; ++i) pc[i] = pcb[ty * RI + i];
    const double pv = prb[cx * CJP + jj], inv = 1.0 / pv;
    if (live && !((c < K) ? (pv > 0.0) : (pv < 0.0))) bad = 2;  // not positive definite (or NaN): pivoting needed
    if (lt == 0) pivv[c] = pv;
    double mul[RI];
#pragma unroll
    for (int i = 0; i < RI; ++i) mul[i] = (ty + GY * i == c) ? 0.0 : pc[i] * inv;
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      if (j > jj) {
#pragma unroll
        for (int i = 0; i < RI; ++i) m[i][j] -= mul[i] * pr[j];
      } else if (j == jj) {
        const double prj = (tx > cx) ? pr[j] : 0.0;
#pragma unroll
        for (int i = 0; i < RI; ++i) m[i][j] -= mul[i] * prj;
      }
    }
  }
  __syncthreads();
  {  // solution: the right-hand-side column (col nb) through LDS, indexed by original row
    const int jn = nb / GX, cn = nb - jn * GX;
    if (tx == cn) {
#pragma unroll
      for (int j = 0; j < CJ; ++j)
        if (j == jn) {
#pragma unroll
          for (int i = 0; i < RI; ++i) pcol[ty * RI + i] = m[i][j];
        }
    }
  }
  __syncthreads();
  double zz = 0.0, s2 = 0.0;
  if (live)
    for (int c = lt; c < nb; c += T) {
      const int p = PIV ? perm[c] : c;
      const double x = pcol[(p % GY) * RI + p / GY] / pivv[c];
      if (c < K) zz += x * a.Z[sel[c]];
      s2 += x * bvec[c];
    }
#pragma unroll
  for (int o = W / 2; o > 0; o >>= 1) {
    zz += __shfl_xor(zz, o, W);
    s2 += __shfl_xor(s2, o, W);
  }
  if (T > 64) {
    __syncthreads();
    if ((lt & 63) == 0) { pivv[lt >> 6] = zz; prow[lt >> 6] = s2; }
    __syncthreads();
    zz = pivv[0];
    s2 = prow[0];
#pragma unroll
    for (int w = 1; w < NW; ++w) {
      zz += pivv[w];
      s2 += prow[w];
    }
  }
  if (live && lt == 0) {
    a.z[pt] = zz;
    a.ss[pt] = -s2;
    if (bad) atomicOr(a.flag, bad);
  }
}

}  // namespace mik
