// mik_k_mw.h -- moving-window kriging (n_closest_points)
// (one of the section headers mik_kernels.h is the umbrella of; every section is included by exactly one translation unit of the library)
#pragma once
#include "mik_dev.h"
#include "mik_k_mw_chol.h"

namespace mik {

// ------------------------------------------------------------------------------------------------
// Moving-window kriging (n_closest_points; ok.py:929-986, 722-758, cok.pyx:98-193, ok3d.py:697-733).
//   k_mw_knn   : the k nearest stations of every point, ascending distance (cKDTree.query(k=..., eps=0)),
//                brute force, one wavefront per point (threshold filter + LDS bitonic cuts, see below).
//   k_mw_rhs   : right-hand sides -gamma(bd) with the eps rule, in place over the distances.
//   k_mw_solve : per point the (k+1) x (k+1) system (a_all[sel][:, sel] computed from the selected stations'
//                coordinates, ones border, zero corner -- cok.pyx:138-147), solved by Gauss-Jordan
//                elimination with partial pivoting (dgesv's pivot choice) in the registers of a G x G thread
//                grid; z = x.Z[sel], ss = -x.b.
// ------------------------------------------------------------------------------------------------
#define MIK_MW_KMAX 127

// One wavefront per point over a uniform grid of station cells (stations sorted by cell on the host, cstart[] = first
// sorted position of every cell).  Rings of cells around the point's cell are visited outwards; a ring row is one
// contiguous range of sorted stations.  The 64 lanes take 64 stations at a time; squared distances not above the current
// K-th best (tau) are appended to an LDS candidate buffer by ballot + prefix count; the buffer is bitonic-sorted in LDS
// and cut back to the best K when it is about to overflow and at the end of every ring that has >= K candidates, which
// tightens tau.  Any station outside rings 0..r is at least r * cell away, so the search stops as soon as
// tau <= (r * cell)^2: the work per point follows K, not N.  A 1-cell grid is the plain brute-force scan.
// Ties are broken by station index (what a scan in index order would keep).
// First pass with a BOUND (round 3): a cell holds ~max(8, K) stations, so a disc of radius sqrt(tau0) < cell around the
// point is expected to hold K + 4 sqrt(K) + 2 of them, all inside rings 0 and 1.  Only those become candidates: one scan of
// the 3 x 3 cells and ONE sort of ~1.5 K entries instead of a cut-back sort for every ~2 K candidates (the sorts were 80 % of
// the search).  If fewer than K stations lie within the bound (sparse corner, point far outside the stations) the walk starts
// again without it.
// CAP (a power of two, >= K + 256) candidates: keys[CAP] doubles then vals[CAP] ints of dynamic LDS.
struct KnnArgs {
  const double *px, *py, *pz;  // points (this chunk)
  int npt;
  const double *gx, *gy, *gz;  // stations sorted by cell
  const int* orig;             // sorted position -> station index
  const int* cstart;           // ncell + 1
  int N, K, CAP;
  int nx, ny, nz;
  double x0, y0, z0, inv_cell, cell2;  // grid origin, 1 / cell edge, cell edge squared
  double tau0;                         // first-pass bound on the squared distance (<= cell2), 0 = none: see k_mw_knn
  int* idx_out;
  double* dist_out;
  // round 4: k_mw_knn_lane leaves the points it could not finish in todo[0 .. *todo_count); k_mw_knn then walks that list instead
  // of all points (todo == nullptr: all points)
  int* todo;
  int* todo_count;
};

// Neighbour search for SMALL windows (K <= KMAX <= 32) over points that arrive in spatial order (the rows of a grid): one LANE
// per point.  The 64 consecutive points of a wavefront share the box of station cells that covers all their 3 x 3 (x 3)
// neighbourhoods; its stations are staged through LDS 64 at a time (one coalesced load per batch) and every lane keeps its KMAX
// nearest -- ascending by (squared distance, station index), the order k_mw_knn and cKDTree.query produce -- in registers by
// sorted insertion with compile-time indices: no candidate buffer, no bitonic sort, one pass (the wave-per-point kernel spends
// ~80 % of its time sorting ~1.5 K candidates per point).  A lane is done when its K-th distance is within its distance to the
// box's nearest open side (no station outside the box can be closer); lanes that are not -- sparse corners, points far outside
// the stations, waves whose points are scattered (a shuffled point list) -- are appended to `todo` and finished by k_mw_knn.
// Reference: cKDTree.query(k) of ok.py:957-960 / ok3d.py:904-908.
template <int NDIM, int KMAX>
__global__ void __launch_bounds__(64) k_mw_knn_lane(KnnArgs a) {
  __shared__ double sx[64], sy[64], sz[64];
  __shared__ int sid[64];
  const int l = threadIdx.x, K = a.K;
  constexpr int MAXCELLS = (NDIM == 3) ? 125 : 40;
  auto wmin = [](int v) {
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
  };
  auto wmax = [](int v) {
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
  };
  for (long base = (long)blockIdx.x * 64; base < a.npt; base += (long)gridDim.x * 64) {
    const long t = base + l;
    const bool ok = t < a.npt;
    const long ts = ok ? t : base;  // padding lanes shadow the wave's first point (they write nothing)
    const double qx = a.px[ts], qy = a.py[ts], qz = (NDIM == 3) ? a.pz[ts] : 0.0;
    const int cx = min(a.nx - 1, max(0, (int)floor((qx - a.x0) * a.inv_cell)));
    const int cy = min(a.ny - 1, max(0, (int)floor((qy - a.y0) * a.inv_cell)));
    const int cz = (NDIM == 3) ? min(a.nz - 1, max(0, (int)floor((qz - a.z0) * a.inv_cell))) : 0;
    const int xa = max(0, wmin(cx) - 1), xb = min(a.nx - 1, wmax(cx) + 1);
    const int ya = max(0, wmin(cy) - 1), yb = min(a.ny - 1, wmax(cy) + 1);
    const int za = (NDIM == 3) ? max(0, wmin(cz) - 1) : 0, zb = (NDIM == 3) ? min(a.nz - 1, wmax(cz) + 1) : 0;
    const long cells = (long)(xb - xa + 1) * (yb - ya + 1) * (zb - za + 1);
    bool done = false;
    double key[KMAX];
    int id[KMAX];
    if (cells <= MAXCELLS) {  // wave-uniform
#pragma unroll
      for (int q = 0; q < KMAX; ++q) {
        key[q] = 1e300;
        id[q] = 0x7fffffff;
      }
      for (int z = za; z <= zb; ++z)
        for (int y = ya; y <= yb; ++y) {
          const long row = ((long)z * a.ny + y) * a.nx;
          const int beg = a.cstart[row + xa], end = a.cstart[row + xb + 1];
          for (int j0 = beg; j0 < end; j0 += 64) {
            const int j = j0 + l;
            if (j < end) {
              sx[l] = a.gx[j];
              sy[l] = a.gy[j];
              if (NDIM == 3) sz[l] = a.gz[j];
              sid[l] = a.orig[j];
            }
            __syncthreads();
            const int n = min(64, end - j0);
            for (int s = 0; s < n; ++s) {
              const double dx = qx - sx[s], dy = qy - sy[s];
              double d2 = dx * dx + dy * dy;
              if (NDIM == 3) {
                const double dz = qz - sz[s];
                d2 += dz * dz;
              }
              const int st = sid[s];
              if (d2 < key[KMAX - 1] || (d2 == key[KMAX - 1] && st < id[KMAX - 1])) {
                bool placed = false;
#pragma unroll
                for (int q = KMAX - 1; q > 0; --q) {
                  if (!placed) {
                    const bool sh = d2 < key[q - 1] || (d2 == key[q - 1] && st < id[q - 1]);
                    key[q] = sh ? key[q - 1] : d2;
                    id[q] = sh ? id[q - 1] : st;
                    placed = !sh;
                  }
                }
                if (!placed) {
                  key[0] = d2;
                  id[0] = st;
                }
              }
            }
            __syncthreads();
          }
        }
      // the K-th nearest so far (K - 1 is not a compile-time index)
      double tau = 1e300;
#pragma unroll
      for (int q = 0; q < KMAX; ++q)
        if (q == K - 1) tau = key[q];
      // distance to the nearest OPEN side of the box (a side at the edge of the grid is closed: no station lies beyond it)
      const double cell = 1.0 / a.inv_cell;
      double reach = 1e300;
      if (xa > 0) reach = fmin(reach, qx - (a.x0 + xa * cell));
      if (xb < a.nx - 1) reach = fmin(reach, (a.x0 + (xb + 1) * cell) - qx);
      if (ya > 0) reach = fmin(reach, qy - (a.y0 + ya * cell));
      if (yb < a.ny - 1) reach = fmin(reach, (a.y0 + (yb + 1) * cell) - qy);
      if (NDIM == 3) {
        if (za > 0) reach = fmin(reach, qz - (a.z0 + za * cell));
        if (zb < a.nz - 1) reach = fmin(reach, (a.z0 + (zb + 1) * cell) - qz);
      }
      // (the cell edges are recomputed here with a different rounding than the binning used: keep a relative margin)
      done = tau < 1e300 && reach > 0.0 && tau <= reach * reach * (1.0 - 1e-9);
    }
    if (ok && done) {
#pragma unroll
      for (int q = 0; q < KMAX; ++q)
        if (q < K) {
          a.idx_out[t * K + q] = id[q];
          a.dist_out[t * K + q] = sqrt(key[q]);
        }
    }
    const bool later = ok && !done;
    const unsigned long long m = __ballot(later);
    if (m) {
      int pos = 0;
      if (l == 0) pos = atomicAdd(a.todo_count, __popcll(m));
      pos = __shfl(pos, 0);
      if (later) a.todo[pos + __popcll(m & ((1ULL << l) - 1ULL))] = (int)t;
    }
  }
}

template <int NDIM>
__global__ void __launch_bounds__(64) k_mw_knn(KnnArgs a) {
  extern __shared__ double knn_lds[];
  const int K = a.K, CAP = a.CAP;
  double* keys = knn_lds;
  int* vals = reinterpret_cast<int*>(keys + CAP);
  const int l = threadIdx.x;
  const unsigned long long below = (l == 0) ? 0ull : (~0ull >> (64 - l));
  // cut back to the best K as soon as ~2K candidates are in (an early, small sort tightens tau for the rest of the scan),
  // at the latest when the next trip's 256 stations might not fit
  const int cut_at = min(CAP - 256, max(2 * K, 192));
  const long nwork = a.todo ? (long)*a.todo_count : (long)a.npt;
  for (long w = blockIdx.x; w < nwork; w += gridDim.x) {
    const long t = a.todo ? (long)a.todo[w] : w;
    const double qx = a.px[t], qy = a.py[t], qz = (NDIM == 3) ? a.pz[t] : 0.0;
    int cnt = 0;
    double tau = a.tau0 > 0.0 ? a.tau0 : 1e300;
    bool bounded = a.tau0 > 0.0;
    // sort the first S = pow2 >= cnt entries ascending by (distance, station index), keep the best K
    auto cut = [&]() {
      int S = 64;
      while (S < cnt) S <<= 1;
      for (int i = cnt + l; i < S; i += 64) {
        keys[i] = 1e300;
        vals[i] = 0x7fffffff;
      }
      __syncthreads();
      for (int k = 2; k <= S; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int i = l; i < (S >> 1); i += 64) {
            const int lo = ((i & ~(j - 1)) << 1) | (i & (j - 1)), hi = lo | j;
            const double ka = keys[lo], kb = keys[hi];
            const int va = vals[lo], vb = vals[hi];
            const bool gt = (ka > kb) || (ka == kb && va > vb);
            if (gt == ((lo & k) == 0)) {
              keys[lo] = kb;
              keys[hi] = ka;
              vals[lo] = vb;
              vals[hi] = va;
            }
          }
          __syncthreads();
        }
      if (cnt > K) cnt = K;
      if (cnt == K) tau = keys[K - 1];
    };
    // candidates from the sorted stations [beg, end)
    auto scan = [&](int beg, int end) {
      for (int j0 = beg; j0 < end; j0 += 256) {
        // four batches of 64 stations per trip: their coordinate loads are issued together
        double d2[4];
        int id[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = j0 + u * 64 + l;
          d2[u] = 1e300;
          id[u] = 0;
          if (j < end) {
            const double dx = qx - a.gx[j], dy = qy - a.gy[j];
            d2[u] = dx * dx + dy * dy;
            if (NDIM == 3) {
              const double dz = qz - a.gz[j];
              d2[u] += dz * dz;
            }
            id[u] = a.orig[j];
          }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const bool take = (j0 + u * 64 + l < end) && (d2[u] <= tau);
          const unsigned long long m = __ballot(take);
          if (take) {
            const int pos = cnt + __popcll(m & below);
            keys[pos] = d2[u];
            vals[pos] = id[u];
          }
          cnt += __popcll(m);
          if (j0 + (u + 1) * 64 >= end) break;  // wave-uniform
        }
        if (cnt > cut_at) cut();
      }
    };
    const int cx = min(a.nx - 1, max(0, (int)floor((qx - a.x0) * a.inv_cell)));
    const int cy = min(a.ny - 1, max(0, (int)floor((qy - a.y0) * a.inv_cell)));
    const int cz = (NDIM == 3) ? min(a.nz - 1, max(0, (int)floor((qz - a.z0) * a.inv_cell))) : 0;
    for (int r = 0;; ++r) {
      const int zr = (NDIM == 3) ? r : 0;
      for (int dz = -zr; dz <= zr; ++dz) {
        const int z = cz + dz;
        if (z < 0 || z >= a.nz) continue;
        for (int dy = -r; dy <= r; ++dy) {
          const int y = cy + dy;
          if (y < 0 || y >= a.ny) continue;
          const long row = ((long)z * a.ny + y) * a.nx;
          const bool shell = (dy == -r || dy == r || (NDIM == 3 && (dz == -r || dz == r)));
          if (shell) {  // the whole row of the block is new
            const int xa = max(0, cx - r), xb = min(a.nx - 1, cx + r);
            scan(a.cstart[row + xa], a.cstart[row + xb + 1]);
          } else {      // only its two end cells are
            if (cx - r >= 0) scan(a.cstart[row + cx - r], a.cstart[row + cx - r + 1]);
            if (cx + r < a.nx) scan(a.cstart[row + cx + r], a.cstart[row + cx + r + 1]);
          }
        }
      }
      const bool all = cx - r <= 0 && cx + r >= a.nx - 1 && cy - r <= 0 && cy + r >= a.ny - 1 &&
                       (NDIM != 3 || (cz - r <= 0 && cz + r >= a.nz - 1));
      if (bounded && cnt < K && (r >= 1 || all)) {  // the bound was too tight here: again, without it
        bounded = false;
        cnt = 0;
        tau = 1e300;
        r = -1;
        continue;
      }
      if (cnt >= K || all) cut();
      const double reach = (double)r * (double)r * a.cell2;
      if (all || (cnt == K && tau <= reach)) break;
    }
    for (int q = l; q < K; q += 64) {
      const int st = vals[q];
      a.idx_out[t * K + q] = (st >= 0 && st < a.N) ? st : 0;  // fewer than K finite distances (NaN coordinates): stay in bounds
      a.dist_out[t * K + q] = sqrt(keys[q]);
    }
    __syncthreads();  // the buffer is reused by the next point
  }
}

// custom variogram, moving window: distances between the selected stations of every point, [point][row][col]
__global__ void __launch_bounds__(256)
k_mw_pairdist(const int* __restrict__ idx, long npt, int K, const double* __restrict__ sx, const double* __restrict__ sy,
              const double* __restrict__ sz, int mode, double* __restrict__ out) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= npt * K * K) return;
  const long pt = e / ((long)K * K);
  const int rc = (int)(e - pt * K * K), r = rc / K, c = rc - r * K;
  const int s1 = idx[pt * K + r], s2 = idx[pt * K + c];
  double d = 0.0;
  if (r != c) {
    if (mode == 1) {
      const double la1 = sy[s1] * MIK_PI / 180.0, la2 = sy[s2] * MIK_PI / 180.0;
      d = gc_dist(sx[s1], cos(la1), sin(la1), sx[s2], cos(la2), sin(la2));
    } else {
      const double dx = sx[s1] - sx[s2], dy = sy[s1] - sy[s2], dz = (mode == 3) ? sz[s1] - sz[s2] : 0.0;
      d = sqrt(dx * dx + dy * dy + dz * dz);
    }
  }
  out[e] = d;
}
// custom variogram, moving window: b = -gamma (host-mapped copy of the distances), 0 on an exact hit
__global__ void __launch_bounds__(256)
k_mw_rhs_table(double* __restrict__ dist, const double* __restrict__ gam, long n, int exact, double eps) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const double d = dist[e];
  dist[e] = (exact && d <= eps) ? 0.0 : -gam[e];
}

// right-hand sides in place: dist[e] (distance to the e-th selected station) -> b = -gamma(d), 0 on an exact hit
// (cok.pyx:150-158 with check_b_vect, cok.pyx:196-203)
template <int MODEL>
__global__ void __launch_bounds__(256) k_mw_rhs(double* __restrict__ dist, long n, Vario v, int exact, double eps) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const double d = dist[e];
  double b = -vario<MODEL, false>(v, d, d * d);
  if (exact && d <= eps) b = 0.0;
  dist[e] = b;
}

// (k_mw_solve -- the Gauss-Jordan solver with partial pivoting in registers -- lives in mik_k_mw_solve.h: a translation unit of its own, round 6)

// ---- windows beyond the register classes (K > MIK_MW_CHOL_KMAX): BLOCKED Cholesky of the SPD-shifted local system ----------
// One 256-thread block per point (grid-strided over the chunk); the (ldc + 64) x ldc system -- lower triangle of
// C = s 11^T - Gamma padded with identity to ldc = 64 ceil(K / 64), and the three right-hand sides {b + s, 1, Z} as rows
// ldc..ldc+2 -- sits in a per-block scratch slot (2.4 MB at K = 512: L2 / Infinity-Cache resident).  64-wide panels:
//   (a) the diagonal block is factored in LDS (64 steps, 256 threads);
//   (b) every row below it is solved against it by ONE thread (forward substitution, the 64 entries in registers, broadcast
//       LDS reads of the factor) -- the right-hand-side rows included: their forward substitution is this step;
//   (c) the trailing matrix is updated in 64 x 64 tiles, both operand panels staged k-major in LDS, a 4 x 4 micro-tile per
//       thread (two ds_read_b128 per operand and k).
// z and sigma^2 are the inner products of the three solved rows, as in k_mw_chol (C = L L^T here, so no D^-1).  A
// non-positive pivot raises flag bit 1 and the call is redone by the pivoted kernel (k_mw_solve_big).  Replaces the unblocked
// HBM elimination for named variogram models: k = 512 went from 3.9 k to > 100 k points/s (profiles/r03_moving_window_timing.txt).
// Reference: lib/cok.pyx:98-193 (one dgesv per point), ok.py:929-986.
#define MIK_MWP 64
#define MIK_MWP_LD 66  // LDS row stride of the operand panels (even: the 4-element fragment reads are 16-byte aligned)
__global__ void __launch_bounds__(256, 2) k_mw_chol_blocked(MwArgs a, double* __restrict__ scratch, long slot, int ldc) {
  extern __shared__ double mwc_lds[];
  double* LR = mwc_lds;                          // [64][66]: diagonal block (row-major) / row-block operand, k-major
  double* LS = mwc_lds + MIK_MWP * MIK_MWP_LD;   // [64][66]: column-block operand, k-major
  __shared__ double red[5][4];
  __shared__ double rdiag[MIK_MWP];              // 1 / L_jj of the diagonal block being used
  __shared__ double sh_shift;
  __shared__ int sh_bad;
  const int K = a.K, l = threadIdx.x, lane = l & 63, wave = l >> 6;
  const int nP = ldc / MIK_MWP;
  double* A = scratch + (long)blockIdx.x * slot;       // (ldc + 64) x ldc
  double* cs = A + (long)(ldc + MIK_MWP) * ldc;        // coordinates of the selected stations: x | y | z, K each
  const double* bv = nullptr;
#ifdef MIK_MW_PROFILE
  long long tph[6] = {0, 0, 0, 0, 0, 0}, tlast = 0;
#define MWP_TICK(i) do { __syncthreads(); const long long now_ = wall_clock64(); tph[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define MWP_TICK(i) do { } while (0)
#endif
  for (long pt = blockIdx.x; pt < a.npt; pt += gridDim.x) {
    __syncthreads();
#ifdef MIK_MW_PROFILE
    tlast = wall_clock64();
#endif
    if (l == 0) sh_bad = 0;
    bv = a.dist + pt * K;  // b = -gamma(d), 0 on an exact hit (k_mw_rhs)
    double gmax = 0.0;
    for (int r = l; r < K; r += 256) {
      const int st = a.idx[pt * K + r];
      double x = a.sx[st], y = a.sy[st], z = (a.mode == 3) ? a.sz[st] : 0.0;
      if (a.mode == 1) {
        const double lat = y * MIK_PI / 180.0;
        y = cos(lat);
        z = sin(lat);
      }
      cs[r] = x, cs[K + r] = y, cs[2 * K + r] = z;
      gmax = fmax(gmax, -bv[r]);
      // right-hand-side rows (columns < K; the padding columns stay 0)
      A[(long)(ldc + 2) * ldc + r] = a.Z[st];
    }
    if (a.v.model < 2) {  // no sill: shift by four times the largest gamma of the window (>= gamma(2 d_k))
      for (int o = 32; o > 0; o >>= 1) gmax = fmax(gmax, __shfl_xor(gmax, o));
      if (lane == 0) red[0][wave] = gmax;
    }
    __syncthreads();
    if (l == 0) {
      double s = a.v.p0 + a.v.p2;
      if (a.v.model < 2) s = 4.0 * fmax(fmax(red[0][0], red[0][1]), fmax(red[0][2], red[0][3]));
      if (!(s > 0.0)) s = 1.0;
      sh_shift = s;
    }
    __syncthreads();
    const double shift = sh_shift;
    for (int r = l; r < ldc; r += 256) {
      A[(long)ldc * ldc + r] = r < K ? bv[r] + shift : 0.0;
      A[(long)(ldc + 1) * ldc + r] = r < K ? 1.0 : 0.0;
      if (r >= K) A[(long)(ldc + 2) * ldc + r] = 0.0;
    }
    // lower triangle of the shifted matrix, identity in the padding
    for (int r = wave; r < ldc; r += 4) {
      double* row = A + (long)r * ldc;
      if (r < K) {
        const double xr = cs[r], yr = cs[K + r], zr = cs[2 * K + r];
        for (int c = lane; c <= r; c += 64)
          row[c] = (c == r) ? shift : shift + mw_entry(a.v, a.mode, xr, yr, zr, cs[c], cs[K + c], cs[2 * K + c]);
      } else {
        for (int c = lane; c <= r; c += 64) row[c] = (c == r) ? 1.0 : 0.0;
      }
    }
    __syncthreads();
    MWP_TICK(0);  // set-up: stations, right-hand sides, matrix fill
    for (int p = 0; p < nP; ++p) {
      const int c0 = p * MIK_MWP;
      // (a) diagonal block -> LDS, row-major, lower part; Cholesky in place
      for (int e = l; e < MIK_MWP * MIK_MWP; e += 256) {
        const int i = e >> 6, k = e & 63;
        LR[i * MIK_MWP_LD + k] = (k <= i) ? A[(long)(c0 + i) * ldc + c0 + k] : 0.0;
      }
      __syncthreads();
      // right-looking elimination WITHOUT scaling the pivot column first: (i, k) -= a_ij a_kj / d_j uses the raw column j, which
      // no later step touches -- one barrier per step instead of two, and no square root or division in the loop (round 3: this
      // loop was more than half of the kernel at K = 257 .. 512); the columns are scaled to the Cholesky factor in one pass after it
      for (int j = 0; j < MIK_MWP; ++j) {
        const double d = LR[j * MIK_MWP_LD + j];  // (the barrier at the end of the previous step ordered its updates before this)
        const double inv = pivot_recip(d > 0.0 ? d : 1.0);
        if (l == 0) {
          if (!(d > 0.0)) sh_bad = 1;
          rdiag[j] = inv;  // 1 / d_j for now
        }
        {  // trailing part of the block: (i, k), j < k <= i < 64
          const int i = l & 63;
          const double aij = LR[i * MIK_MWP_LD + j] * inv;
          for (int k = j + 1 + (l >> 6); k <= i; k += 4) LR[i * MIK_MWP_LD + k] -= aij * LR[k * MIK_MWP_LD + j];
        }
        __syncthreads();
      }
      {  // L_ij = a_ij / sqrt(d_j) (i > j), L_jj = sqrt(d_j), rdiag[j] = 1 / L_jj
        const int j = l & 63;
        const double rs = sqrt(rdiag[j]);
        __syncthreads();  // everyone has read 1 / d_j
        for (int i = j + (l >> 6); i < MIK_MWP; i += 4) LR[i * MIK_MWP_LD + j] *= rs;  // (the diagonal: d_j / sqrt(d_j))
        if (l < MIK_MWP) rdiag[j] = rs;
      }
      __syncthreads();
      for (int e = l; e < MIK_MWP * MIK_MWP; e += 256) {  // the factored block goes back (lower part)
        const int i = e >> 6, k = e & 63;
        if (k <= i) A[(long)(c0 + i) * ldc + c0 + k] = LR[i * MIK_MWP_LD + k];
      }
      MWP_TICK(1);  // (a) diagonal block
      // (b) the rows below: x L^T = a  ->  x_j = (a_j - sum_{k<j} x_k L_jk) / L_jj, one row per thread, 16 entries at a time:
      // the solved part of the row is read back from the scratch slot (a fully unrolled 64-entry register version spilled)
      // (rows K..ldc-1 are identity padding: zero in this panel, nothing to solve; the thread index runs over the real rows)
      for (int rr = l; rr < (K - c0 - MIK_MWP > 0 ? K - c0 - MIK_MWP : 0) + 3; rr += 256) {
        const int nreal = K - c0 - MIK_MWP > 0 ? K - c0 - MIK_MWP : 0;
        const int r = rr < nreal ? c0 + MIK_MWP + rr : ldc + (rr - nreal);
        double* row = A + (long)r * ldc + c0;
        for (int sb4 = 0; sb4 < 4; ++sb4) {
          double x[16];
#pragma unroll
          for (int k = 0; k < 16; k += 2) {
            const double2 v = *reinterpret_cast<const double2*>(row + 16 * sb4 + k);
            x[k] = v.x, x[k + 1] = v.y;
          }
          for (int q = 0; q < sb4; ++q) {
            double xq[16];
#pragma unroll
            for (int k = 0; k < 16; k += 2) {
              const double2 v = *reinterpret_cast<const double2*>(row + 16 * q + k);
              xq[k] = v.x, xq[k + 1] = v.y;
            }
            const double* Lb = LR + (16 * sb4) * MIK_MWP_LD + 16 * q;
#pragma unroll
            for (int j = 0; j < 16; ++j)
#pragma unroll
              for (int k = 0; k < 16; ++k) x[j] -= xq[k] * Lb[j * MIK_MWP_LD + k];
          }
          const double* Ld = LR + (16 * sb4) * MIK_MWP_LD + 16 * sb4;
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            double sacc = x[j];
#pragma unroll
            for (int k = 0; k < j; ++k) sacc -= x[k] * Ld[j * MIK_MWP_LD + k];
            x[j] = sacc * rdiag[16 * sb4 + j];
          }
#pragma unroll
          for (int k = 0; k < 16; k += 2) *reinterpret_cast<double2*>(row + 16 * sb4 + k) = make_double2(x[k], x[k + 1]);
        }
      }
      __syncthreads();
      MWP_TICK(2);  // (b) panel solve
      // (c) trailing update, tiles (rb, sb) of 64 x 64 with sb <= rb; the right-hand sides are the 3-row block after the matrix
      const int nb_rows = nP - p - 1;  // matrix row blocks below the panel
      for (int rb = 0; rb <= nb_rows; ++rb) {
        const bool rhs_blk = rb == nb_rows;
        const int r0 = c0 + MIK_MWP + rb * MIK_MWP;  // == ldc for the right-hand-side block
        if (rhs_blk && nb_rows == 0) break;           // last panel: nothing to the right of it
        if (!rhs_blk && r0 >= K) continue;            // a row block of identity padding
        // row-block operand, k-major: LR[k][row]
        for (int e = l; e < MIK_MWP * MIK_MWP; e += 256) {
          const int i = e >> 6, k = e & 63;
          LR[k * MIK_MWP_LD + i] = (!rhs_blk || i < 3) ? A[(long)(r0 + i) * ldc + c0 + k] : 0.0;
        }
        const int sb_end = rhs_blk ? nb_rows - 1 : rb;
        for (int sb = 0; sb <= sb_end; ++sb) {
          const int s0 = c0 + MIK_MWP + sb * MIK_MWP;
          if (s0 >= K) break;  // column blocks of identity padding (block-uniform)
          __syncthreads();  // LR is staged / the previous tile is done with LS
          if (!rhs_blk && sb == rb) {
            for (int e = l; e < MIK_MWP * MIK_MWP_LD; e += 256) LS[e] = LR[e];
          } else {
            for (int e = l; e < MIK_MWP * MIK_MWP; e += 256) {
              const int i = e >> 6, k = e & 63;
              LS[k * MIK_MWP_LD + i] = A[(long)(s0 + i) * ldc + c0 + k];
            }
          }
          __syncthreads();
          const int ty = l >> 4, tx = l & 15;
          if (!rhs_blk || ty == 0) {
            double acc[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 4; ++j) acc[i][j] = 0.0;
#pragma unroll 8
            for (int k = 0; k < MIK_MWP; ++k) {
              double av[4], bw[4];
#pragma unroll
              for (int i = 0; i < 4; ++i) av[i] = LR[k * MIK_MWP_LD + 4 * ty + i], bw[i] = LS[k * MIK_MWP_LD + 4 * tx + i];
#pragma unroll
              for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bw[j];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              if (rhs_blk && i == 3) break;
              double* out = A + (long)(r0 + 4 * ty + i) * ldc + s0 + 4 * tx;
              double2 v0 = *reinterpret_cast<double2*>(out), v1 = *reinterpret_cast<double2*>(out + 2);
              v0.x -= acc[i][0], v0.y -= acc[i][1], v1.x -= acc[i][2], v1.y -= acc[i][3];
              *reinterpret_cast<double2*>(out) = v0;
              *reinterpret_cast<double2*>(out + 2) = v1;
            }
          }
        }
        __syncthreads();  // the tiles of this row block are done with LR
      }
      __syncthreads();
      MWP_TICK(3);  // (c) trailing update
    }
    MWP_TICK(4);
    // the three solved rows y_q = L^-1 rhs_q; G_pq = y_p . y_q
    double g00 = 0.0, g01 = 0.0, g11 = 0.0, g02 = 0.0, g12 = 0.0;
    for (int c = l; c < K; c += 256) {
      const double y0 = A[(long)ldc * ldc + c], y1 = A[(long)(ldc + 1) * ldc + c], y2 = A[(long)(ldc + 2) * ldc + c];
      g00 += y0 * y0, g01 += y0 * y1, g11 += y1 * y1, g02 += y0 * y2, g12 += y1 * y2;
    }
    for (int o = 32; o > 0; o >>= 1) {
      g00 += __shfl_xor(g00, o), g01 += __shfl_xor(g01, o), g11 += __shfl_xor(g11, o), g02 += __shfl_xor(g02, o), g12 += __shfl_xor(g12, o);
    }
    if (lane == 0) red[0][wave] = g00, red[1][wave] = g01, red[2][wave] = g11, red[3][wave] = g02, red[4][wave] = g12;
    __syncthreads();
    if (l == 0) {
      g00 = red[0][0] + red[0][1] + red[0][2] + red[0][3];
      g01 = red[1][0] + red[1][1] + red[1][2] + red[1][3];
      g11 = red[2][0] + red[2][1] + red[2][2] + red[2][3];
      g02 = red[3][0] + red[3][1] + red[3][2] + red[3][3];
      g12 = red[4][0] + red[4][1] + red[4][2] + red[4][3];
      const double mu = (g01 - 1.0) / g11;
      a.z[pt] = g02 - mu * g12;
      a.ss[pt] = -(g00 - mu * g01) + shift - mu;
      if (sh_bad || !(g11 > 0.0)) atomicOr(a.flag, 2);
    }
    MWP_TICK(5);
  }
#ifdef MIK_MW_PROFILE
  if (blockIdx.x == 0 && l == 0)
    printf("[k_mw_chol_blocked K=%d] per block, 100 MHz ticks: set-up %lld | diagonal %lld | panel solve %lld | trailing update %lld | (gap) %lld | reduction %lld\n",
           K, tph[0], tph[1], tph[2], tph[3], tph[4], tph[5]);
#endif
}

// ---- n_closest_points > MIK_MW_KMAX: the same two steps with their working sets in HBM instead of registers / LDS ----
// k_mw_knn_big : one thread per point; its ascending candidate list lives in a [rank][point] work array (neighbouring
//                threads touch neighbouring addresses while they are at the same rank) and is copied to the usual
//                [point][rank] layout at the end.
// k_mw_solve_big: one 256-thread block per point (grid-strided over the chunk); the augmented (k+1) x (k+2) system sits in
//                a per-block HBM/L2 scratch slot; LU forward elimination with partial pivoting (dgesv's pivot order,
//                cok.pyx:165) + column-oriented back substitution.
template <int NDIM>
__global__ void __launch_bounds__(256)
k_mw_knn_big(const double* __restrict__ px, const double* __restrict__ py, const double* __restrict__ pz, int npt,
             const double* __restrict__ xs, const double* __restrict__ ys, const double* __restrict__ zs, int N, int K,
             double* __restrict__ wd, int* __restrict__ wi, int* __restrict__ idx_out, double* __restrict__ dist_out) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= npt) return;
  const long P = npt;
  const double qx = px[t], qy = py[t], qz = (NDIM == 3) ? pz[t] : 0.0;
  int cnt = 0;
  double worst = 1e300;
  for (int j = 0; j < N; ++j) {
    const double dx = qx - xs[j], dy = qy - ys[j];
    double d2 = dx * dx + dy * dy;
    if (NDIM == 3) {
      const double dz = qz - zs[j];
      d2 += dz * dz;
    }
    if (cnt < K || d2 < worst) {
      int p = (cnt < K) ? cnt : K - 1;
      while (p > 0) {
        const double prev = wd[(long)(p - 1) * P + t];
        if (!(prev > d2)) break;
        wd[(long)p * P + t] = prev;
        wi[(long)p * P + t] = wi[(long)(p - 1) * P + t];
        --p;
      }
      wd[(long)p * P + t] = d2;
      wi[(long)p * P + t] = j;
      if (cnt < K) ++cnt;
      if (cnt == K) worst = wd[(long)(K - 1) * P + t];
    }
  }
  for (int q = 0; q < K; ++q) {
    idx_out[(long)t * K + q] = wi[(long)q * P + t];
    dist_out[(long)t * K + q] = sqrt(wd[(long)q * P + t]);
  }
}

__global__ void __launch_bounds__(256) k_mw_solve_big(MwArgs a, double* __restrict__ scratch) {
  extern __shared__ double mwb_lds[];  // mul[nb] | x[nb] | sel[nb] (ints)
  const int K = a.K, nb = K + 1, st = nb + 1, l = threadIdx.x;
  double* mul = mwb_lds;
  double* xv = mul + nb;
  int* sel = reinterpret_cast<int*>(xv + nb);
  double* aug = scratch + (long)blockIdx.x * nb * st;
  __shared__ double redv[4];
  __shared__ int redr[4];
  int bad = 0;
  for (long pt = blockIdx.x; pt < a.npt; pt += gridDim.x) {
    __syncthreads();
    for (int r = l; r < K; r += 256) sel[r] = a.idx[pt * K + r];
    __syncthreads();
    for (long e = l; e < (long)nb * nb; e += 256) {
      const int r = (int)(e / nb), c = (int)(e - (long)r * nb);
      double v;
      if (r < K && c < K) {
        v = 0.0;
        if (r != c && a.gtab) {
          v = -a.gtab[(pt * K + r) * K + c];
        } else if (r != c) {
          const int s1 = sel[r], s2 = sel[c];
          double y1 = a.sy[s1], y2 = a.sy[s2], z1 = (a.mode == 3) ? a.sz[s1] : 0.0, z2 = (a.mode == 3) ? a.sz[s2] : 0.0;
          if (a.mode == 1) {
            const double la1 = y1 * MIK_PI / 180.0, la2 = y2 * MIK_PI / 180.0;
            y1 = cos(la1), z1 = sin(la1), y2 = cos(la2), z2 = sin(la2);
          }
          v = mw_entry(a.v, a.mode, a.sx[s1], y1, z1, a.sx[s2], y2, z2);
        }
      } else {
        v = (r == K && c == K) ? 0.0 : 1.0;
      }
      aug[(long)r * st + c] = v;
    }
    for (int r = l; r < nb; r += 256) {
      const double b = (r < K) ? a.dist[pt * K + r] : 1.0;  // dist holds b (k_mw_rhs)
      aug[(long)r * st + nb] = b;
      xv[r] = b;  // kept for ss = -x.b
    }
    __syncthreads();
    for (int c = 0; c < nb; ++c) {
      double bv = -1.0;
      int br = 0x7fffffff;
      for (int r = c + l; r < nb; r += 256) {
        const double v = fabs(aug[(long)r * st + c]);
        if (v > bv) { bv = v; br = r; }
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const double v2 = __shfl_xor(bv, o, 64);
        const int r2 = __shfl_xor(br, o, 64);
        if (v2 > bv || (v2 == bv && r2 < br)) { bv = v2; br = r2; }
      }
      if ((l & 63) == 0) { redv[l >> 6] = bv; redr[l >> 6] = br; }
      __syncthreads();
      bv = redv[0];
      br = redr[0];
#pragma unroll
      for (int w = 1; w < 4; ++w)
        if (redv[w] > bv || (redv[w] == bv && redr[w] < br)) { bv = redv[w]; br = redr[w]; }
      if (!(bv > 0.0)) bad = 1;
      if (br != c && br < nb)
        for (int j = c + l; j <= nb; j += 256) {
          const double t0 = aug[(long)c * st + j];
          aug[(long)c * st + j] = aug[(long)br * st + j];
          aug[(long)br * st + j] = t0;
        }
      __syncthreads();
      const double pinv = 1.0 / aug[(long)c * st + c];
      for (int r = c + 1 + l; r < nb; r += 256) mul[r] = aug[(long)r * st + c] * pinv;
      __syncthreads();
      const int w = nb - c;         // columns c+1 .. nb (incl. the right-hand side)
      const int rows = nb - c - 1;  // rows below the pivot
      for (long e = l; e < (long)rows * w; e += 256) {
        const int r = c + 1 + (int)(e / w), j = c + 1 + (int)(e - (long)(r - c - 1) * w);
        aug[(long)r * st + j] -= mul[r] * aug[(long)c * st + j];
      }
      __syncthreads();
    }
    // back substitution, column oriented: x[r] = rhs[r] / U[r][r]; rhs[0..r-1] -= U[0..r-1][r] * x[r]
    for (int r = nb - 1; r >= 0; --r) {
      const double x = aug[(long)r * st + nb] / aug[(long)r * st + r];
      __syncthreads();  // everybody has read rhs[r] before row r-1.. are updated again
      for (int i = l; i < r; i += 256) aug[(long)i * st + nb] -= aug[(long)i * st + r] * x;
      if (l == 0) aug[(long)r * st + nb] = x;  // store the solution in place of rhs[r]
      __syncthreads();
    }
    // z = x[:K].Z[sel], ss = -x.b  (block reduction)
    double zz = 0.0, s2 = 0.0;
    for (int r = l; r < nb; r += 256) {
      const double x = aug[(long)r * st + nb];
      if (r < K) zz += x * a.Z[sel[r]];
      s2 += x * xv[r];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      zz += __shfl_xor(zz, o, 64);
      s2 += __shfl_xor(s2, o, 64);
    }
    __shared__ double rz[4], rs[4];
    if ((l & 63) == 0) { rz[l >> 6] = zz; rs[l >> 6] = s2; }
    __syncthreads();
    if (l == 0) {
      a.z[pt] = rz[0] + rz[1] + rz[2] + rz[3];
      a.ss[pt] = -(rs[0] + rs[1] + rs[2] + rs[3]);
    }
  }
  if (bad && l == 0) atomicOr(a.flag, 1);
}

// geographic moving window: the neighbour search runs on unit-sphere Cartesian coordinates (ok.py:934-955), the
// distances handed to the solve are great-circle again (ok.py:962-970)
__global__ void __launch_bounds__(256) k_geo_unit(const double* __restrict__ lon, const double* __restrict__ lat, int n,
                                                  double* __restrict__ ux, double* __restrict__ uy,
                                                  double* __restrict__ uz) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double lo = lon[i] * MIK_PI / 180.0, la = lat[i] * MIK_PI / 180.0;
  ux[i] = cos(lo) * cos(la);
  uy[i] = sin(lo) * cos(la);
  uz[i] = sin(la);
}
__global__ void __launch_bounds__(256)
k_mw_geo_dist(const double* __restrict__ plon, const double* __restrict__ plat, long npt, int K,
              const double* __restrict__ slon, const double* __restrict__ slat, const int* __restrict__ idx,
              double* __restrict__ dist) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= npt * K) return;
  const long t = e / K;
  const int s = idx[e];
  const double la1 = plat[t] * MIK_PI / 180.0, la2 = slat[s] * MIK_PI / 180.0;
  dist[e] = gc_dist(plon[t], cos(la1), sin(la1), slon[s], cos(la2), sin(la2));
}


// coordinates in sorted order / results back in the caller's order (moving window over sorted points: mik_mw.hip, one_predict_mw; the order itself comes from sort_points, mik_predict.hip)
__global__ void __launch_bounds__(256) k_ps_gather(const unsigned* __restrict__ perm, long npt, const double* __restrict__ x,
                                                   const double* __restrict__ y, const double* __restrict__ z,
                                                   double* __restrict__ xs, double* __restrict__ ys, double* __restrict__ zs) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= npt) return;
  const long s = perm[t];
  xs[t] = x[s];
  ys[t] = y[s];
  if (z) zs[t] = z[s];
}
__global__ void __launch_bounds__(256) k_ps_unsort(const unsigned* __restrict__ perm, long npt, const double* __restrict__ a_s,
                                                   const double* __restrict__ b_s, double* __restrict__ a, double* __restrict__ b) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= npt) return;
  const long s = perm[t];
  a[s] = a_s[t];
  b[s] = b_s[t];
}

}  // namespace mik
