// mik_kernels.h -- gfx950 (CDNA4) device code of the kriging execute() path.  fp64 throughout.
//
//   K1  k_assemble            kriging matrix A (or its SPD-shifted form) from station coordinates
//   K2  k_diag_inv, k_panel, k_update (+ k_piv_* for the pivoted path)   block Gauss-Jordan inverse, in place
//   K3a k_rhs                 right-hand sides b_g for a chunk of points (+ z_g = c.b_g), written point-major
//   K3b k_contract            sigma^2_g = -b_g^T A_inv b_g as a dense contraction on v_mfma_f64_4x4x4_4b_f64
//       (k_contract_valu: the same contraction on v_fma_f64, kept as an independent second engine)
//       compact-support (spherical) variogram: k_rhs<.., SP> writes delta = b + s u, k_sp_cand / k_sp_lists_g / k_sp_tiles_g build the
//       lists of active K tiles and the tile records, k_contract_spg contracts tiles of eight gathered 16-row groups (k_contract_sp:
//       aligned 128-row blocks), k_ps_* put the points of every launch in Hilbert-curve order (device radix sort)
//   gemm_core                 the shared MFMA tile loop: LDS-DMA staging, XOR-swizzled LDS, ds_read_b128 fragments
//   k_mw_knn, k_mw_solve      moving-window kriging (n_closest_points)
//   k_stat_*                  variogram-fit statistics (bordered-inverse recursion)
//   k_vg_minmax, k_vg_bin     experimental semivariogram of the constructor
//   NDIM template value 1 = geographic lon/lat (great-circle distance), 2 / 3 = Euclidean
//
// Reference arithmetic restated (paths under /root/reference/src/pykrige): variogram_models.py:25-81,
// ok.py:626-683, uk.py:861-1009, ok3d.py:603-657, uk3d.py:688-811, lib/cok.pyx:56-94.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double d4 __attribute__((ext_vector_type(4)));

namespace mik {

// ------------------------------------------------------------------------------------------------
// variogram functors (variogram_models.py:25-81).  c0 is a host-precomputed constant with the
// reference's own operation order: gaussian (range*4/7)^2, exponential / hole-effect range/3.
// ------------------------------------------------------------------------------------------------
struct Vario {
  int model;
  double p0, p1, p2;
  double c0;     // gaussian (range*4/7)^2 ; exponential / hole-effect range/3
  double c0inv;  // 1 / c0
  double sa, sb; // spherical: 3/(2 range), 1/(2 range^3)
};

// FAST = the per-point right-hand-side path (5e9 evaluations at config 2, VALU-bound): divisions by
// the model constants become multiplications by their host-computed reciprocals (<= 1 ulp change of the
// exp argument; 1e-16 relative on gamma, tolerance is 1e-8).  FAST = false keeps the reference's operation
// order and is used where it is free (the N x N matrix assembly).
template <int MODEL, bool FAST>
__device__ __forceinline__ double vario(const Vario& v, double d, double d2) {
  if (MODEL == 0) return v.p0 * d + v.p1;                               // linear   :25-29
  if (MODEL == 1) return v.p0 * pow(d, v.p1) + v.p2;                    // power    :32-37
  if (MODEL == 2) {                                                     // gaussian :40-45 (needs d^2 only)
    return v.p0 * (1.0 - exp(FAST ? -d2 * v.c0inv : -d2 / v.c0)) + v.p2;
  }
  if (MODEL == 3) {                                                     // spherical:56-70 (d <= range)
    const double r = v.p1;
    if (d <= r) {
      if (FAST) return v.p0 * (d * v.sa - (d2 * d) * v.sb) + v.p2;
      return v.p0 * ((3.0 * d) / (2.0 * r) - (d * d * d) / (2.0 * (r * r * r))) + v.p2;
    }
    return v.p0 + v.p2;
  }
  if (MODEL == 4) return v.p0 * (1.0 - exp(FAST ? -d * v.c0inv : -d / v.c0)) + v.p2;  // exponential :48-53
  {                                                                     // hole-effect :73-81
    const double q = FAST ? d * v.c0inv : d / v.c0;
    return v.p0 * (1.0 - (1.0 - q) * exp(-q)) + v.p2;
  }
}

// point_log drift value incl. the -inf -> -100 rule (uk.py:885-896, 957-966)
__device__ __forceinline__ double well_drift(double x, double y, const double* __restrict__ w) {
  const double dx = x - w[0], dy = y - w[1];
  double ld = log(sqrt(dx * dx + dy * dy));
  if (isinf(ld)) ld = -100.0;
  return -w[2] * ld;
}

// great-circle distance in degrees, arctan form (core.py:36-97), with cos/sin of the latitudes precomputed:
// point 1 = (lon1, c1 = cos(lat1 pi/180), s1 = sin(lat1 pi/180)), point 2 likewise.  Kernels instantiated
// with NDIM == 1 use it instead of the Euclidean distance (coordinates_type='geographic', ok.py:634-640, 990-996).
#define MIK_PI 3.14159265358979323846
__device__ __forceinline__ double gc_dist(double lon1, double c1, double s1, double lon2, double c2, double s2) {
  const double dlon = (lon1 - lon2) * MIK_PI / 180.0;
  double sd, cd;
  sincos(dlon, &sd, &cd);
  const double a = c2 * sd, b = c1 * s2 - s1 * c2 * cd;
  return 180.0 / MIK_PI * atan2(sqrt(a * a + b * b), s1 * s2 + c1 * c2 * cd);
}

// ------------------------------------------------------------------------------------------------
// K1: kriging matrix.  T is Mp x Mp (Mp = M rounded up to 128), row-major, ld = Mp.
//   [i<N, j<N]   -gamma(|X_i - X_j|) + shift, diagonal = 0 + shift   (ok.py:630-644)
//   [i<N, N+c]   drift c at station i, symmetric                     (uk.py:876-910)
//   [i<N, M-1]   1 ; lower-right (p+1)x(p+1) block 0                  (ok.py:645-647, uk.py:915-918)
//   padding      identity (keeps the padded matrix invertible; its inverse is [[A^-1,0],[0,I]])
// shift = 0 gives the reference matrix itself; shift = s > 0 gives A + s.u.u^T with u = [1_N;0],
// whose inverse is A^-1 - s.e_last.e_last^T (A.e_last = u), used by the unpivoted sweep.
// One 64x64 tile per 256-thread block; the tile's row-station coordinates are staged in LDS.
// ------------------------------------------------------------------------------------------------
struct AsmArgs {
  double* T;
  long ld;
  int N, p, M, Mp, ndim;
  const double *xs, *ys, *zs;
  Vario v;
  double shift;
  int rl, nwells, nextra;
  const double* wells;  // nwells x 3
  const double* extra;  // nextra x N
  // drift equilibration (round 4; nullptr = the reference's raw matrix): drift term j enters as (f_j - dsc[2j]) * dsc[2j + 1].
  // With the unbiasedness row present, span{1, f_j} = span{1, s_j (f_j - c_j)}: the kriging weights of the stations, z and sigma^2
  // are unchanged (A' = S A S^T, b' = S b with S = I outside the drift rows), while coordinates of 1e6 next to semivariances of
  // 1e2 (UTM stations under a regional-linear drift: cond(A) 3e14 on the reference's own KT3D test case) no longer sit in one
  // matrix (cond 2e6 there).  mik_get_matrix undoes it.
  const double* dsc;
};

__device__ __forceinline__ double station_drift_raw(const AsmArgs& a, int c, int s) {
  if (a.rl) {
    if (c < a.ndim) return c == 0 ? a.xs[s] : (c == 1 ? a.ys[s] : a.zs[s]);
    c -= a.ndim;
  }
  if (c < a.nwells) return well_drift(a.xs[s], a.ys[s], a.wells + 3 * c);
  c -= a.nwells;
  return a.extra[(long)c * a.N + s];
}
__device__ __forceinline__ double station_drift(const AsmArgs& a, int c, int s) {
  const double v = station_drift_raw(a, c, s);
  return a.dsc ? (v - a.dsc[2 * c]) * a.dsc[2 * c + 1] : v;
}

template <int MODEL, int NDIM>
__global__ void __launch_bounds__(256) k_assemble(AsmArgs a) {
  __shared__ double sx[64], sy[64], sz[64];
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64;
  if (threadIdx.x < 64) {
    const int i = i0 + threadIdx.x;
    const bool st = i < a.N;
    sx[threadIdx.x] = st ? a.xs[i] : 0.0;
    sy[threadIdx.x] = st ? a.ys[i] : 0.0;
    sz[threadIdx.x] = (st && NDIM == 3) ? a.zs[i] : 0.0;
    if (NDIM == 1) {  // geographic: (lon, cos lat, sin lat)
      const double lat = sy[threadIdx.x] * MIK_PI / 180.0;
      sy[threadIdx.x] = cos(lat);
      sz[threadIdx.x] = sin(lat);
    }
  }
  __syncthreads();
  double xj = 0.0, yj = 0.0, zj = 0.0;
  if (j < a.N) {
    xj = a.xs[j];
    yj = a.ys[j];
    if (NDIM == 3) zj = a.zs[j];
    if (NDIM == 1) {
      const double lat = yj * MIK_PI / 180.0;
      yj = cos(lat);
      zj = sin(lat);
    }
  }
  for (int r = threadIdx.x >> 6; r < 64; r += 4) {
    const int i = i0 + r;
    double val;
    if (i >= a.M || j >= a.M) {
      val = (i == j) ? 1.0 : 0.0;
    } else if (i < a.N && j < a.N) {
      if (i == j) {
        val = a.shift;  // np.fill_diagonal(a, 0.0)
      } else {
        double d, s2;
        if (NDIM == 1) {
          d = gc_dist(sx[r], sy[r], sz[r], xj, yj, zj);
          s2 = d * d;
        } else {
          const double dx = sx[r] - xj, dy = sy[r] - yj;
          if (NDIM == 3) {
            const double dz = sz[r] - zj;
            s2 = dx * dx + dy * dy + dz * dz;
          } else {
            s2 = dx * dx + dy * dy;
          }
          d = sqrt(s2);
        }
        // MODEL 7 / 6 = the two passes of a custom (host callable) variogram: 7 leaves the distance in the matrix slot,
        // the host maps d -> gamma(d) over the station block, 6 picks gamma up from the slot
        if (MODEL == 7) val = d;
        else if (MODEL == 6) val = a.shift - a.T[(long)i * a.ld + j];
        else val = a.shift - vario<MODEL, false>(a.v, d, s2);
      }
    } else if (i >= a.N && j >= a.N) {
      val = 0.0;
    } else {
      const int s = i < j ? i : j;
      const int c = (i < j ? j : i) - a.N;
      val = (c == a.p) ? 1.0 : station_drift(a, c, s);
    }
    a.T[(long)i * a.ld + j] = val;
  }
}

// T[idx][idx] += v (corner fix after the shifted inverse)
__global__ void k_add_diag(double* T, long ld, int idx, double v) { T[(long)idx * ld + idx] += v; }

// c_i = sum_{j<N} Ainv[i][j] * Z[j], one wave per row  (z_g = c.b_g; A_inv symmetric)
__global__ void __launch_bounds__(256) k_cvec(const double* __restrict__ Ainv, long ld, int M, int N,
                                              const double* __restrict__ Z, double* __restrict__ c, int Mp) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= Mp) return;
  double s = 0.0;
  if (row < M) {
    const double* r = Ainv + (long)row * ld;
    for (int j = lane; j < N; j += 64) s += r[j] * Z[j];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  }
  if (lane == 0) c[row] = s;
}

// Order-independent checksum of a device array seen as 64-bit words: sum of the words and sum of word x (2 i + 1), both
// modulo 2^64 (integer adds commute, so any grid / any atomic order gives the same two numbers).  Used after the factor
// exchange of a device group: every member's copy of the inverse must carry the leader's checksum -- a broken exchange is
// detected instead of kriging with a wrong inverse.  out[0], out[1] are zeroed by the caller.
__global__ void __launch_bounds__(256) k_checksum(const unsigned long long* __restrict__ w, size_t n, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long sa[4], sb[4];
  unsigned long long a = 0ull, b = 0ull;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const unsigned long long v = w[i];
    a += v;
    b += v * (2ull * (unsigned long long)i + 1ull);
  }
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o);
    b += __shfl_xor(b, o);
  }
  if ((threadIdx.x & 63) == 0) sa[threadIdx.x >> 6] = a, sb[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicAdd(out, sa[0] + sa[1] + sa[2] + sa[3]);
    atomicAdd(out + 1, sb[0] + sb[1] + sb[2] + sb[3]);
  }
}

// ------------------------------------------------------------------------------------------------
// Prediction points of style='grid' / 'masked' generated from the AXES (mik_set_grid): replaces np.meshgrid + the
// anisotropy adjustment of every grid point on the host (ok.py:863-885, ok3d.py:866-883; core.py:120-193) and the H2D
// copy of npt x d doubles -- what crosses PCIe is O(nx + ny [+ nz]).  Point t of the slab is cell cell0 + t of the
// reference's flattened meshgrid (2-D: iy nx + ix; 3-D: (iz ny + iy) nx + ix), or cell cell0 + idx[t] when a mask compacted
// the sequence.  Arithmetic in the reference's order: X -= c ; rot . X ; stretch . (..) ; += c, each dot product
// accumulated k-ascending with fused multiply-adds (what the BLAS kernels behind np.dot do); the result is within an
// ulp of NumPy's, far inside the |d| <= eps = 1e-10 coincidence rule (ok.py:665).  adjust == 0 (geographic
// coordinates, ok.py:892-896): the axes' values as they are.
// ------------------------------------------------------------------------------------------------
struct GridArgs {
  const double *gx, *gy, *gz;  // the axes on the device
  long nx, ny, nz;
  long cell0, n;               // this slab: n points; point t is cell cell0 + t, or cell0 + idx[t] under a mask
  const unsigned* idx;         // nullable: idx[t] = cell (relative to cell0) of the t-th unmasked point of the slab
  int ndim, adjust;
  double c[3], rot[9], st[3];  // centre, rotation (row-major d x d), diagonal of the stretch matrix
  double *px, *py, *pz;
  int from_points;  // 1 = the raw coordinates are already in px / py / pz (mik_adjust_points): transform them in place
};

__global__ void __launch_bounds__(256) k_grid_points(GridArgs a) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= a.n) return;
  double x, y, z = 0.0;
  if (a.from_points) {
    x = a.px[t];
    y = a.py[t];
    if (a.ndim == 3) z = a.pz[t];
  } else {
    const long cell = a.cell0 + (a.idx ? (long)a.idx[t] : t);
    const long ix = cell % a.nx, r = cell / a.nx;
    x = a.gx[ix];
    if (a.ndim == 3) {
      y = a.gy[r % a.ny];
      z = a.gz[r / a.ny];
    } else {
      y = a.gy[r];
    }
  }
  if (a.adjust) {
    // __dmul_rn / __dadd_rn: never contracted into FMAs (hipcc contracts a * b + c by default); only the accumulation of
    // a dot product is fused, like in the BLAS kernel -- measured bit-identical to np.dot on the hosts tried
    const double dx = x - a.c[0], dy = y - a.c[1];
    if (a.ndim == 3) {
      const double dz = z - a.c[2];
      const double r0 = __fma_rn(a.rot[2], dz, __fma_rn(a.rot[1], dy, __dmul_rn(a.rot[0], dx)));
      const double r1 = __fma_rn(a.rot[5], dz, __fma_rn(a.rot[4], dy, __dmul_rn(a.rot[3], dx)));
      const double r2 = __fma_rn(a.rot[8], dz, __fma_rn(a.rot[7], dy, __dmul_rn(a.rot[6], dx)));
      // stretch = diag(1, s_y, s_z): row i of the product is st[i] * r_i plus exact zeros
      x = __dadd_rn(__dmul_rn(a.st[0], r0), a.c[0]);
      y = __dadd_rn(__dmul_rn(a.st[1], r1), a.c[1]);
      z = __dadd_rn(__dmul_rn(a.st[2], r2), a.c[2]);
    } else {
      const double r0 = __fma_rn(a.rot[1], dy, __dmul_rn(a.rot[0], dx));
      const double r1 = __fma_rn(a.rot[3], dy, __dmul_rn(a.rot[2], dx));
      x = __dadd_rn(__dmul_rn(a.st[0], r0), a.c[0]);
      y = __dadd_rn(__dmul_rn(a.st[1], r1), a.c[1]);
    }
  }
  a.px[t] = x;
  a.py[t] = y;
  if (a.ndim == 3) a.pz[t] = z;
}

// ------------------------------------------------------------------------------------------------
// style='masked' (ok.py:700 np.nonzero(~mask); cok.pyx:57-58): the ascending list of the unmasked cells, built on the device
// from the caller's byte mask -- count per 4096-cell block, exclusive scan of the counts by one block, ordered write.  The
// mask buffer is padded with "masked" bytes to a whole number of blocks, so no kernel checks a bound.  Replaces an
// O(cells) host pass that also had to first-touch 8 bytes per unmasked cell.
// ------------------------------------------------------------------------------------------------
#define MIK_MASK_CELLS 4096
__device__ __forceinline__ unsigned mask_zero_bytes(unsigned w) {
  return ((w & 0xffu) == 0u) + ((w & 0xff00u) == 0u) + ((w & 0xff0000u) == 0u) + ((w & 0xff000000u) == 0u);
}

__global__ void __launch_bounds__(256) k_mask_count(const uint4* __restrict__ mask, unsigned* __restrict__ counts) {
  const uint4 m = mask[(size_t)blockIdx.x * 256 + threadIdx.x];
  unsigned c = mask_zero_bytes(m.x) + mask_zero_bytes(m.y) + mask_zero_bytes(m.z) + mask_zero_bytes(m.w);
  for (int o = 32; o; o >>= 1) c += __shfl_down(c, o);
  __shared__ unsigned w[4];
  if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = w[0] + w[1] + w[2] + w[3];
}

// counts[0 .. nblk) -> their exclusive prefix sums in place, counts[nblk] = the total (fewer than 2^32 cells per call)
__global__ void __launch_bounds__(1024) k_mask_scan(unsigned* counts, long nblk) {
  __shared__ unsigned ws[16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned carry = 0;
  for (long base = 0; base < nblk; base += 1024) {
    const long i = base + threadIdx.x;
    const unsigned v = i < nblk ? counts[i] : 0u;
    unsigned s = v;
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(s, o);
      if (lane >= o) s += t;
    }
    if (lane == 63) ws[wv] = s;
    __syncthreads();
    unsigned before = 0, total = 0;
    for (int k = 0; k < 16; ++k) {
      const unsigned x = ws[k];
      before += k < wv ? x : 0u;
      total += x;
    }
    if (i < nblk) counts[i] = carry + before + s - v;
    carry += total;
    __syncthreads();
  }
  if (threadIdx.x == 0) counts[nblk] = carry;
}

__global__ void __launch_bounds__(256) k_mask_write(const uint4* __restrict__ mask, const unsigned* __restrict__ offs, unsigned* __restrict__ idx) {
  const uint4 m = mask[(size_t)blockIdx.x * 256 + threadIdx.x];
  const unsigned c = mask_zero_bytes(m.x) + mask_zero_bytes(m.y) + mask_zero_bytes(m.z) + mask_zero_bytes(m.w);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  unsigned s = c;
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(s, o);
    if (lane >= o) s += t;
  }
  __shared__ unsigned w[4];
  if (lane == 63) w[wv] = s;
  __syncthreads();
  unsigned k = offs[blockIdx.x] + s - c;
  for (int q = 0; q < wv; ++q) k += w[q];
  const unsigned cell = blockIdx.x * (unsigned)MIK_MASK_CELLS + threadIdx.x * 16u;
  const unsigned words[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < 4; ++b)
      if (((words[q] >> (8 * b)) & 0xffu) == 0u) idx[k++] = cell + 4u * q + b;
}

// ------------------------------------------------------------------------------------------------
// K3a: right-hand sides for a chunk of points, written POINT-MAJOR: Bt[t][j], j contiguous, ld = Mp
// (this is the reference's `b` array layout, ok.py:669, and the "NT" operand layout of k_gemm_nt).
//   j <  N      : -gamma(|g_t - X_j|), 0 if |d| <= eps and exact_values  (ok.py:665-672, cok.pyx:196-203)
//   N <= j < N+p: drift rows  (uk.py:949-979; uk3d.py:767-783)
//   j == N+p    : 1           (ok.py:673)             j > N+p : 0 (padding)
// Also z_t = sum_j c_j b_tj (ok.py:680 restated through c = A_inv[:, :n].Z).
// One block = 8 points; threads stride over j so every store is a coalesced row segment.
// ------------------------------------------------------------------------------------------------
#define MIK_TP 8
struct RhsArgs {
  double* Bt;
  long ld;
  int palloc;  // rows of Bt to fill (multiple of 128)
  int nvalid;  // points of this chunk that exist
  const double *px, *py, *pz;  // chunk base pointers
  int N, p, M, Mp, ndim;
  const double *xs, *ys, *zs;
  Vario v;
  int exact;
  double eps;
  int rl, nwells, nextra;
  const double* wells;
  const double* extra;  // chunk base, row stride = extra_stride
  long extra_stride;
  const double* cvec;
  double* zout;  // chunk base
  // SP (range-aware contraction, see k_contract_sp): delta = b + sill on the station entries; only the candidate station blocks
  // of the point block are computed and stored; flags[point block][K tile] = 1 where a nonzero was written
  const unsigned char* cand;  // [point block][nK16]
  unsigned char* flags;       // [point block][nK16]
  int nIblk, nK16;
  double sill;
  const double* dsc;  // drift equilibration, as in AsmArgs (nullptr = raw drift values)
  const unsigned* perm;  // SP, nullable: the chunk's points in sorted order -- point t of the chunk is point perm[t] of the WHOLE list;
                         // px / py / pz / extra / zout are then the list's base pointers, not the chunk's (option "sort_points")
};

template <int MODEL, int NDIM, bool SP = false>
__global__ void __launch_bounds__(256) k_rhs(RhsArgs a) {
  __shared__ double red[4][MIK_TP];
  const int t0 = blockIdx.x * MIK_TP;
  double qx[MIK_TP], qy[MIK_TP], qz[MIK_TP];
  bool ok[MIK_TP];
  long pidx[MIK_TP];  // where point q's coordinates, host-evaluated drift values and z live
#pragma unroll
  for (int q = 0; q < MIK_TP; ++q) {
    ok[q] = (t0 + q) < a.nvalid;
    const long idx = (SP && a.perm) ? (long)a.perm[ok[q] ? t0 + q : 0] : (long)(ok[q] ? t0 + q : 0);
    pidx[q] = idx;
    qx[q] = a.px[idx];
    qy[q] = a.py[idx];
    qz[q] = (NDIM == 3) ? a.pz[idx] : 0.0;
    if (NDIM == 1) {  // geographic: (lon, cos lat, sin lat) of the point
      const double lat = qy[q] * MIK_PI / 180.0;
      qy[q] = cos(lat);
      qz[q] = sin(lat);
    }
  }
  double zacc[MIK_TP];
#pragma unroll
  for (int q = 0; q < MIK_TP; ++q) zacc[q] = 0.0;

  for (int j = threadIdx.x; j < a.Mp; j += 256) {
    double val[MIK_TP];
    if (SP && !a.cand[(long)(t0 >> 7) * a.nK16 + (j >> 4)]) continue;  // per K tile: 16 consecutive lanes leave or stay together
    if (j < a.N) {
      const double sx = a.xs[j];
      double sy = a.ys[j];
      double sz = (NDIM == 3) ? a.zs[j] : 0.0;
      if (NDIM == 1) {
        const double lat = sy * MIK_PI / 180.0;
        sy = cos(lat);
        sz = sin(lat);
      }
#pragma unroll
      for (int q = 0; q < MIK_TP; ++q) {
        double g;
        if (MODEL == 6 || MODEL == 7) {  // custom variogram, see k_assemble: 7 writes d, 6 reads gamma(d) back
          double d;
          if (NDIM == 1) {
            d = gc_dist(qx[q], qy[q], qz[q], sx, sy, sz);
          } else {
            const double dx = qx[q] - sx, dy = qy[q] - sy, dz = (NDIM == 3) ? qz[q] - sz : 0.0;
            d = sqrt(dz * dz + dy * dy + dx * dx);
          }
          if (MODEL == 7) {
            g = d;
          } else {
            g = -a.Bt[(long)(t0 + q) * a.ld + j];
            if (a.exact && d <= a.eps) g = 0.0;
          }
        } else if (NDIM == 1) {
          const double d = gc_dist(qx[q], qy[q], qz[q], sx, sy, sz);  // point first (ok.py:990-996)
          g = -vario<MODEL, true>(a.v, d, d * d);
          if (a.exact && d <= a.eps) g = 0.0;
        } else {
          const double dx = qx[q] - sx, dy = qy[q] - sy;
          double s2;
          if (NDIM == 3) {
            const double dz = qz[q] - sz;
            s2 = dz * dz + dy * dy + dx * dx;
          } else {
            s2 = dx * dx + dy * dy;
          }
          // gaussian needs only d^2: no sqrt, and |d| <= eps becomes d^2 <= eps^2 (ok.py:665: abs(bd) <= eps)
          const double d = (MODEL == 2) ? 0.0 : sqrt(s2);
          g = -vario<MODEL, true>(a.v, d, s2);
          if (a.exact && ((MODEL == 2) ? (s2 <= a.eps * a.eps) : (d <= a.eps))) g = 0.0;
        }
        val[q] = SP ? a.sill + g : g;  // SP: beyond the range g = -(psill + nugget) = -sill exactly, delta = 0 exactly
      }
    } else if (j < a.N + a.p) {
      int c = j - a.N;
      const double dc = a.dsc ? a.dsc[2 * c] : 0.0, ds = a.dsc ? a.dsc[2 * c + 1] : 1.0;
      int kind = 2;  // 0 regional-linear, 1 well, 2 extra
      if (a.rl) {
        if (c < a.ndim) kind = 0; else c -= a.ndim;
      }
      if (kind == 2) {
        if (c < a.nwells) kind = 1; else c -= a.nwells;
      }
#pragma unroll
      for (int q = 0; q < MIK_TP; ++q) {
        double dv;
        if (kind == 0) dv = (c == 0) ? qx[q] : (c == 1 ? qy[q] : qz[q]);
        else if (kind == 1) dv = well_drift(qx[q], qy[q], a.wells + 3 * c);
        else dv = ok[q] ? a.extra[(long)c * a.extra_stride + (SP ? pidx[q] : (long)(t0 + q))] : 0.0;
        val[q] = a.dsc ? (dv - dc) * ds : dv;
      }
    } else {
      const double one = (j == a.N + a.p) ? 1.0 : 0.0;
#pragma unroll
      for (int q = 0; q < MIK_TP; ++q) val[q] = one;
    }
    const double cj = (j < a.M) ? a.cvec[j] : 0.0;
    bool nz = false;
#pragma unroll
    for (int q = 0; q < MIK_TP; ++q) {
      const double v = ok[q] ? val[q] : 0.0;
      a.Bt[(long)(t0 + q) * a.ld + j] = v;
      zacc[q] += cj * v;
      if (SP) nz = nz || v != 0.0;
    }
    if (SP) {  // 16 lanes = one K tile; every writer writes the same 1
      const unsigned long long m = __ballot(nz);
      const int l = threadIdx.x & 63;
      if ((l & 15) == 0 && ((m >> l) & 0xffffULL) != 0) a.flags[(long)(t0 >> 7) * a.nK16 + (j >> 4)] = 1;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < MIK_TP; ++q) {
    double s = zacc[q];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) red[wave][q] = s;
  }
  __syncthreads();
  if (threadIdx.x < MIK_TP && (t0 + (int)threadIdx.x) < a.nvalid) {
    const long o = (SP && a.perm) ? (long)a.perm[t0 + threadIdx.x] : (long)(t0 + threadIdx.x);
    a.zout[o] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
  }
}

// ------------------------------------------------------------------------------------------------
// The fp64 MFMA "NT" GEMM core:  acc[i][t] += sum_k A[i][k] * B[t][k]   (both operands k-contiguous)
// Block tile 128 x 128, 4 waves as 2 x 2, wave tile 64 x 64.  The matrix instruction is
// v_mfma_f64_4x4x4_4b_f64 (4 independent 4x4x4 blocks, 512 flop, ONE accumulator double per lane):
// measured 73 TFLOP/s from one wave per SIMD (16 cycles/instruction) against 47-49 TFLOP/s for
// v_mfma_f64_16x16x4_f64 (~100 cycles for 2048 flop) -- tools/ubench_f64.hip, profiles/.  Its lane
// mapping was probed on the device (tools/probe_mfma4.hip): A lane (k=l>>4, blk=(l>>2)&3, i=l&3),
// B lane (k, blk, j=l&3), D lane (i=l>>4, blk, j=l&3).  The 64 accumulator doubles of a lane are kept as
// acc[ai][bi][r] <-> row 16*ai + 4*r + (l>>4), column 16*bi + (l&15) of the wave tile.
// K is staged in tiles of 16 through double-buffered LDS by LDS-DMA, one barrier per tile.
// ------------------------------------------------------------------------------------------------
#ifndef MIK_CP_A
#define MIK_CP_A ""
#endif
#ifndef MIK_CP_B
#define MIK_CP_B ""
#endif
#define MIK_BM 128
#define MIK_BN 128
#define MIK_BK 16

// K tiles of 128 rows x 16 doubles, UNPADDED (row = 128 B = 8 slots of 16 B) so that the image is
// lane-linear and can be filled by LDS-DMA (global_load_lds_dwordx4: LDS address = wave base + 16*lane,
// no staging VGPRs, no ds_write).  Bank conflicts of the fragment reads are removed by an XOR swizzle
// applied to the per-lane SOURCE address and to the reads: element (row r, k) lives in 16-byte slot
// ((k>>1) ^ swz(r)) of row r, swz = r & 2 for the A tile and (r>>1) & 7 for the B tile.
// row group of a wave inside the block tile.  (Dealing the row groups so that the two waves sharing a SIMD have equal triangular
// diagonal-block work -- {0, 3} / {1, 2} -- was measured: no difference, 23.95 vs 23.93 ms per launch.)
template <int NAI, int BM, bool TRI>
__device__ __forceinline__ int gemm_wm(int wave) {
  return wave >> 1;
}

template <int BM>
struct GemmSmemT {
  double As[2][BM][MIK_BK];
  double Bs[2][MIK_BN][MIK_BK];
  long next;  // persistent kernels: the queue position broadcast to the block (kept inside the one LDS object)
};
typedef GemmSmemT<MIK_BM> GemmSmem;

typedef __attribute__((address_space(1))) const void* mik_gptr_t;
typedef __attribute__((address_space(3))) void* mik_lptr_t;
typedef __attribute__((address_space(4))) const unsigned mik_cu32_t;  // a dword in the constant address space (uniform loads -> s_load)

// NAI = 16-row groups per wave: 4 -> wave tile 64 x 64, 4 waves (256 threads); 2 -> wave tile 32 x 64,
// 8 waves (512 threads).  The block tile is 128 x 128 either way.
// ABL (tools/kernel_bench only; 0 in the library): 32 = generate the B tile on the VALU instead of loading it,
// 1 = skip the LDS-DMA, 2 = skip the fragment ds_reads,
// 4 = skip the per-tile barrier, 8 = DMA always re-reads k-tile 0 (cache-resident source).  Results are garbage; the variants exist to price each component.
// kscale: the accumulators are doubled just before the K tile that starts at kscale is contracted (symmetric
// form: everything above the diagonal block counts twice); pass a value that is never a tile start to disable.
// BM (round 3, tools/kernel_bench only): rows of the block tile, 128 (library) or 256 -- 16 waves of 32 x 64, one block per CU, the
// A operand staged in two passes and the B operand in one (the tile-shape experiment of profiles/r03_kernel_bench.txt).
// TRI (round 3, symmetric contraction): the K range ends with the tile's DIAGONAL block [ktri, ktri + 128) and only its upper
// triangle is contracted, at the granularity of the 16-row accumulator groups: group g of the block (rows ktri + 16 g ..) takes
// the K tiles above its own 16 x 16 diagonal square with weight 2, the square itself with weight 1 and skips the tiles below
// it (their mirror images have been counted twice).  "Weight 2" = the group's accumulators are doubled when the loop reaches
// its square, as kscale does for the whole tile.  36 of the 64 (group, K tile) products of a diagonal block remain; the
// branches are wave-uniform.
template <int NAI, int ABL = 0, int BM = MIK_BM, bool TRI = false>
__device__ __forceinline__ void gemm_core(const double* __restrict__ Ag, long lda, const double* __restrict__ Bg,
                                          long ldb, int kbeg, int kend, d4 (&acc)[NAI][4], GemmSmemT<BM>& sm,
                                          int kscale = -1, int ktri = 0, bool prestaged = false) {
  // prestaged (k_contract PRE): the first K tile (kend - 16) has already been sent to LDS buffer 1 by gemm_prefetch_first()
  // while the block was in the previous tile's epilogue; the loop starts there instead of staging it now
  if (kbeg >= kend) return;  // block-uniform
  constexpr int WROWS = 16 * NAI;            // rows of the wave tile
  constexpr int NTHR = 64 * 2 * (BM / WROWS);
  constexpr int PROWS = NTHR / 8;            // rows staged per pass (8 threads x 16 B per 128-B row)
  constexpr int NPASS_A = BM / PROWS;        // passes over the A tile
  constexpr int NPASS_B = MIK_BN / PROWS;    // passes over the B tile (fewer when BM > MIK_BN, more when BM < MIK_BN)
  constexpr int NPASS = NPASS_A > NPASS_B ? NPASS_A : NPASS_B;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = gemm_wm<NAI, BM, TRI>(wave), wn = wave & 1;
  // staging: thread -> (row lrow + PROWS*p, 16-byte slot tid&7); the SOURCE k-pair is the slot XOR the row's swizzle
  // (A tile: r & 2; B tile: (r>>1) & 7 -- see the fragment reads below).  Both are pass-independent.
  // Addresses are split into a wave-uniform 64-bit base (Ag + k, advanced with scalar adds) and per-lane 32-bit
  // byte offsets fixed for the whole K loop, and the LDS destinations are wave-uniform integers: the K loop then
  // carries no 64-bit vector address arithmetic and no v_readfirstlane per LDS-DMA (they cost ~5 % of the MFMA rate).
  const int lrow = tid >> 3, slot = tid & 7;
  unsigned aoffb[NPASS], boffb[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    aoffb[p] = (unsigned)(((long)(lrow + PROWS * (p < NPASS_A ? p : 0)) * lda + ((slot ^ (lrow & 2)) << 1)) * 8);
    boffb[p] = (unsigned)(((long)(lrow + PROWS * (p < NPASS_B ? p : 0)) * ldb + ((slot ^ ((lrow >> 1) & 7)) << 1)) * 8);
  }
  const unsigned ldsA = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&sm.As[0][wave * 8][0]);
  const unsigned ldsB = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&sm.Bs[0][wave * 8][0]);
  constexpr unsigned LDS_PASS = PROWS * MIK_BK * 8, LDS_BUF = BM * MIK_BK * 8, LDS_BUF_B = MIK_BN * MIK_BK * 8;
  // LDS-DMA in the saddr form (wave-uniform 64-bit base in SGPRs + 32-bit lane offset), written as inline asm:
  // the builtin always materialises a 64-bit per-lane address (2 v_lshl_add_u64 + v_readfirstlane per piece).
  // M0 (LDS destination) is written in the same statement that uses it; hipcc does not count these loads, so the
  // loop drains them itself (s_waitcnt vmcnt(0)) before each barrier.
  auto uniform_ptr = [](const double* q) {  // make the wave-uniformity of a block-uniform pointer provable ("s" operand)
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const double*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
  };
  const double* Agu = uniform_ptr(Ag);
  const double* Bgu = uniform_ptr(Bg);
  auto stage = [&](int k, int b) {
    const double* abase = uniform_ptr(Agu + k);  // once per K tile (hipcc sometimes does the k arithmetic on the VALU)
    const double* bbase = uniform_ptr(Bgu + k);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const unsigned la = ldsA + b * LDS_BUF + p * LDS_PASS, lb = ldsB + b * LDS_BUF_B + p * LDS_PASS;
      // MIK_CP_A / MIK_CP_B: cache-policy modifiers of the two operand streams (tools/kernel_bench experiments: " nt", " sc1", ..)
      if (p == 0) {  // the bases come straight from v_readfirstlane: VALU-written SGPR -> VMEM address needs 5 wait states
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" MIK_CP_A ::"v"(aoffb[p]), "s"(abase), "s"(la) : "memory");
        if (!(ABL & 32)) asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" MIK_CP_B ::"v"(boffb[p]), "s"(bbase), "s"(lb) : "memory");
      } else {
        if (p < NPASS_A) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" MIK_CP_A ::"v"(aoffb[p]), "s"(abase), "s"(la) : "memory");
        if (!(ABL & 32) && p < NPASS_B) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" MIK_CP_B ::"v"(boffb[p]), "s"(bbase), "s"(lb) : "memory");
      }
      if (ABL & 32) {
        // experiment (tools/kernel_bench): the B tile is not loaded but GENERATED -- per thread and pass two
        // exponential-variogram values from a point (its row) and two stations (its k pair), as a kernel fused with
        // the right-hand-side assembly would do -- and written to the slot the DMA would have filled
        const int row = lrow + PROWS * p;
        const double qx = 1e-3 * row, qy = 2e-3 * row;
        const int ks = (k + 2 * slot) & 4094;
        const double2 sx = *reinterpret_cast<const double2*>(Agu + ks), sy = *reinterpret_cast<const double2*>(Agu + lda + ks);
        const double dx0 = qx - sx.x, dy0 = qy - sy.x, dx1 = qx - sx.y, dy1 = qy - sy.y;
        double2 g;
        g.x = -(1.0 - exp(-sqrt(dx0 * dx0 + dy0 * dy0) * 3.3));
        g.y = -(1.0 - exp(-sqrt(dx1 * dx1 + dy1 * dy1) * 3.3));
        *reinterpret_cast<double2*>(&sm.Bs[b][row][slot * 2]) = g;
      }
    }
  };
  auto drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  // Fragment reads are ds_read_b128: lane group kq = lane>>4 owns the k PAIR c = 4m + kq of the 16-wide
  // tile (m = 0, 1), i.e. MFMA step t = 2m + h contracts k = 8m + 2kq + h -- the same bijection of k on
  // both operands.  A: row wm*64 + 4x + i (i = lane&3), identical for the 4 blocks (broadcast);
  // B: row wn*64 + 16x + j (j = lane&15).  With the swizzles above both patterns are bank-conflict
  // free in every 16-lane ds_read_b128 service group.
  const int kq = lane >> 4, ia = lane & 3, jb = lane & 15;
  int aoff[2], boff[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    aoff[m] = (wm * WROWS + ia) * MIK_BK + (((4 * m + kq) ^ (ia & 2)) << 1);
    boff[m] = (wn * 64 + jb) * MIK_BK + (((4 * m + kq) ^ ((jb >> 1) & 7)) << 1);
  }
  // K runs DOWNWARDS (kend-16, kend-32, .. kbeg): in the symmetric contraction every tile then starts at
  // the same k = kend, so the tiles of a supertile stream the same operand panels in near lockstep (L2 reuse).
  int buf = 0;
  if (prestaged) buf = 1;  // block-uniform
  else stage(kend - MIK_BK, 0);
  drain();
  __syncthreads();
  const int kmain = TRI ? (ktri + 128 > kbeg ? ktri + 128 : kbeg) : kbeg;  // TRI: the diagonal block has a loop of its own
  for (int k = kend - MIK_BK; k >= kmain; k -= MIK_BK) {
    if (k > kbeg && !(ABL & 1)) stage((ABL & 8) ? 0 : k - MIK_BK, buf ^ 1);
    const double* as = &sm.As[buf][0][0];
    const double* bs = &sm.Bs[buf][0][0];
    if (k == kscale) {
#pragma unroll
      for (int x = 0; x < NAI; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[x][y] *= 2.0;
    }
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      // v_mfma_f64_4x4x4_4b_f64: A lane (k=l>>4, blk=(l>>2)&3, i=l&3), B lane (k, blk, j=l&3), D lane (i=l>>4, blk, j).
      // A fragments are replicated over the 4 blocks, B fragments put 4 column groups in the 4 blocks, so
      // MFMA (ra, bi) yields rows 4*ra + (l>>4), columns 16*bi + (l&15) of the wave tile.
      double2 fa[4 * NAI], fb[4];
      if (ABL & 2) {
#pragma unroll
        for (int x = 0; x < 4 * NAI; ++x) fa[x] = make_double2(1.0 + x + k, 2.0 - x);
#pragma unroll
        for (int x = 0; x < 4; ++x) fb[x] = make_double2(0.5 + x, 1.5 * x - k);
      } else {
#pragma unroll
        for (int x = 0; x < 4 * NAI; ++x) fa[x] = *reinterpret_cast<const double2*>(as + aoff[m] + 4 * x * MIK_BK);
#pragma unroll
        for (int x = 0; x < 4; ++x) fb[x] = *reinterpret_cast<const double2*>(bs + boff[m] + 16 * x * MIK_BK);
      }
      // all accumulators once (first k of the pair), then all again: dependent MFMAs are >= 32 issues apart
#pragma unroll
      for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int bi = 0; bi < 4; ++bi)
            acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[4 * ai + r].x, fb[bi].x, acc[ai][bi][r], 0, 0, 0);
#pragma unroll
      for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int bi = 0; bi < 4; ++bi)
            acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[4 * ai + r].y, fb[bi].y, acc[ai][bi][r], 0, 0, 0);
    }
    drain();  // the tile staged at the top of this iteration has had the whole compute phase to land
    if (!(ABL & 4)) __syncthreads();
    buf ^= 1;
  }
  if (TRI) {
    // The diagonal block.  This wave's rows are wm*WROWS + 16 ai: accumulator group ai has its 16 x 16 diagonal square in K tile
    // gd0 + ai of the block; in K tile kt the groups ai <= kt - gd0 take part (wave-uniform branches), a group is doubled when
    // the loop reaches its square.  One group at a time: fragments of 16 rows, 16 + 16 MFMAs (dependent ones 16 issues apart).
    const int gd0 = __builtin_amdgcn_readfirstlane(wm * NAI);
    const int ktop = (ktri + 128 < kend ? ktri + 128 : kend) - MIK_BK;
    for (int k = ktop; k >= kbeg; k -= MIK_BK) {
      if (k > kbeg) stage(k - MIK_BK, buf ^ 1);
      const double* as = &sm.As[buf][0][0];
      const double* bs = &sm.Bs[buf][0][0];
      const int alive = ((k - ktri) >> 4) - gd0 + 1;  // groups ai < alive take part in this K tile
#pragma unroll
      for (int ai = 0; ai < NAI; ++ai)
        if (alive == ai + 1) {
#pragma unroll
          for (int y = 0; y < 4; ++y) acc[ai][y] *= 2.0;
        }
      if (alive > 0) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          double2 fb[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) fb[x] = *reinterpret_cast<const double2*>(bs + boff[m] + 16 * x * MIK_BK);
#pragma unroll
          for (int ai = 0; ai < NAI; ++ai)
            if (ai < alive) {
              double2 fa[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) fa[r] = *reinterpret_cast<const double2*>(as + aoff[m] + 4 * (4 * ai + r) * MIK_BK);
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int bi = 0; bi < 4; ++bi)
                  acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[r].x, fb[bi].x, acc[ai][bi][r], 0, 0, 0);
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int bi = 0; bi < 4; ++bi)
                  acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[r].y, fb[bi].y, acc[ai][bi][r], 0, 0, 0);
            }
        }
      }
      drain();
      __syncthreads();
      buf ^= 1;
    }
  }
}

// The first K tile (k = kend - 16) of a 128 x 128 tile into LDS buffer 1, asynchronously: gemm_core's own staging (same thread ->
// (row, slot) map, swizzles and LDS-DMA form, BM = 128), issued by a block that is about to run its previous tile's epilogue --
// that tile's K loop has ended with a barrier, the epilogue reduces through buffer 0.  Nothing is waited for here: the next
// gemm_core call (prestaged = true) drains and synchronises before it reads the buffer.
template <int NAI>
__device__ __forceinline__ void gemm_prefetch_first(const double* __restrict__ Ag, long lda, const double* __restrict__ Bg, long ldb,
                                                    int k, GemmSmem& sm) {
  constexpr int WROWS = 16 * NAI, NTHR = 64 * 2 * (MIK_BM / WROWS), PROWS = NTHR / 8, NPASS = MIK_BM / PROWS;
  const int tid = threadIdx.x, wave = tid >> 6, lrow = tid >> 3, slot = tid & 7;
  const unsigned ldsA = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&sm.As[1][wave * 8][0]);
  const unsigned ldsB = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&sm.Bs[1][wave * 8][0]);
  constexpr unsigned LDS_PASS = PROWS * MIK_BK * 8;
  auto uniform_ptr = [](const double* q) {
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const double*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
  };
  const double* abase = uniform_ptr(Ag + k);
  const double* bbase = uniform_ptr(Bg + k);
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    const unsigned ao = (unsigned)(((long)(lrow + PROWS * p) * lda + ((slot ^ (lrow & 2)) << 1)) * 8);
    const unsigned bo = (unsigned)(((long)(lrow + PROWS * p) * ldb + ((slot ^ ((lrow >> 1) & 7)) << 1)) * 8);
    const unsigned la = ldsA + p * LDS_PASS, lb = ldsB + p * LDS_PASS;
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" MIK_CP_A ::"v"(ao), "s"(abase), "s"(la) : "memory");
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" MIK_CP_B ::"v"(bo), "s"(bbase), "s"(lb) : "memory");
  }
}

// XCD-aware tile index: blocks b, b+8, b+16.. run on the same XCD (block b -> XCD b % 8), so give
// each XCD a contiguous range of logical tiles; neighbours in that range share an operand panel
// in the XCD's private L2.  Launch 8*ceil(total/8) blocks; returns -1 for the overhang.
__device__ __forceinline__ long xcd_tile(long total) {
  const long per = (total + 7) / 8;
  const long L = (long)(blockIdx.x % 8) * per + blockIdx.x / 8;
  return L < total ? L : -1;
}
// the same ranges walked from their ends: XCD x's q-th block takes the q-th tile from the END of the XCD's range
__device__ __forceinline__ long xcd_tile_rev(long total) {
  const long per = (total + 7) / 8;
  const long x = blockIdx.x % 8, q = blockIdx.x / 8;
  const long cnt = (total - x * per < per) ? total - x * per : per;
  return q < cnt ? x * per + (cnt - 1 - q) : -1;
}

// Supertile order for the contraction: each XCD's contiguous range of logical tiles is cut into
// supertiles of MIK_SI row blocks x MIK_ST point blocks = 64 tiles = what 32 CUs x 2 blocks hold at once.
// The co-resident tiles share MIK_SI A row-panels and MIK_ST B point-panels through the XCD's L2.  The row
// blocks of a supertile are adjacent, so in the symmetric form their K extents differ by at most
// MIK_SI-1 blocks and (K running downwards from kend) they stream the panels in near lockstep.
// False = padding slot.
// Shape (round 3, profiles/r03_supertile_shape_ab.txt): 16 row blocks x 4 point blocks.  Rounds 1-2 used 4 x 16; measured in one
// run at config-2 size (symmetric form, 65 536 points): 1 x 64 25.9 ms, 2 x 32 25.0, 4 x 16 25.0, 8 x 8 24.65, 16 x 4 24.5, 32 x 2
// 24.5, 64 x 1 25.3 -- the tall shapes re-read a point panel of B (HBM; the inverse sits in the Infinity Cache) 2.5 x per launch
// instead of 10 x.  -1.4 % at N = 8000, a tie at N = 2000.
#ifndef MIK_SI  // (tools/kernel_bench builds other shapes with -DMIK_SI=.. -DMIK_ST=..; MIK_SI * MIK_ST = 64)
#define MIK_SI 16
#define MIK_ST 4
#endif
__host__ __device__ inline long super_tiles_total(int nIblk, int nTblk) {
  return (long)((nIblk + MIK_SI - 1) / MIK_SI) * ((nTblk + MIK_ST - 1) / MIK_ST) * 64;
}
__host__ __device__ inline long super_grid(int nIblk, int nTblk) {  // blocks to launch
  return 8 * ((((super_tiles_total(nIblk, nTblk) / 64) + 7) / 8) * 64);
}
// queue form: position `seq` of XCD `xcd`'s tile sequence; returns 0 = tile, 1 = padding slot, 2 = sequence exhausted
__device__ __forceinline__ int super_tile_at(int nIblk, int nTblk, int xcd, long seq, int& iblk, int& tblk) {
  const long nsuper = super_tiles_total(nIblk, nTblk) / 64;
  const long s = (seq >> 6) * 8 + xcd;
  if (s >= nsuper) return 2;
  const int r = (int)(seq & 63);
  const int nTg = (nTblk + MIK_ST - 1) / MIK_ST;
#ifdef MIK_DEAL_ROWFAST  // experiment: consecutive supertiles (= the 8 XCDs at one time) are the row groups of ONE point group
  const int nRg = (nIblk + MIK_SI - 1) / MIK_SI;
  const int rg = (int)(s % nRg), tg = (int)(s / nRg);
#else
  const int rg = (int)(s / nTg), tg = (int)(s % nTg);
#endif
#ifdef MIK_POP_ROWFAST  // rounds 1-2: consecutive queue positions walk the row blocks of one point block
  iblk = rg * MIK_SI + (r % MIK_SI);
  tblk = tg * MIK_ST + (r / MIK_SI);
#else  // consecutive positions walk the point blocks of one row block (round 3: -0.8 % per launch, and the shape then hardly matters)
  iblk = rg * MIK_SI + (r / MIK_ST);
  tblk = tg * MIK_ST + (r % MIK_ST);
#endif
  return (iblk < nIblk && tblk < nTblk) ? 0 : 1;
}

// Symmetric form: the queue's unit of work is a PAIR of row blocks (p, nIblk-1-p) of one point block -- the long tile
// (nIblk - p K blocks) followed by the short one (p + 1): nIblk + 1 K blocks whatever p is.  Tiles of the symmetric form
// are 1..nIblk K blocks long; popped one by one, the 64 co-resident blocks of an XCD soon finish at different times, their
// tiles no longer stream the shared operand panels together, and the XCD's L2 stops serving them (measured: 28 % hits,
// against 71 % for the equal-length tiles of the full form).  Equal-length units are popped together and end together, gang
// after gang.  A gang = MIK_SI pair-rows x MIK_ST point blocks = 64 units; position `seq` of XCD `xcd`'s sequence;
// returns 0 = unit, 1 = padding slot, 2 = exhausted.  With an odd nIblk the middle row block stands alone (half a unit);
// it belongs to the last pair-row group, i.e. to the end of the launch.
__device__ __forceinline__ int pair_unit_at(int nIblk, int nTblk, int xcd, long seq, int& p, int& tblk) {
  const int nP = (nIblk + 1) / 2;
  const int nTg = (nTblk + MIK_ST - 1) / MIK_ST;
  const long ngang = (long)((nP + MIK_SI - 1) / MIK_SI) * nTg;
  const long s = (seq >> 6) * 8 + xcd;
  if (s >= ngang) return 2;
  const int r = (int)(seq & 63);
  p = (int)(s / nTg) * MIK_SI + (r % MIK_SI);
  tblk = (int)(s % nTg) * MIK_ST + (r / MIK_SI);
  return (p < nP && tblk < nTblk) ? 0 : 1;
}

__device__ __forceinline__ bool super_tile(int nIblk, int nTblk, int& iblk, int& tblk) {
  const long nsuper = super_tiles_total(nIblk, nTblk) / 64;
  // block b runs on XCD b % 8; its position in that XCD's dispatch sequence is b / 8.  64 consecutive
  // positions of one XCD form one supertile; supertiles are dealt to the XCDs round-robin in global order
  // (row-block groups ascending = longest tiles first in symmetric mode, so a launch ends with its shortest tiles).
  const long seq = blockIdx.x / 8;
  const long s = (seq >> 6) * 8 + (blockIdx.x % 8);
  if (s >= nsuper) return false;
  const int r = (int)(seq & 63);
  const int nTg = (nTblk + MIK_ST - 1) / MIK_ST;
  iblk = (int)(s / nTg) * MIK_SI + (r % MIK_SI);
  tblk = (int)(s % nTg) * MIK_ST + (r / MIK_SI);
  return iblk < nIblk && tblk < nTblk;
}

// ------------------------------------------------------------------------------------------------
// K3b: sigma^2 partials.  Tile (iblk, tblk): W = A_inv[iblk rows, :] . B[:, tblk points] on MFMA,
// then the fused epilogue part[iblk][t] = sum_{i in iblk} b_ti * W_it  (ok.py:681 without the sign;
// k_ss_reduce applies it).  W itself never leaves registers.
// SYM: A_inv is symmetric, so b^T A_inv b = sum_I b_I^T (A_II b_I + 2 sum_{J>I} A_IJ b_J): the K loop
// starts at the diagonal block, which is weighted 1/2 (exact) before the final factor 2.
// Tile order: tblk slow, iblk fast -> consecutive tiles share the B panel; in SYM mode iblk ascending
// is also longest-first.
// ------------------------------------------------------------------------------------------------
// PERSISTENT: the launch is 2 blocks per CU; each block pops tiles from the tile sequence of the XCD it runs on
// (one relaxed device-scope atomicAdd per tile, the XCD id read from HW_REG_XCC_ID) until that sequence is
// exhausted, then helps with the other XCDs' sequences.  With one block per tile the in-order workgroup dispatcher stalls behind whichever XCD is still
// busy once tile lengths differ (symmetric form: 1..nIblk K blocks): measured 8 % of the MFMA rate.
// PERSIST = false is the one-block-per-tile form (grid = super_grid(), queue unused), kept for A/B measurements.
// PAIR (symmetric + persistent only): the queue hands out pairs of row blocks of equal total length (pair_unit_at).
// TRI (symmetric form only): the diagonal block is contracted as a triangle of 16-row groups (gemm_core), 36 instead of 64
// group products per diagonal block.
// PRE (persistent, single tiles): the block pops its NEXT tile before the epilogue of the current one and sends that tile's first
// K tile to LDS (gemm_prefetch_first) -- the queue pop and the first operand fetch of a tile, ~3 us during which the block issued
// nothing, now run under the epilogue's own memory latency.
template <bool SYM, int NAI, bool PERSIST = true, bool PAIR = false, bool TRI = false, bool PRE = false>
__global__ void __launch_bounds__(64 * 2 * (8 / NAI), 2 * (4 / NAI))
k_contract(const double* __restrict__ Ainv, long lda, const double* __restrict__ Bt, long ldb,
           double* __restrict__ part, int palloc, int nIblk, int kend, unsigned long long* __restrict__ queue) {
  constexpr int WROWS = 16 * NAI, NWM = 128 / WROWS;
  __shared__ GemmSmem sm;
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int xcd = (int)(xcc & 7);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = gemm_wm<NAI, MIK_BM, TRI>(wave), wn = wave & 1, lq = lane >> 4, lc = lane & 15;
  int steal = 0;  // 0 = own XCD's sequence; then the other seven in turn: every tile is done whatever the placement
  static_assert(!PAIR || (SYM && PERSIST), "pair units exist for the symmetric persistent form");
  static_assert(!TRI || SYM, "the triangular diagonal block belongs to the symmetric form");
  static_assert(!PRE || (PERSIST && !PAIR), "the prefetch belongs to the persistent single-tile form");
  // one tile's K loop into acc
  auto contract_tile = [&](int iblk, int tblk, d4 (&acc)[NAI][4], bool prestaged) {
    const int i0 = iblk * MIK_BM, t0 = tblk * MIK_BN;
    const double* Ag = Ainv + (long)i0 * lda;
    const double* Bg = Bt + (long)t0 * ldb;
#pragma unroll
    for (int x = 0; x < NAI; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
    if (SYM) {  // result = diag + 2 * offdiag: one K loop downwards from kend; the off-diagonal part is doubled
                // when the loop enters the diagonal block (k < i0 + 128), which is contracted last
      const int kd = (i0 + MIK_BM) < kend ? (i0 + MIK_BM) : kend;
      if (TRI) gemm_core<NAI, 0, MIK_BM, true>(Ag, lda, Bg, ldb, i0, kend, acc, sm, -1, i0, prestaged);  // (a short last block: groups
                                                                                                          // beyond kend hold padding rows, b = 0)
      else gemm_core<NAI>(Ag, lda, Bg, ldb, i0, kend, acc, sm, kd - MIK_BK, 0, prestaged);
    } else {
      gemm_core<NAI>(Ag, lda, Bg, ldb, 0, kend, acc, sm, -1, 0, prestaged);
    }
  };
  // epilogue: column sums of B .* W over this wave's rows; independent loads issued in batches
  // (the fragment registers are dead here); without the scheduling barriers hipcc serialises
  // load -> wait -> fma once per element (~1 us each)
  auto epilogue = [&](int iblk, int tblk, d4 (&acc)[NAI][4]) {
    const int i0 = iblk * MIK_BM, t0 = tblk * MIK_BN;
    double cs[4];
#pragma unroll
    for (int bp = 0; bp < 2; ++bp) {
      double bv[2][4 * NAI];
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const long t = t0 + wn * 64 + (2 * bp + b2) * 16 + lc;
        const double* brow = Bt + t * ldb + i0 + wm * WROWS + lq;
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[b2][ai * 4 + r] = brow[ai * 16 + 4 * r];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int bi = 2 * bp + b2;
        double s = 0.0;
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r) s += bv[b2][ai * 4 + r] * acc[ai][bi][r];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        cs[bi] = s;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    double* red = &sm.As[0][0][0];  // gemm_core ended with a barrier: staging LDS is free (PRE: buffer 1 is being filled)
    if (lq == 0) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) red[wm * 128 + wn * 64 + bi * 16 + lc] = cs[bi];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < NWM; ++w) v += red[w * 128 + threadIdx.x];
      part[(long)iblk * palloc + t0 + threadIdx.x] = v;
    }
  };
  // next position of the tile queues: false when all eight sequences are exhausted
  auto pop = [&](int& iblk, int& tblk, int& pair_p) -> bool {
    for (;;) {
      const int xq = (xcd + steal) & 7;
      if (threadIdx.x == 0)
        sm.next = (long)__hip_atomic_fetch_add(&queue[xq], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const long seq = sm.next;
      const int kind = PAIR ? pair_unit_at(nIblk, palloc / MIK_BN, xq, seq, pair_p, tblk)
                            : super_tile_at(nIblk, palloc / MIK_BN, xq, seq, iblk, tblk);
      __syncthreads();  // everyone has read sm.next (and the previous tile's `red`) before anything is overwritten
      if (kind == 2) {  // this sequence is exhausted: help the next XCD's (correctness never depends on XCC_ID)
        if (++steal == 8) return false;
        continue;
      }
      if (kind == 1) continue;
      return true;
    }
  };
  if (PRE) {
    int iblk = 0, tblk = 0, dummy = 0;
    bool have = pop(iblk, tblk, dummy), pre = false;
    while (have) {
      d4 acc[NAI][4];
      contract_tile(iblk, tblk, acc, pre);
      int ni = 0, nt = 0;
      const bool more = pop(ni, nt, dummy);  // (its barriers also order this tile's K loop before the prefetch's LDS writes)
      if (more)
        gemm_prefetch_first<NAI>(Ainv + (long)ni * MIK_BM * lda, lda, Bt + (long)nt * MIK_BN * ldb, ldb, kend - MIK_BK, sm);
      epilogue(iblk, tblk, acc);
      iblk = ni, tblk = nt, have = more, pre = more;
    }
    return;
  }
  for (;;) {
    int iblk = 0, tblk = 0, pair_p = 0;
    if (PERSIST) {
      if (!pop(iblk, tblk, pair_p)) return;
    } else if (!super_tile(nIblk, palloc / MIK_BN, iblk, tblk)) {
      return;
    }
    for (int half = 0; half < (PAIR ? 2 : 1); ++half) {  // PAIR: the long tile of the pair, then the short one
      if (PAIR) {
        iblk = half == 0 ? pair_p : nIblk - 1 - pair_p;
        if (half == 1) {
          if (iblk == pair_p) break;  // odd nIblk: the middle row block has no partner
          __syncthreads();            // the first tile's `red` has been read before the staging LDS is filled again
        }
      }
      d4 acc[NAI][4];
      contract_tile(iblk, tblk, acc, false);
      epilogue(iblk, tblk, acc);
    }
    if (!PERSIST) return;
  }  // for (;;): next tile of this XCD's sequence
}

// ------------------------------------------------------------------------------------------------
// K3b, VALU engine.  On gfx950 the fp64 vector FMA pipe sustains more than the fp64 matrix pipe
// (tools/ubench_f64.hip, profiles/: v_fma_f64 64-72 TFLOP/s at 2-8 waves/SIMD vs 47-49 for
// v_mfma_f64_16x16x4_f64), so the same contraction is also available as a classic register-tiled
// FMA kernel: 256 threads as 16 x 16, each owning an 8 x 8 micro-tile of the 128 x 128 block tile,
// interleaved in 16-byte chunks (rows ty*2 + 32a + {0,1}, columns tx*2 + 32b + {0,1}) so every
// fragment read is a conflict-free ds_read_b128.  LDS holds the K tile TRANSPOSED (k-major):
// As[k][i], Bs[k][t]; global -> LDS staging is one row per lane (conflict-free ds_write_b64).
// Per k step and thread: 8 ds_read_b128 feed 64 v_fma_f64.
// ------------------------------------------------------------------------------------------------
#define MIK_VS 128  // LDS row stride (doubles) of the k-major tiles
struct ValuSmem {  // one spare k row per tile: the register pipeline reads one row past the end (never used)
  double As[2][MIK_BK + 1][MIK_VS];
  double Bs[2][MIK_BK + 1][MIK_VS];
};

__device__ __forceinline__ void valu_core(const double* __restrict__ Ag, long lda, const double* __restrict__ Bg,
                                          long ldb, int kbeg, int kend, double (&acc)[8][8], ValuSmem& sm) {
  if (kbeg >= kend) return;  // block-uniform
  const int tid = threadIdx.x;
  const int tx = tid & 15, ty = tid >> 4;
  const int srow = tid & 127, sk = (tid >> 7) * 8;  // staging: row srow, k offsets sk .. sk+7
  const double* ap = Ag + (long)srow * lda + sk;
  const double* bp = Bg + (long)srow * ldb + sk;
  double2 ra[4], rb[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    ra[p] = *reinterpret_cast<const double2*>(ap + kbeg + 2 * p);
    rb[p] = *reinterpret_cast<const double2*>(bp + kbeg + 2 * p);
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    sm.As[0][sk + 2 * p][srow] = ra[p].x;
    sm.As[0][sk + 2 * p + 1][srow] = ra[p].y;
    sm.Bs[0][sk + 2 * p][srow] = rb[p].x;
    sm.Bs[0][sk + 2 * p + 1][srow] = rb[p].y;
  }
  __syncthreads();
  int buf = 0;
  for (int k = kbeg; k < kend; k += MIK_BK) {
    const bool more = (k + MIK_BK) < kend;
    if (more) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        ra[p] = *reinterpret_cast<const double2*>(ap + k + MIK_BK + 2 * p);
        rb[p] = *reinterpret_cast<const double2*>(bp + k + MIK_BK + 2 * p);
      }
    }
    {
      // fragments double-buffered in registers: the reads of step kk+1 are in flight behind the 64 FMAs of step kk
      const double* asrc = &sm.As[buf][0][ty * 2];
      const double* bsrc = &sm.Bs[buf][0][tx * 2];
      double2 a0[4], b0[4], a1[4], b1[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        a0[c] = *reinterpret_cast<const double2*>(asrc + 32 * c);
        b0[c] = *reinterpret_cast<const double2*>(bsrc + 32 * c);
      }
#pragma unroll 1
      for (int kk = 0; kk < MIK_BK; kk += 2) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          a1[c] = *reinterpret_cast<const double2*>(asrc + (kk + 1) * MIK_VS + 32 * c);
          b1[c] = *reinterpret_cast<const double2*>(bsrc + (kk + 1) * MIK_VS + 32 * c);
        }
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
          for (int y = 0; y < 8; ++y)
            acc[x][y] = __builtin_fma((x & 1) ? a0[x >> 1].y : a0[x >> 1].x, (y & 1) ? b0[y >> 1].y : b0[y >> 1].x, acc[x][y]);
#pragma unroll
        for (int c = 0; c < 4; ++c) {  // kk + 2 == MIK_BK reads the spare row; those values are discarded
          a0[c] = *reinterpret_cast<const double2*>(asrc + (kk + 2) * MIK_VS + 32 * c);
          b0[c] = *reinterpret_cast<const double2*>(bsrc + (kk + 2) * MIK_VS + 32 * c);
        }
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
          for (int y = 0; y < 8; ++y)
            acc[x][y] = __builtin_fma((x & 1) ? a1[x >> 1].y : a1[x >> 1].x, (y & 1) ? b1[y >> 1].y : b1[y >> 1].x, acc[x][y]);
      }
    }
    if (more) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        sm.As[buf ^ 1][sk + 2 * p][srow] = ra[p].x;
        sm.As[buf ^ 1][sk + 2 * p + 1][srow] = ra[p].y;
        sm.Bs[buf ^ 1][sk + 2 * p][srow] = rb[p].x;
        sm.Bs[buf ^ 1][sk + 2 * p + 1][srow] = rb[p].y;
      }
    }
    __syncthreads();
    buf ^= 1;
  }
}

template <bool SYM>
__global__ void __launch_bounds__(256, 2)
k_contract_valu(const double* __restrict__ Ainv, long lda, const double* __restrict__ Bt, long ldb,
                double* __restrict__ part, int palloc, int nIblk, int kend) {
  __shared__ ValuSmem sm;
  const long L = xcd_tile((long)nIblk * (palloc / MIK_BN));
  if (L < 0) return;
  const int iblk = (int)(L % nIblk), tblk = (int)(L / nIblk);
  const int i0 = iblk * MIK_BM, t0 = tblk * MIK_BN;
  double acc[8][8];
#pragma unroll
  for (int x = 0; x < 8; ++x)
#pragma unroll
    for (int y = 0; y < 8; ++y) acc[x][y] = 0.0;
  const double* Ag = Ainv + (long)i0 * lda;
  const double* Bg = Bt + (long)t0 * ldb;
  if (SYM) {
    const int kd = (i0 + MIK_BM) < kend ? (i0 + MIK_BM) : kend;
    valu_core(Ag, lda, Bg, ldb, i0, kd, acc, sm);
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
      for (int y = 0; y < 8; ++y) acc[x][y] *= 0.5;
    valu_core(Ag, lda, Bg, ldb, i0 + MIK_BM, kend, acc, sm);
  } else {
    valu_core(Ag, lda, Bg, ldb, 0, kend, acc, sm);
  }
  // epilogue: thread (ty,tx) holds rows i0 + ty*2 + 32*(x>>1) + (x&1), columns t0 + tx*2 + 32*(y>>1) + (y&1)
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  double* red = &sm.As[0][0][0];  // 16 x 128 doubles, free after the core's final barrier
#pragma unroll
  for (int yp = 0; yp < 2; ++yp) {
    double2 bv[4][4];
#pragma unroll
    for (int y4 = 0; y4 < 4; ++y4) {
      const int y = 4 * yp + y4;
      const int tc = tx * 2 + 32 * (y >> 1) + (y & 1);
      const double* brow = Bt + (long)(t0 + tc) * ldb + i0 + ty * 2;
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[y4][c] = *reinterpret_cast<const double2*>(brow + 32 * c);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int y4 = 0; y4 < 4; ++y4) {
      const int y = 4 * yp + y4;
      const int tc = tx * 2 + 32 * (y >> 1) + (y & 1);
      double s = 0.0;
#pragma unroll
      for (int c = 0; c < 4; ++c) s += bv[y4][c].x * acc[2 * c][y] + bv[y4][c].y * acc[2 * c + 1][y];
      red[ty * 128 + tc] = s;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  if (threadIdx.x < 128) {
    double v = 0.0;
#pragma unroll
    for (int r = 0; r < 16; ++r) v += red[r * 128 + threadIdx.x];
    part[(long)iblk * palloc + t0 + threadIdx.x] = SYM ? 2.0 * v : v;
  }
}

// ss[t] = -sum_iblk part[iblk][t]   (ok.py:681: sigmasq = sum(x * -b))
__global__ void __launch_bounds__(256) k_ss_reduce(const double* __restrict__ part, int palloc, int nIblk, int nvalid,
                                                   double* __restrict__ ss) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nvalid) return;
  double s = 0.0;
  for (int b = 0; b < nIblk; ++b) s += part[(long)b * palloc + t];
  ss[t] = -s;
}

// ------------------------------------------------------------------------------------------------
// Range-aware contraction for variograms with COMPACT SUPPORT (round 4).  The reference's spherical model is constant beyond
// its range (variogram_models.py:56-70): gamma(d) = s = psill + nugget for d > range.  With u = [1_N; 0] the right-hand side of
// ok.py:669-673 / uk.py:949-981 is b = -s u + delta, where delta_k = s - gamma(d_k) for the stations (EXACTLY zero beyond the
// range; = s at an exact hit, whose b_k is zeroed), delta = b on the drift rows and on the last row.  The kriging matrix has
// A e_last = u (its last column is [1_N; 0], ok.py:645-647, uk.py:915-918), hence A^-1 u = e_last and
//     x = A^-1 b = -s e_last + A^-1 delta ,   z = [Z;0] . x = c . delta ,
//     sigma^2 = -b . x = 2 s - delta^T A^-1 delta        (u . e_last = 0,  u . A^-1 delta = delta_last = 1 = delta . e_last)
// -- the same two numbers from a vector that is mostly zeros.  The stations are laid out along a Hilbert curve (mik_set_problem), so
// 16 consecutive stations are neighbours in space; k_rhs<.., SP> writes delta and records, per block of 128 points, which K tiles
// (16 stations) hold a nonzero; k_sp_lists turns the flags into lists; k_contract_sp contracts, for every ACTIVE row block of a
// point block, only the active K tiles above it and the row block's own (triangular) diagonal block.  Nothing is thresholded:
// a skipped product is a product with exact zeros.
// ------------------------------------------------------------------------------------------------

// candidates: which K tiles (16 consecutive stations of the Hilbert order) can hold a station within `radius` of any of the 128
// points of a point block (bounding boxes; a superset of the truth).  k_rhs computes and stores only these; everything else is
// delta = 0 and is never read.  (Round 4, second session: per K tile; per 128-station block before -- 20-25 % fewer entries of
// delta are computed and written.)  sbox: per K tile lo[3], hi[3] (host, mik_set_problem); tiles [nforced_from, nforced_to) hold the
// drift rows and the last row and are always candidates.  whole128: candidates in whole aligned groups of eight K tiles (the form
// with aligned 128-row blocks reads every K tile of an active block).  One 128-thread block per point block.
// perm (nullable): the launch's points in sorted order, perm[t] = index into px / py / pz (then chunk-independent base pointers)
__global__ void __launch_bounds__(128) k_sp_cand(const double* __restrict__ px, const double* __restrict__ py,
                                                 const double* __restrict__ pz, int nvalid, const double* __restrict__ sbox,
                                                 int nK16, int nforced_from, int nforced_to, double radius,
                                                 unsigned char* __restrict__ cand, const unsigned* __restrict__ perm, int whole128) {
  __shared__ double red[6][2];
  const int tb = blockIdx.x, t = tb * 128 + threadIdx.x;
  const bool ok = t < nvalid;
  double lo[3], hi[3];
  const long ti = (ok && perm) ? (long)perm[t] : t;
  const double c[3] = {ok ? px[ti] : 0.0, ok ? py[ti] : 0.0, (ok && pz) ? pz[ti] : 0.0};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = ok ? c[d] : 1e300;
    hi[d] = ok ? c[d] : -1e300;
    for (int o = 32; o > 0; o >>= 1) {
      lo[d] = fmin(lo[d], __shfl_xor(lo[d], o));
      hi[d] = fmax(hi[d], __shfl_xor(hi[d], o));
    }
    if ((threadIdx.x & 63) == 0) {
      red[d][threadIdx.x >> 6] = lo[d];
      red[3 + d][threadIdx.x >> 6] = hi[d];
    }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    lo[d] = fmin(red[d][0], red[d][1]);
    hi[d] = fmax(red[3 + d][0], red[3 + d][1]);
  }
  const double r2 = radius * radius * (1.0 + 1e-9);
  for (int jb = threadIdx.x; jb < nK16; jb += 128) {
    const double* sb = sbox + 6 * jb;
    double d2 = 0.0;
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const double gap = fmax(0.0, fmax(sb[d] - hi[d], lo[d] - sb[3 + d]));
      d2 += gap * gap;
    }
    bool c = (jb >= nforced_from && jb < nforced_to) || (jb < nforced_from && d2 <= r2);
    if (whole128) {  // k_contract_sp reads whole aligned blocks of 128 rows / 8 K tiles: a candidate makes its seven neighbours candidates
      const unsigned long long m = __ballot(c);  // (jb = lane mod 8 inside a group of eight: nK16 and the stride are multiples of 8)
      c = ((m >> (threadIdx.x & 56)) & 0xffULL) != 0ULL;
    }
    cand[(long)tb * nK16 + jb] = c ? 1 : 0;
  }
}

// flags (one byte per point block and K tile, written by k_rhs SP) -> per point block: the ascending list of active K tiles
// (klist, as k / 16), the ascending list of active ROW blocks (rows: a row block is active when any of its 8 K tiles is), and
// for each active row block the position in klist of the first K tile beyond it (rstart).  One wavefront per point block.
__global__ void __launch_bounds__(64) k_sp_lists(const unsigned char* __restrict__ flags, int nK16, int nIblk,
                                                 unsigned short* __restrict__ klist, int* __restrict__ kcount,
                                                 unsigned short* __restrict__ rows, unsigned short* __restrict__ rstart,
                                                 int* __restrict__ nrows) {
  const int tb = blockIdx.x, lane = threadIdx.x;
  const unsigned char* f = flags + (long)tb * nK16;
  unsigned short* kl = klist + (long)tb * nK16;
  unsigned short* rw = rows + (long)tb * nIblk;
  unsigned short* rs = rstart + (long)tb * nIblk;
  int nk = 0, nr = 0;
  for (int base = 0; base < nK16; base += 64) {
    const int k16 = base + lane;
    const bool on = k16 < nK16 && f[k16] != 0;
    const unsigned long long m = __ballot(on);
    if (on) kl[nk + __popcll(m & ((1ULL << lane) - 1ULL))] = (unsigned short)k16;
    // the 8 row blocks this batch covers: lane l < 8 looks at byte l of the mask
    const bool ract = lane < 8 && ((m >> (8 * lane)) & 0xffULL) != 0 && (base / 8 + lane) < nIblk;
    const unsigned long long rm = __ballot(ract);
    if (ract) {
      const int pos = nr + __popcll(rm & ((1ULL << lane) - 1ULL));
      rw[pos] = (unsigned short)(base / 8 + lane);
      const unsigned long long upto = lane == 7 ? m : (m & ((1ULL << (8 * (lane + 1))) - 1ULL));
      rs[pos] = (unsigned short)(nk + __popcll(upto));
    }
    nk += __popcll(m);
    nr += __popcll(rm);
  }
  if (lane == 0) {
    kcount[tb] = nk;
    nrows[tb] = nr;
  }
}

// The tile sequences of k_contract_sp.  Point blocks are taken in groups of MIK_ST; group g belongs to XCD g % 8 (adjacent point
// blocks have nearly the same active sets: the tiles an XCD has in flight share their row panels of A_inv and their B panels in
// its L2).  Inside a group: row position ascending (= longest K loops first), point block fast.  tiles[] entry = tblk << 10 | rpos.
// xoff[x] .. xoff[x + 1] = XCD x's range of tiles[].  stats: [0] tiles, [1] off-diagonal K tiles summed over the tiles.
// One block of 1024 threads (<= 1024 point blocks per launch).
__global__ void __launch_bounds__(1024) k_sp_tiles(const int* __restrict__ nrows, const int* __restrict__ kcount,
                                                   const unsigned short* __restrict__ rstart, int nIblk, int nTblk,
                                                   unsigned* __restrict__ tiles, int* __restrict__ xoff,
                                                   unsigned long long* __restrict__ stats) {
  __shared__ int gcnt[1024 / MIK_ST + 1], goff[1024 / MIK_ST + 1], xtot[9];
  __shared__ unsigned long long ksum;
  const int nG = (nTblk + MIK_ST - 1) / MIK_ST;
  const int g = threadIdx.x;
  if (g == 0) ksum = 0ULL;
  __syncthreads();
  if (g < nG) {
    int c = 0;
    unsigned long long ks = 0ULL;
    for (int q = 0; q < MIK_ST; ++q) {
      const int tb = g * MIK_ST + q;
      if (tb >= nTblk) break;
      const int nr = nrows[tb], nk = kcount[tb];
      c += nr;
      for (int r = 0; r < nr; ++r) ks += (unsigned long long)(nk - rstart[(long)tb * nIblk + r]);
    }
    gcnt[g] = c;
    atomicAdd(&ksum, ks);
  }
  __syncthreads();
  if (g < 8) {  // exclusive scan of the groups of XCD g
    int s = 0;
    for (int q = g; q < nG; q += 8) {
      goff[q] = s;
      s += gcnt[q];
    }
    xtot[g] = s;
  }
  __syncthreads();
  if (g == 0) {
    int s = 0;
    for (int x = 0; x < 8; ++x) {
      const int c = xtot[x];
      xoff[x] = s;
      s += c;
    }
    xoff[8] = s;
    stats[0] = (unsigned long long)s;
    stats[1] = ksum;
  }
  __syncthreads();
  if (g < nG) {
    int xbase = 0;
    for (int x = 0; x < (g & 7); ++x) xbase += xtot[x];
    unsigned* out = tiles + xbase + goff[g];
    int nr[MIK_ST], maxr = 0;
    for (int q = 0; q < MIK_ST; ++q) {
      const int tb = g * MIK_ST + q;
      nr[q] = tb < nTblk ? nrows[tb] : 0;
      maxr = nr[q] > maxr ? nr[q] : maxr;
    }
    int w = 0;
    for (int r = 0; r < maxr; ++r)
      for (int q = 0; q < MIK_ST; ++q)
        if (r < nr[q]) out[w++] = ((unsigned)(g * MIK_ST + q) << 10) | (unsigned)r;
  }
}

// ss[t] = 2 s - sum over the active row blocks of the point's block  (see the identity above)
__global__ void __launch_bounds__(256) k_ss_reduce_sp(const double* __restrict__ part, int palloc, const int* __restrict__ nrows,
                                                      int nvalid, double two_s, double* __restrict__ ss,
                                                      const unsigned* __restrict__ perm) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= nvalid) return;
  const int nr = nrows[t >> 7];
  double s = 0.0;
  for (int r = 0; r < nr; ++r) s += part[(long)r * palloc + t];
  ss[perm ? (long)perm[t] : (long)t] = two_s - s;  // (perm: ss is then the whole list's base, see k_ps_*)
}

// The tile loop of the range-aware contraction: gemm_core's staging (LDS-DMA, saddr form), LDS image, fragment reads and MFMA
// order (NAI 16-row groups per wave, block tile 128 x 128, K tiles of 16) with the K tiles taken from a LIST: entries
// [vlo, vhi) of kl (k / 16, ascending; all beyond the tile's row block) downwards, then the row block's own diagonal block
// [ktri, min(ktri + 128, kend)) as a triangle of 16-row groups exactly as gemm_core<.., TRI> does it (a group's accumulators
// are doubled when the loop reaches its 16 x 16 square).
template <int NAI>
__device__ __forceinline__ void gemm_core_sp(const double* __restrict__ Ag, long lda, const double* __restrict__ Bg, long ldb,
                                             const unsigned short* kl, int vlo, int vhi, int ktri, int kend, d4 (&acc)[NAI][4],
                                             GemmSmem& sm) {
  constexpr int WROWS = 16 * NAI;
  constexpr int NTHR = 64 * 2 * (MIK_BM / WROWS);
  constexpr int PROWS = NTHR / 8;
  constexpr int NPASS = MIK_BM / PROWS;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = tid >> 3, slot = tid & 7;
  unsigned aoffb[NPASS], boffb[NPASS];
#pragma unroll
  for (int p = 0; p < NPASS; ++p) {
    aoffb[p] = (unsigned)(((long)(lrow + PROWS * p) * lda + ((slot ^ (lrow & 2)) << 1)) * 8);
    boffb[p] = (unsigned)(((long)(lrow + PROWS * p) * ldb + ((slot ^ ((lrow >> 1) & 7)) << 1)) * 8);
  }
  const unsigned ldsA = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&sm.As[0][wave * 8][0]);
  const unsigned ldsB = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&sm.Bs[0][wave * 8][0]);
  constexpr unsigned LDS_PASS = PROWS * MIK_BK * 8, LDS_BUF = MIK_BM * MIK_BK * 8;
  auto uniform_ptr = [](const double* q) {
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const double*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
  };
  const double* Agu = uniform_ptr(Ag);
  const double* Bgu = uniform_ptr(Bg);
  auto stage = [&](int k, int b) {
    const double* abase = uniform_ptr(Agu + k);
    const double* bbase = uniform_ptr(Bgu + k);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const unsigned la = ldsA + b * LDS_BUF + p * LDS_PASS, lb = ldsB + b * LDS_BUF + p * LDS_PASS;
      if (p == 0) {
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(aoffb[p]), "s"(abase), "s"(la) : "memory");
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(boffb[p]), "s"(bbase), "s"(lb) : "memory");
      } else {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(aoffb[p]), "s"(abase), "s"(la) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(boffb[p]), "s"(bbase), "s"(lb) : "memory");
      }
    }
  };
  auto drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  const int kq = lane >> 4, ia = lane & 3, jb = lane & 15;
  int aoff[2], boff[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    aoff[m] = (wm * WROWS + ia) * MIK_BK + (((4 * m + kq) ^ (ia & 2)) << 1);
    boff[m] = (wn * 64 + jb) * MIK_BK + (((4 * m + kq) ^ ((jb >> 1) & 7)) << 1);
  }
  const int ktop = (ktri + 128 < kend ? ktri + 128 : kend) - MIK_BK;  // first K tile of the diagonal block
  int buf = 0;
  stage(vhi > vlo ? 16 * (int)kl[vhi - 1] : ktop, 0);
  drain();
  __syncthreads();
  for (int v = vhi - 1; v >= vlo; --v) {
    stage(v > vlo ? 16 * (int)kl[v - 1] : ktop, buf ^ 1);
    const double* as = &sm.As[buf][0][0];
    const double* bs = &sm.Bs[buf][0][0];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      double2 fa[4 * NAI], fb[4];
#pragma unroll
      for (int x = 0; x < 4 * NAI; ++x) fa[x] = *reinterpret_cast<const double2*>(as + aoff[m] + 4 * x * MIK_BK);
#pragma unroll
      for (int x = 0; x < 4; ++x) fb[x] = *reinterpret_cast<const double2*>(bs + boff[m] + 16 * x * MIK_BK);
#pragma unroll
      for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int bi = 0; bi < 4; ++bi)
            acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[4 * ai + r].x, fb[bi].x, acc[ai][bi][r], 0, 0, 0);
#pragma unroll
      for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int bi = 0; bi < 4; ++bi)
            acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[4 * ai + r].y, fb[bi].y, acc[ai][bi][r], 0, 0, 0);
    }
    drain();
    __syncthreads();
    buf ^= 1;
  }
  // the diagonal block (gemm_core TRI)
  const int gd0 = __builtin_amdgcn_readfirstlane(wm * NAI);
  for (int k = ktop; k >= ktri; k -= MIK_BK) {
    if (k > ktri) stage(k - MIK_BK, buf ^ 1);
    const double* as = &sm.As[buf][0][0];
    const double* bs = &sm.Bs[buf][0][0];
    const int alive = ((k - ktri) >> 4) - gd0 + 1;
#pragma unroll
    for (int ai = 0; ai < NAI; ++ai)
      if (alive == ai + 1) {
#pragma unroll
        for (int y = 0; y < 4; ++y) acc[ai][y] *= 2.0;
      }
    if (alive > 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        double2 fb[4];
#pragma unroll
        for (int x = 0; x < 4; ++x) fb[x] = *reinterpret_cast<const double2*>(bs + boff[m] + 16 * x * MIK_BK);
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
          if (ai < alive) {
            double2 fa[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) fa[r] = *reinterpret_cast<const double2*>(as + aoff[m] + 4 * (4 * ai + r) * MIK_BK);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int bi = 0; bi < 4; ++bi)
                acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[r].x, fb[bi].x, acc[ai][bi][r], 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
              for (int bi = 0; bi < 4; ++bi)
                acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[r].y, fb[bi].y, acc[ai][bi][r], 0, 0, 0);
          }
      }
    }
    drain();
    __syncthreads();
    buf ^= 1;
  }
}

#define MIK_SP_MAXK16 4096  // K tiles a point block's list can hold in LDS (Mp <= 65536)
struct SpArgs {
  const double* Ainv;
  long lda;
  const double* Bt;
  long ldb;
  double* part;
  int palloc, kend, nIblk, nK16;
  const unsigned short* klist;   // [tblk][nK16]
  const int* kcount;             // [tblk]
  const unsigned short* rows;    // [tblk][nIblk]
  const unsigned short* rstart;  // [tblk][nIblk]
  const unsigned* tiles;
  const int* xoff;               // [9]
  unsigned long long* queue;     // [8]
};

// Persistent like k_contract: 2 blocks per CU pop tiles from the sequence of the XCD they run on, then from the others'.
// Tile = (point block tblk, position rpos in its list of active row blocks): W = A_inv[row block, active K tiles] . delta, fused
// epilogue part[rpos][t] = sum_i delta_ti W_it (k_contract's, indexed by the position instead of the row block).
template <int NAI>
__global__ void __launch_bounds__(64 * 2 * (8 / NAI), 2 * (4 / NAI)) k_contract_sp(SpArgs a) {
  constexpr int WROWS = 16 * NAI, NWM = 128 / WROWS;
  __shared__ GemmSmem sm;
  __shared__ unsigned short skl[MIK_SP_MAXK16];
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int xcd = (int)(xcc & 7);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, lq = lane >> 4, lc = lane & 15;
  int steal = 0;
  for (;;) {
    // next position of the tile queues
    unsigned entry = 0;
    for (;;) {
      const int xq = (xcd + steal) & 7;
      if (threadIdx.x == 0) sm.next = (long)__hip_atomic_fetch_add(&a.queue[xq], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      const long seq = sm.next;
      const int lo = a.xoff[xq], hi = a.xoff[xq + 1];
      const bool have = seq < (long)(hi - lo);
      if (have) entry = a.tiles[lo + seq];
      __syncthreads();  // everyone has read sm.next (and the previous tile's `red`, and is out of its K loop: skl is free)
      if (have) break;
      if (++steal == 8) return;
    }
    const int tblk = (int)(entry >> 10), rpos = (int)(entry & 1023u);
    const int iblk = a.rows[(long)tblk * a.nIblk + rpos];
    const int vlo = a.rstart[(long)tblk * a.nIblk + rpos], vhi = a.kcount[tblk];
    {  // this tile's part of the K-tile list into LDS
      const unsigned short* src = a.klist + (long)tblk * a.nK16;
      for (int v = vlo + (int)threadIdx.x; v < vhi; v += (int)blockDim.x) skl[v] = src[v];
    }
    __syncthreads();
    const int i0 = iblk * MIK_BM, t0 = tblk * MIK_BN;
    d4 acc[NAI][4];
#pragma unroll
    for (int x = 0; x < NAI; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
    gemm_core_sp<NAI>(a.Ainv + (long)i0 * a.lda, a.lda, a.Bt + (long)t0 * a.ldb, a.ldb, skl, vlo, vhi, i0, a.kend, acc, sm);
    // epilogue (k_contract's)
    double cs[4];
#pragma unroll
    for (int bp = 0; bp < 2; ++bp) {
      double bv[2][4 * NAI];
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const long t = t0 + wn * 64 + (2 * bp + b2) * 16 + lc;
        const double* brow = a.Bt + t * a.ldb + i0 + wm * WROWS + lq;
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[b2][ai * 4 + r] = brow[ai * 16 + 4 * r];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int bi = 2 * bp + b2;
        double s = 0.0;
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r) s += bv[b2][ai * 4 + r] * acc[ai][bi][r];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        cs[bi] = s;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    double* red = &sm.As[0][0][0];
    if (lq == 0) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) red[wm * 128 + wn * 64 + bi * 16 + lc] = cs[bi];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < NWM; ++w) v += red[w * 128 + threadIdx.x];
      a.part[(long)rpos * a.palloc + t0 + threadIdx.x] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Range-aware contraction, GATHERED ROW GROUPS (round 4, second session; option "sparse_rows" 16).  k_contract_sp above takes the
// rows of A_inv in aligned blocks of 128: a row block is contracted whole when one of its eight 16-station groups is in range
// (active row blocks are 79 % full at BASELINE config 5).  Here a tile's 128 rows are ANY eight active 16-row groups: the ascending
// list of a point block's active K tiles (klist) is also the list of its active row groups, tile r takes entries [8r, 8r + 8) as
// rows and entries [8r + 8, nk) as its off-diagonal K tiles, then its own eight groups as the triangular diagonal part (the group
// in list position j is contracted with the K tiles of positions >= j: doubled accumulators + the 16 x 16 square, as gemm_core's
// TRI form does inside an aligned block).  Skipped groups and K tiles hold exact zeros of delta, so this is the same sum.
// What else differs from k_contract_sp:
//  * rows are dealt to the LDS image so that wave-row wm owns list positions {wm, wm + 4}: the triangular part then needs
//    2+2+2+2+1+1+1+1 = 12 group-steps per wave instead of 15 (the per-K-tile barrier makes a step as long as its busiest wave);
//  * a tile arrives as ONE 32-byte record written by k_sp_tiles_g (tile id, nk, the first two K tiles, the eight row groups):
//    the queue position of the NEXT tile is fetched (atomic) when the current tile starts, its record is read and its first K
//    tile sent to LDS before the current tile's epilogue -- the pop -> metadata -> list -> first-fetch chain of k_contract_sp
//    (about six dependent memory round trips per ~35-K-tile tile) is one LDS broadcast;
//  * K-tile ids are read from the list in global memory two steps ahead (one wave-uniform load per step, waited for by the
//    step's own drain): no list in LDS, no list copy.
// 32-bit LDS-DMA offsets address the whole inverse here (rows are anywhere): the host takes this form only while Mp * lda * 8 < 2^32.
// ------------------------------------------------------------------------------------------------

// flags -> klist / kcount as k_sp_lists, and the number of 128-row tiles of gathered groups: ceil(nk / 8)
__global__ void __launch_bounds__(64) k_sp_lists_g(const unsigned char* __restrict__ flags, int nK16,
                                                   unsigned short* __restrict__ klist, int* __restrict__ kcount,
                                                   int* __restrict__ ntiles) {
  const int tb = blockIdx.x, lane = threadIdx.x;
  const unsigned char* f = flags + (long)tb * nK16;
  unsigned short* kl = klist + (long)tb * nK16;
  int nk = 0;
  for (int base = 0; base < nK16; base += 64) {
    const int k16 = base + lane;
    const bool on = k16 < nK16 && f[k16] != 0;
    const unsigned long long m = __ballot(on);
    if (on) kl[nk + __popcll(m & ((1ULL << lane) - 1ULL))] = (unsigned short)k16;
    nk += __popcll(m);
  }
  if (lane == 0) {
    kcount[tb] = nk;
    ntiles[tb] = (nk + 7) / 8;
  }
}

// Tile records of k_contract_spg, in k_sp_tiles' order (groups of MIK_ST point blocks, group g on XCD g % 8, inside a group tile
// position ascending = longest K loops first, point block fast).  Record (two uint4):
//   [0] = {tblk << 10 | r, nk, klist[nk - 1], klist[nk - 2]}      [1] = the eight row groups klist[8 r .. 8 r + 7] (u16 each)
// stats: [0] tiles, [1] off-diagonal K tiles summed over the tiles, [2] (row group, K tile) products of the triangular parts.
__global__ void __launch_bounds__(1024) k_sp_tiles_g(const int* __restrict__ ntiles, const int* __restrict__ kcount,
                                                     const unsigned short* __restrict__ klist, int nK16, int nTblk,
                                                     uint4* __restrict__ recs, int* __restrict__ xoff,
                                                     unsigned long long* __restrict__ stats, int st) {
  // st = point blocks per group (option "sparse_group", 1 .. 16; 4 by default = MIK_ST)
  __shared__ int gcnt[1024 + 1], goff[1024 + 1], xtot[9];
  __shared__ unsigned long long ksum, dsum;
  const int nG = (nTblk + st - 1) / st;
  const int g = threadIdx.x;
  if (g == 0) ksum = 0ULL, dsum = 0ULL;
  __syncthreads();
  if (g < nG) {
    int c = 0;
    unsigned long long ks = 0ULL, ds = 0ULL;
    for (int q = 0; q < st; ++q) {
      const int tb = g * st + q;
      if (tb >= nTblk) break;
      const int nk = kcount[tb], full = nk / 8, rem = nk - 8 * full;
      c += ntiles[tb];
      ks += (unsigned long long)((long)full * nk - 4L * full * (full + 1));  // sum over full tiles r of nk - 8 (r + 1)
      ds += (unsigned long long)(36 * full + rem * (rem + 1) / 2);
    }
    gcnt[g] = c;
    atomicAdd(&ksum, ks);
    atomicAdd(&dsum, ds);
  }
  __syncthreads();
  if (g < 8) {  // exclusive scan of the groups of XCD g
    int s = 0;
    for (int q = g; q < nG; q += 8) {
      goff[q] = s;
      s += gcnt[q];
    }
    xtot[g] = s;
  }
  __syncthreads();
  if (g == 0) {
    int s = 0;
    for (int x = 0; x < 8; ++x) {
      const int c = xtot[x];
      xoff[x] = s;
      s += c;
    }
    xoff[8] = s;
    stats[0] = (unsigned long long)s;
    stats[1] = ksum;
    stats[2] = dsum;
  }
  __syncthreads();
  const int tb = threadIdx.x;  // one thread per point block writes that block's records
  if (tb < nTblk) {
    const int gg = tb / st, q = tb % st;
    int xbase = 0;
    for (int x = 0; x < (gg & 7); ++x) xbase += xtot[x];
    int nr[16];
    for (int qq = 0; qq < 16; ++qq) {
      const int t2 = gg * st + qq;
      nr[qq] = (qq < st && t2 < nTblk) ? ntiles[t2] : 0;
    }
    const int nk = kcount[tb];
    const unsigned short* kl = klist + (long)tb * nK16;
    const unsigned k1 = nk >= 1 ? kl[nk - 1] : 0u, k2 = nk >= 2 ? kl[nk - 2] : 0u;
    int w = xbase + goff[gg];
    for (int r = 0; r < nr[q]; ++r) {  // (tiles of the other point blocks beyond nr[q] lie behind this block's last one or belong to them)
      int before = 0, all = 0;
      for (int qq = 0; qq < 16; ++qq) {
        const int on = nr[qq] > r ? 1 : 0;
        all += on;
        if (qq < q) before += on;
      }
      uint4* out = recs + 2L * (w + before);
      out[0] = make_uint4(((unsigned)tb << 10) | (unsigned)r, (unsigned)nk, k1, k2);
      out[1] = *reinterpret_cast<const uint4*>(kl + 8 * r);  // 16-byte aligned: nK16 is a multiple of 8
      w += all;
    }
  }
}

struct SpgArgs {
  const double* Ainv;
  long lda;
  const double* Bt;
  long ldb;
  double* part;
  int palloc, nK16;
  const unsigned short* klist;  // [tblk][nK16]
  const uint4* recs;            // tile records (k_sp_tiles_g)
  const int* xoff;              // [9]
  unsigned long long* queue;    // [8], zeroed per launch (the low words are the counters)
};

// EPI (option "sparse_epilogue" 1; not the default): a group's term of part[r][t] = sum_i delta_ti W_it is formed at the K step of the
// group's own 16 x 16 square -- its accumulators are final there, and the delta it needs IS that step's B tile in LDS -- instead of
// from global memory after the K loop: no operand reads in the epilogue (a tenth of the kernel's fabric traffic, two memory round
// trips per tile).  Measured 1.7 % slower (config 5: 43.1 against 42.4 ms): the sums live in registers through the triangle loop.
template <int NAI, bool EPI = false>
__global__ void __launch_bounds__(64 * 2 * (8 / NAI), 2 * (4 / NAI)) k_contract_spg(SpgArgs a) {
  static_assert(NAI == 2, "8 waves: 4 wave-rows of two 16-row groups x 2 wave-columns of 64 points");
  constexpr int WROWS = 16 * NAI, NWM = 128 / WROWS;
  constexpr int NTHR = 64 * 2 * (MIK_BM / WROWS), PROWS = NTHR / 8, NPASS = MIK_BM / PROWS;
  constexpr unsigned LDS_PASS = PROWS * MIK_BK * 8, LDS_BUF = MIK_BM * MIK_BK * 8;
  __shared__ GemmSmem sm;
  __shared__ uint4 srec[4];  // two tile records: the current tile's and the next one's
  __shared__ int sst[4];     // thread 0's queue state: [0] sequences tried, [1] first record and [2] record count of the current sequence
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int xcd = (int)(xcc & 7);
  // The kernel sits at its register budget (128 VGPRs = 4 wavefronts per SIMD) inside the K loop; nothing lane-dependent may stay
  // live across it except what the loop itself needs.  The wave index is kept in a scalar register, the lane index is re-derived
  // (v_mbcnt, opaque to the optimiser) wherever the code between two K loops needs it.
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int wm = wave >> 1, wn = wave & 1;
  auto lane_now = []() -> int {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
  };
  const unsigned ldsA = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&sm.As[0][wave * 8][0]);
  const unsigned ldsB = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&sm.Bs[0][wave * 8][0]);
  auto uniform_ptr = [](const double* q) {
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const double*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
  };
  auto drain = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
  const double* Agu = uniform_ptr(a.Ainv);
  // what the K loop needs per lane: the DMA source offsets of its two staged rows of each operand, its fragment offsets in the LDS image
  // (one register each where gemm_core keeps two: the second pass of the B operand is the first one PROWS rows further down -- a
  // scalar base; the fragment offsets of the upper K half are aoff + 8 -- an instruction offset -- and boff ^ 8 -- one XOR per step)
  unsigned aoffb[NPASS], boffb;
  int aoff, boff;
  {
    const int lane = (int)(threadIdx.x & 63), tid = wave * 64 + lane;
    const int lrow = tid >> 3, slot = tid & 7;
    boffb = (unsigned)(((long)lrow * a.ldb + ((slot ^ ((lrow >> 1) & 7)) << 1)) * 8);
    const int kq = lane >> 4, ia = lane & 3, jb = lane & 15;
    aoff = (wm * WROWS + ia) * MIK_BK + ((kq ^ (ia & 2)) << 1);          // m = 1: (4 + kq) ^ (ia & 2) = 4 + (kq ^ (ia & 2))
    boff = (wn * 64 + jb) * MIK_BK + ((kq ^ ((jb >> 1) & 7)) << 1);      // m = 1: ((4 + kq) ^ s) << 1 = ((kq ^ s) << 1) ^ 8
  }
  auto boff_hi = [&]() -> int {  // boff ^ 8 formed per K step (the empty asm keeps it from being hoisted into a register of its own)
    int b = boff;
    asm volatile("" : "+v"(b));
    return b ^ 8;
  };
  // LDS row slot s (16 rows) holds the row group of list position (s >> 1) + 4 (s & 1): wave-row wm owns positions wm and wm + 4
  auto group_of = [](const uint4& r1, int gi) -> unsigned {
    const unsigned w = gi < 2 ? r1.x : gi < 4 ? r1.y : gi < 6 ? r1.z : r1.w;
    return (w >> (16 * (gi & 1))) & 0xffffu;
  };
  auto stage = [&](const double* Bgu, int k, int b) {
    const double* abase = uniform_ptr(Agu + k);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const double* bbase = uniform_ptr(Bgu + (long)(PROWS * p) * a.ldb + k);
      const unsigned la = ldsA + b * LDS_BUF + p * LDS_PASS, lb = ldsB + b * LDS_BUF + p * LDS_PASS;
      if (p == 0) {
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(aoffb[p]), "s"(abase), "s"(la) : "memory");
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(boffb), "s"(bbase), "s"(lb) : "memory");
      } else {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(aoffb[p]), "s"(abase), "s"(la) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(boffb), "s"(bbase), "s"(lb) : "memory");
      }
    }
  };
  // Thread 0 owns the queue, one tile ahead.  fetch_next() sits right behind the first barrier of a tile's K loop: it pops the position
  // of the NEXT tile (atomic) and reads that tile's record into the other half of srec -- wavefront 0 waits two L2 round trips there
  // while the other wavefronts of its SIMD use the matrix pipe, and catches up inside the same K step.  (Keeping the atomic's result
  // in a register until the tile ends does not work: hipcc waits for it at once and spills it.)  acquire(), after the K loop, then
  // finds the record in LDS; only when a sequence has run out does it walk on to the next XCD's (a few times per block and launch).
  constexpr unsigned REC_END = 0xffffffffu, REC_MORE = 0xfffffffeu;
  auto fetch = [&](int xq) { return __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(&a.queue[xq]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  int cur = 1;  // srec[2 cur], srec[2 cur + 1] = the current tile's record
  auto fetch_next = [&]() {
    if (threadIdx.x == 0) {
      const int steal = sst[0];
      uint4 r0 = make_uint4(REC_END, 0u, 0u, 0u), r1 = make_uint4(0u, 0u, 0u, 0u);
      if (steal < 8) {
        const unsigned seq = fetch((xcd + steal) & 7);
        if (seq < (unsigned)sst[2]) {
          const uint4* rp = a.recs + 2L * (sst[1] + (long)seq);
          r0 = rp[0];
          r1 = rp[1];
        } else {
          r0.x = REC_MORE;
        }
      }
      srec[2 * (cur ^ 1)] = r0;
      srec[2 * (cur ^ 1) + 1] = r1;
    }
  };
  auto acquire = [&]() -> bool {  // one barrier; block-uniform result
    if (threadIdx.x == 0 && srec[2 * (cur ^ 1)].x == REC_MORE) {
      uint4 r0 = make_uint4(REC_END, 0u, 0u, 0u), r1 = make_uint4(0u, 0u, 0u, 0u);
      int steal = sst[0];
      while (++steal < 8) {
        const int xq = (xcd + steal) & 7;  // help the next XCD's sequence
        const int qlo = a.xoff[xq], qcnt = a.xoff[xq + 1] - qlo;
        const unsigned seq = fetch(xq);
        if (seq < (unsigned)qcnt) {
          const uint4* rp = a.recs + 2L * (qlo + (long)seq);
          r0 = rp[0];
          r1 = rp[1];
          sst[1] = qlo;
          sst[2] = qcnt;
          break;
        }
      }
      sst[0] = steal;
      srec[2 * (cur ^ 1)] = r0;
      srec[2 * (cur ^ 1) + 1] = r1;
    }
    __syncthreads();
    cur ^= 1;
    return __builtin_amdgcn_readfirstlane(srec[2 * cur].x) != REC_END;
  };
  // the current tile's state: block- or wave-uniform values in scalar registers
  int tblk, rpos, n, ksec, erow[NAI];
  const mik_cu32_t* ksrc;  // the point block's list from this tile's first group on (16-byte aligned), as dwords in the CONSTANT address
                           // space: a uniform load from there is a scalar load (s_load_dword: no vector registers, no vmcnt); the list
                           // was written by an earlier kernel and is not modified during this one
  const double* Bgu;
  auto list_at = [&](int i) -> int { return (int)((ksrc[i >> 1] >> (16 * (i & 1))) & 0xffffu); };
  auto adopt = [&]() {  // srec -> the state above, first K tile into buffer 1 (nothing is waited for)
    const uint4 r0 = srec[2 * cur], r1 = srec[2 * cur + 1];
    const unsigned tile = __builtin_amdgcn_readfirstlane(r0.x);
    tblk = (int)(tile >> 10);
    rpos = (int)(tile & 1023u);
    const int nk = __builtin_amdgcn_readfirstlane((int)r0.y), g0 = 8 * rpos;
    const int kfirst = __builtin_amdgcn_readfirstlane((int)r0.z);
    n = nk - g0;  // K tiles of this tile: n - 8 off-diagonal ones, then its own min(n, 8) groups
    ksec = __builtin_amdgcn_readfirstlane((int)r0.w);
    const int ng = n < 8 ? n : 8;
    {  // byte offsets (relative to A_inv) of this thread's two staged rows
      const int tid = wave * 64 + lane_now(), lrow = tid >> 3, slot = tid & 7;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int R = lrow + PROWS * p, s = R >> 4;
        int gi = (s >> 1) + 4 * (s & 1);
        gi = gi < ng ? gi : ng - 1;  // a short last tile: the missing groups alias its last one (their accumulators stay zero)
        const long grow = 16L * (long)group_of(r1, gi) + (R & 15);
        aoffb[p] = (unsigned)((grow * a.lda + ((slot ^ (lrow & 2)) << 1)) * 8);
      }
    }
#pragma unroll
    for (int ai = 0; ai < NAI; ++ai) {
      int gi = wm + 4 * ai;
      gi = gi < ng ? gi : ng - 1;
      erow[ai] = __builtin_amdgcn_readfirstlane(16 * (int)group_of(r1, gi));  // wave-uniform (wm)
    }
    ksrc = (const mik_cu32_t*)(uintptr_t)(a.klist + (long)tblk * a.nK16 + g0);
    Bgu = uniform_ptr(a.Bt + (long)tblk * MIK_BN * a.ldb);
    stage(Bgu, 16 * kfirst, 1);
  };
  if (threadIdx.x == 0) {
    const int lo = a.xoff[xcd];
    sst[0] = 0;
    sst[1] = lo;
    sst[2] = a.xoff[xcd + 1] - lo;
  }
  fetch_next();
  bool have = acquire();
  if (have) adopt();
  while (have) {
    d4 acc[NAI][4];
#pragma unroll
    for (int x = 0; x < NAI; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
    // K loop over list positions w = n - 1 .. 0 (relative to the tile's first group); position w's K tile is in buffer `buf`
    int buf = 1, w = n - 1;
    int kn = ksec;  // K tile of position w - 1
    drain();
    __syncthreads();
    fetch_next();
    for (; w >= 8; --w) {
      stage(Bgu, 16 * kn, buf ^ 1);
      int kn2 = 0;
      if (w >= 2) kn2 = list_at(w - 2);  // scalar load, in flight during this step's MFMAs
      const double* as = &sm.As[buf][0][0] + aoff;
      const double* bs = &sm.Bs[buf][0][0];
      const int bo[2] = {boff, boff_hi()};
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        double2 fa[4 * NAI], fb[4];
#pragma unroll
        for (int x = 0; x < 4 * NAI; ++x) fa[x] = *reinterpret_cast<const double2*>(as + 8 * m + 4 * x * MIK_BK);
#pragma unroll
        for (int x = 0; x < 4; ++x) fb[x] = *reinterpret_cast<const double2*>(bs + bo[m] + 16 * x * MIK_BK);
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int bi = 0; bi < 4; ++bi)
              acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[4 * ai + r].x, fb[bi].x, acc[ai][bi][r], 0, 0, 0);
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int bi = 0; bi < 4; ++bi)
              acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[4 * ai + r].y, fb[bi].y, acc[ai][bi][r], 0, 0, 0);
      }
      drain();
      __syncthreads();
      buf ^= 1;
      kn = kn2;
    }
    // the tile's own groups: position w's K tile meets the groups of positions <= w; a group's accumulators are doubled when the
    // loop reaches its own 16 x 16 square (everything above it counts twice)
    double cs[4] = {0.0, 0.0, 0.0, 0.0};  // EPI: this lane's sums over its rows of delta_ti W_it, points wn * 64 + bi * 16 + (lane & 15)
    for (; w >= 0; --w) {
      if (w >= 1) stage(Bgu, 16 * kn, buf ^ 1);
      int kn2 = 0;
      if (w >= 2) kn2 = list_at(w - 2);
      const double* as = &sm.As[buf][0][0] + aoff;
      const double* bs = &sm.Bs[buf][0][0];
      const int bo[2] = {boff, boff_hi()};
#pragma unroll
      for (int ai = 0; ai < NAI; ++ai)
        if (w == wm + 4 * ai) {
#pragma unroll
          for (int y = 0; y < 4; ++y) acc[ai][y] *= 2.0;
        }
      if (w >= wm) {
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          double2 fb[4];
#pragma unroll
          for (int x = 0; x < 4; ++x) fb[x] = *reinterpret_cast<const double2*>(bs + bo[m] + 16 * x * MIK_BK);
#pragma unroll
          for (int ai = 0; ai < NAI; ++ai)
            if (w >= wm + 4 * ai) {
              double2 fa[4];
#pragma unroll
              for (int r = 0; r < 4; ++r) fa[r] = *reinterpret_cast<const double2*>(as + 8 * m + 4 * (4 * ai + r) * MIK_BK);
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int bi = 0; bi < 4; ++bi)
                  acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[r].x, fb[bi].x, acc[ai][bi][r], 0, 0, 0);
#pragma unroll
              for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int bi = 0; bi < 4; ++bi)
                  acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[r].y, fb[bi].y, acc[ai][bi][r], 0, 0, 0);
            }
        }
      }
      if (EPI) {
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
          if (w == wm + 4 * ai) {  // the group's square was its last K tile: W is final, and delta of its rows is this step's B tile
            const int ln = lane_now(), lq2 = ln >> 4, lc2 = ln & 15;
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
              const int pnt = wn * 64 + bi * 16 + lc2, sw = (pnt >> 1) & 7;  // B image: element (point, k) in slot (k >> 1) ^ sw of its row
              const double* brow = bs + pnt * MIK_BK;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int k = 4 * r + lq2;
                cs[bi] += brow[(((k >> 1) ^ sw) << 1) | (k & 1)] * acc[ai][bi][r];
              }
            }
          }
      }
      drain();
      __syncthreads();
      buf ^= 1;
      kn = kn2;
    }
    // the next tile: record -> LDS (one barrier), its first K tile on the way to buffer 1 while this tile's epilogue runs
    const int t0 = tblk * MIK_BN, rp = rpos;
    int er[NAI];
#pragma unroll
    for (int ai = 0; ai < NAI; ++ai) er[ai] = erow[ai];
    have = acquire();
    if (have) adopt();
    // epilogue (k_contract's): part[r][t] = sum over this tile's rows of delta_ti W_it
    const int lane = lane_now(), lq = lane >> 4, lc = lane & 15;
    if (EPI) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) {
        cs[bi] += __shfl_xor(cs[bi], 16);
        cs[bi] += __shfl_xor(cs[bi], 32);
      }
    } else {
#pragma unroll
    for (int bp = 0; bp < 2; ++bp) {
      double bv[2][4 * NAI];
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const long t = t0 + wn * 64 + (2 * bp + b2) * 16 + lc;
        const double* brow = a.Bt + t * a.ldb + lq;
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[b2][ai * 4 + r] = brow[er[ai] + 4 * r];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int bi = 2 * bp + b2;
        double s = 0.0;
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r) s += bv[b2][ai * 4 + r] * acc[ai][bi][r];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        cs[bi] = s;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    }
    double* red = &sm.As[0][0][0];  // the K loop ended with a barrier; buffer 1 is being filled for the next tile
    if (lq == 0) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) red[wm * 128 + wn * 64 + bi * 16 + lc] = cs[bi];
    }
    __syncthreads();
    if (wave < 2) {
      const int c = wave * 64 + lane;
      double v = 0.0;
#pragma unroll
      for (int x = 0; x < NWM; ++x) v += red[x * 128 + c];
      a.part[(long)rp * a.palloc + t0 + c] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Point order of the range-aware contraction (round 4, second session; option "sort_points").  The contraction's unit is a block of
// 128 consecutive points; what it costs grows with the SQUARE of the number of stations within range of any of them, so a block
// should be a compact patch: a row segment of a 3-D grid (128 of 200 cells) sees most of the domain, a shuffled point list all of
// it.  The points of every launch (one chunk: a segment of the point list) are therefore put in Hilbert-curve order among
// themselves: perm[s] = index of the point at sorted position s, s and perm[s] in the same chunk -- so a launch still produces a
// contiguous range of results and its copy to the host still overlaps the next launch.  k_sp_cand / k_rhs<SP> / k_ss_reduce_sp
// read coordinates and write z, sigma^2 through perm; nothing else knows.  The sort: 2 x 10-bit (3-D: 3 x 6-bit) Hilbert keys
// relative to the segment's bounding box (cubic cells), a stable LSD radix sort with 10-bit digits in two passes, segments side
// by side in every launch (k_ps_bbox, k_ps_keys, then k_ps_hist / k_ps_scan / k_ps_scatter per pass).  Stable + keys that only
// depend on the coordinates = the same order on every device, run and rank.
// ------------------------------------------------------------------------------------------------

// Hilbert-curve index of a lattice point (Skilling, "Programming the Hilbert curve", AIP Conf. Proc. 707 (2004): axes ->
// transposed index, in place; then the bits are interleaved, X[0] first).  n axes, b bits each.  (Host: the station order.)
__host__ __device__ inline uint64_t hilbert_key(uint32_t* X, int n, int b) {
  const uint32_t Mtop = 1u << (b - 1);
  for (uint32_t Q = Mtop; Q > 1; Q >>= 1) {
    const uint32_t P = Q - 1;
    for (int i = 0; i < n; ++i) {
      if (X[i] & Q) X[0] ^= P;
      else {
        const uint32_t t = (X[0] ^ X[i]) & P;
        X[0] ^= t;
        X[i] ^= t;
      }
    }
  }
  for (int i = 1; i < n; ++i) X[i] ^= X[i - 1];
  uint32_t t = 0;
  for (uint32_t Q = Mtop; Q > 1; Q >>= 1)
    if (X[n - 1] & Q) t ^= Q - 1;
  for (int i = 0; i < n; ++i) X[i] ^= t;
  uint64_t key = 0;
  for (int bit = b - 1; bit >= 0; --bit)
    for (int i = 0; i < n; ++i) key = (key << 1) | ((X[i] >> bit) & 1u);
  return key;
}

#define MIK_PS_DB 10                 // digit bits of the radix sort
#define MIK_PS_TILE 4096             // keys per block of the histogram / scatter kernels (4 wavefronts x 1024 consecutive keys)
__host__ __device__ inline int ps_bits(int ndim) { return ndim == 3 ? 6 : 10; }  // per axis: 18- / 20-bit keys = two digits

// box[seg] = {lo x, lo y, lo z, scale}: bounding box of segment seg = points [seg chunk, min(npt, (seg + 1) chunk)), scale = lattice
// cells per unit length (one scale for all axes: cubic cells; 0 for a degenerate or non-finite extent)
__global__ void __launch_bounds__(1024) k_ps_bbox(const double* __restrict__ px, const double* __restrict__ py,
                                                  const double* __restrict__ pz, long npt, long chunk, int bits,
                                                  double* __restrict__ box) {
  __shared__ double red[6][16];
  const long lo = (long)blockIdx.x * chunk, hi = (lo + chunk < npt) ? lo + chunk : npt;
  double mn[3] = {1e300, 1e300, 1e300}, mx[3] = {-1e300, -1e300, -1e300};
  for (long t = lo + threadIdx.x; t < hi; t += 1024) {
    const double c[3] = {px[t], py[t], pz ? pz[t] : 0.0};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      mn[d] = fmin(mn[d], c[d]);  // (fmin / fmax drop a NaN coordinate)
      mx[d] = fmax(mx[d], c[d]);
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    for (int o = 32; o > 0; o >>= 1) {
      mn[d] = fmin(mn[d], __shfl_xor(mn[d], o));
      mx[d] = fmax(mx[d], __shfl_xor(mx[d], o));
    }
    if (lane == 0) {
      red[d][wave] = mn[d];
      red[3 + d][wave] = mx[d];
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ext = 0.0;
    for (int d = 0; d < 3; ++d) {
      double a = 1e300, b = -1e300;
      for (int w = 0; w < 16; ++w) {
        a = fmin(a, red[d][w]);
        b = fmax(b, red[3 + d][w]);
      }
      box[4 * blockIdx.x + d] = a;
      ext = fmax(ext, b - a);
    }
    box[4 * blockIdx.x + 3] = (ext > 0.0 && ext < 1e300) ? (double)((1u << bits) - 1) / ext : 0.0;
  }
}

__global__ void __launch_bounds__(256) k_ps_keys(const double* __restrict__ px, const double* __restrict__ py,
                                                 const double* __restrict__ pz, long npt, long chunk, int ndim, int bits,
                                                 const double* __restrict__ box, unsigned* __restrict__ key,
                                                 unsigned* __restrict__ idx) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= npt) return;
  const double* bx = box + 4 * (t / chunk);
  const double c[3] = {px[t], py[t], (ndim == 3) ? pz[t] : 0.0};
  const double top = (double)((1u << bits) - 1);
  uint32_t X[3] = {0u, 0u, 0u};
  for (int d = 0; d < ndim; ++d) {
    const double q = (c[d] - bx[d]) * bx[3];
    X[d] = (uint32_t)fmin(top, fmax(0.0, (q == q) ? q : 0.0));
  }
  key[t] = (unsigned)hilbert_key(X, ndim, bits);
  idx[t] = (unsigned)t;
}

// digit counts of every block of MIK_PS_TILE keys: table[(seg << DB | digit) * bps + block of the segment]
__global__ void __launch_bounds__(256) k_ps_hist(const unsigned* __restrict__ key, long npt, long chunk, int bps, int shift,
                                                 unsigned* __restrict__ table) {
  __shared__ unsigned h[1 << MIK_PS_DB];
  const int seg = blockIdx.x / bps, b = blockIdx.x % bps;
  const long send = ((long)(seg + 1) * chunk < npt) ? (long)(seg + 1) * chunk : npt;
  const long lo = (long)seg * chunk + (long)b * MIK_PS_TILE, hi = (lo + MIK_PS_TILE < send) ? lo + MIK_PS_TILE : send;
  for (int d = threadIdx.x; d < (1 << MIK_PS_DB); d += 256) h[d] = 0u;
  __syncthreads();
  for (long t = lo + threadIdx.x; t < hi; t += 256) atomicAdd(&h[(key[t] >> shift) & ((1u << MIK_PS_DB) - 1u)], 1u);
  __syncthreads();
  for (int d = threadIdx.x; d < (1 << MIK_PS_DB); d += 256) table[(((long)seg << MIK_PS_DB) | d) * bps + b] = h[d];
}

// exclusive scan of a segment's table (digit major, block minor): one block per segment, thread d owns digit d's row
__global__ void __launch_bounds__(1 << MIK_PS_DB) k_ps_scan(unsigned* __restrict__ table, int bps) {
  __shared__ unsigned wsum[(1 << MIK_PS_DB) / 64];
  unsigned* row = table + (((long)blockIdx.x << MIK_PS_DB) | threadIdx.x) * bps;
  unsigned tot = 0u;
  for (int b = 0; b < bps; ++b) tot += row[b];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned inc = tot;  // inclusive scan over the digits: within the wavefront, then over the wavefronts
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned v = __shfl_up(inc, o);
    if (lane >= o) inc += v;
  }
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  unsigned base = 0u;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  unsigned run = base + inc - tot;
  for (int b = 0; b < bps; ++b) {
    const unsigned c = row[b];
    row[b] = run;
    run += c;
  }
}

// stable scatter of one pass: wavefront w of a block owns the block's keys [1024 w, 1024 w + 1024) and walks them 64 at a time
__global__ void __launch_bounds__(256) k_ps_scatter(const unsigned* __restrict__ key, const unsigned* __restrict__ idx, long npt,
                                                    long chunk, int bps, int shift, const unsigned* __restrict__ table,
                                                    unsigned* __restrict__ key_out, unsigned* __restrict__ idx_out) {
  __shared__ unsigned wh[4][1 << MIK_PS_DB];
  const int seg = blockIdx.x / bps, b = blockIdx.x % bps;
  const long send = ((long)(seg + 1) * chunk < npt) ? (long)(seg + 1) * chunk : npt;
  const long lo = (long)seg * chunk + (long)b * MIK_PS_TILE, hi = (lo + MIK_PS_TILE < send) ? lo + MIK_PS_TILE : send;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long wlo = lo + 1024L * wave, whi = (wlo + 1024 < hi) ? wlo + 1024 : hi;
  const unsigned dmask = (1u << MIK_PS_DB) - 1u;
  for (int d = threadIdx.x; d < 4 * (1 << MIK_PS_DB); d += 256) (&wh[0][0])[d] = 0u;
  __syncthreads();
  for (long t = wlo + lane; t < whi; t += 64) atomicAdd(&wh[wave][(key[t] >> shift) & dmask], 1u);
  __syncthreads();
  for (int d = threadIdx.x; d < (1 << MIK_PS_DB); d += 256) {  // counts -> first output position of every (wavefront, digit)
    unsigned base = table[(((long)seg << MIK_PS_DB) | d) * bps + b];
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const unsigned c = wh[w][d];
      wh[w][d] = base;
      base += c;
    }
  }
  __syncthreads();
  volatile unsigned* run = &wh[wave][0];
  const long out0 = (long)seg * chunk;
  for (long t0 = wlo; t0 < whi; t0 += 64) {
    const long t = t0 + lane;
    const bool valid = t < whi;
    const unsigned k = valid ? key[t] : 0u, d = (k >> shift) & dmask;
    unsigned long long same = __ballot(valid);  // lanes with this lane's digit
#pragma unroll
    for (int bit = 0; bit < MIK_PS_DB; ++bit) {
      const bool on = (d >> bit) & 1u;
      const unsigned long long m = __ballot(on);
      same &= on ? m : ~m;
    }
    const int rank = __popcll(same & ((1ULL << lane) - 1ULL));
    const unsigned old = valid ? run[d] : 0u;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == 0) run[d] = old + (unsigned)__popcll(same);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      key_out[out0 + old + rank] = k;
      idx_out[out0 + old + rank] = idx[t];
    }
  }
}

// coordinates in sorted order / results back in the caller's order (moving window over sorted points: mikrige.hip, one_predict_mw)
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

// ------------------------------------------------------------------------------------------------
// K2: block Gauss-Jordan inverse, block size 128.  For diagonal block K (rows/cols k0..k0+127):
//   Dinv = T_KK^-1 (k_diag_inv) ; Cold = T[:,K] ; Cnew = -Cold.Dinv ; Rt = (Dinv.T[K,:])^T
//   T_ij -= Cold_i . Rt_j^T (i,j not in K) ; T[K,:] = Rt^T ; T[:,K] = Cnew ; T_KK = Dinv
// After all blocks T = (P.A)^-1.  On the symmetric (shifted, unpivoted) path Rt = -sigma_j * Cnew_j
// with sigma_j = -1 for already-swept column blocks and +1 otherwise, so no transposes are needed.
// ------------------------------------------------------------------------------------------------

// The sweep's flag buffer (ints, zeroed before every inverse):
//   [0]                      pivot status bits (1 = zero / non-finite pivot, 2 = non-positive pivot inside the station block)
//   [MIK_F_START + kb]       diagonal inverse kb has STARTED        (relaxed: a scheduling hint, see k_gate)
//   [MIK_F_DDONE + kb]       diagonal inverse kb has FINISHED       (release; its Dinv / DinvT are visible to an acquire)
//   [MIK_F_UCNT + kb]        finished blocks of the update of step kb (release each)
//   [MIK_F_ERR]              a bounded wait below ran out (never in a healthy run; the host turns it into an error)
// The early-diagonal schedule orders its two streams through these instead of cross-stream events: a satisfied
// hipStreamWaitEvent still costs ~12 us of barrier-packet latency per step and stream (profiles/r02_inverse_timeline.txt).
#define MIK_F_STRIDE 4096  // block columns a sweep can have (N x N matrices end long before 524 288 stations)
#define MIK_F_START 1
#define MIK_F_DDONE (1 + MIK_F_STRIDE)
#define MIK_F_UCNT (1 + 2 * MIK_F_STRIDE)
#define MIK_F_ERR (1 + 3 * MIK_F_STRIDE)
#define MIK_F_INTS (2 + 3 * MIK_F_STRIDE)
#define MIK_WAIT_POLLS 4000000  // x (s_sleep 8 + one L2 round trip) > 1 s: only a lost kernel gets there

__device__ __forceinline__ void diag_started(int* flag, int k0) {
  if (threadIdx.x == 0) __hip_atomic_store(flag + MIK_F_START + k0 / 128, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// last action of a diagonal inverse: publish Dinv / DinvT (every thread's stores, through the barrier) and raise the flag
__device__ __forceinline__ void diag_done(int* flag, int k0) {
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(flag + MIK_F_DDONE + k0 / 128, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// one thread waits until flag[idx] >= expect (acquire); false after MIK_WAIT_POLLS polls (and MIK_F_ERR is raised)
__device__ __forceinline__ bool flag_wait_ge(int* flag, int idx, int expect) {
  for (int i = 0; i < MIK_WAIT_POLLS; ++i) {
    if (__hip_atomic_load(flag + idx, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= expect) return true;
    __builtin_amdgcn_s_sleep(8);
  }
  __hip_atomic_store(flag + MIK_F_ERR, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return false;
}
__global__ void k_gate(const int* __restrict__ flag, int idx, int max_polls) {
  for (int i = 0; i < max_polls; ++i) {  // bounded: a late chain only costs this kernel's time, never a hang
    if (__hip_atomic_load(flag + MIK_F_START + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    __builtin_amdgcn_s_sleep(8);
  }
}
// a dependency, not a hint: the kernels behind this one on its stream read what the counted / flagged producers wrote
__global__ void k_wait_ge(int* __restrict__ flag, int idx, int expect) { (void)flag_wait_ge(flag, idx, expect); }

// Out[i][n] = alpha * sum_m A[i][m] * Bt[n][m],  i over Mp rows, n < 128, m < 128 (one tile column)
// NAI: a block does 32 * NAI rows (4 waves as 2 x 2, wave tile 16 NAI x 64).  NAI = 4 is one 128 x 128 tile per block: 22 us, a
// CU's MFMA rate, whatever Mp is; NAI = 1 (round 3, the sweep's default) spreads the same accumulation streams over 4 x the
// blocks -- the panel kernel sits on the update stream's critical path once per step.  Same k order per entry: same bits.
template <int NAI = 4>
__global__ void __launch_bounds__(256, 2)
k_panel(const double* __restrict__ A, long lda, const double* __restrict__ Bt, double alpha,
        double* __restrict__ Out, double* __restrict__ RtOut = nullptr, int k0 = 0, int blk0 = 0, int orow = 0,
        int* __restrict__ flag = nullptr, int wait_diag = -1, int gate_diag = -1) {
  // RtOut (symmetric sweep): also Rt[i][:] = -sigma_i Out[i][:], sigma_i = -1 for row blocks already swept
  // blk0 / orow (early-diagonal chain): start at row block blk0 and store row i at Out / RtOut row i - orow (a one-block launch
  // that leaves the 128 panel rows of one block in a 128 x 128 scratch)
  // flag (early-diagonal schedule): wait_diag >= 0 -- Bt is the DinvT of diagonal inverse wait_diag, running on the other
  // stream: wait for its flag before touching it; gate_diag >= 0 -- block 0 leaves only when diagonal inverse gate_diag has
  // started (k_gate's hint without its launch: the update behind this kernel then finds that inverse already on its CU)
  constexpr int BMR = 32 * NAI;  // rows per block
  __shared__ GemmSmemT<BMR> sm;
  if (flag && wait_diag >= 0) {
    if (threadIdx.x == 0) (void)flag_wait_ge(flag, MIK_F_DDONE + wait_diag, 1);
    __syncthreads();
  }
  const int i0 = blockIdx.x * BMR + blk0 * MIK_BM;
  d4 acc[NAI][4];
#pragma unroll
  for (int x = 0; x < NAI; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
  gemm_core<NAI, 0, BMR>(A + (long)i0 * lda, lda, Bt, 128, 0, 128, acc, sm);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, lq = lane >> 4, lc = lane & 15;
#pragma unroll
  for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wm * (16 * NAI) + ai * 16 + lq + 4 * r;
        const int n = wn * 64 + bi * 16 + lc;
        const double v = alpha * acc[ai][bi][r];
        Out[(long)(i - orow) * 128 + n] = v;
        if (RtOut) RtOut[(long)(i - orow) * 128 + n] = (i < k0) ? v : -v;
      }
  if (flag && gate_diag >= 0 && blockIdx.x == 0 && threadIdx.x == 0) {
    for (int i = 0; i < 20000; ++i) {
      if (__hip_atomic_load(flag + MIK_F_START + gate_diag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
      __builtin_amdgcn_s_sleep(8);
    }
  }
}

// trailing update + panel write-back, one 128x128 tile per block.  part = 0: every tile; part = 1: only block column
// `col` (nblk blocks; the look-ahead launch that frees the next panel early); part = 2: everything but block column `col`.
// SYM (unpivoted sweep): the matrix stays symmetric up to a known sign (T_ab = -T_ba^T when exactly one of the blocks
// a, b has been swept), so only the UPPER block triangle i <= j is maintained (half the tiles): part 0 = all upper tiles,
// part 1 = block column `col` (i <= col) and block row `col` (j >= col) -- what the next panel chain reads --, part 2 =
// the upper tiles outside those.
// NAI = 4: 4 waves per block, wave tile 64 x 64 (228 VGPRs, 2 waves per SIMD); NAI = 2 (round 3): 8 waves, wave tile 32 x 64
// (<= 128 VGPRs, 4 waves per SIMD to cover the short K loop and the read-modify-write epilogue).  Same accumulation order per
// entry: bit-identical results.
// register sets of the read-modify-write epilogue: two for the 4-wave form; the 8-wave form (128-VGPR budget) keeps ONE -- with two
// it spills 25 registers and the inverse is 12-14 % slower (N=5000 4.33 -> 4.90 ms; profiles/r03_k2_panel_stream_ab.txt)
#ifndef MIK_UPD_NTV
#define MIK_UPD_NTV(NAI) ((NAI) == 4 ? 2 : 1)
#endif
template <bool SYM, int NAI = 4>
__global__ void __launch_bounds__(64 * 2 * (8 / NAI), NAI == 2 ? 4 : 2)
k_update(double* __restrict__ T, long ld, int nblk, int kb, const double* __restrict__ Cold,
         const double* __restrict__ Cnew, const double* __restrict__ Rt, const double* __restrict__ Dinv, int part, int col,
         double* __restrict__ Pout, double* __restrict__ Dcopy = nullptr, int* __restrict__ done_cnt = nullptr,
         const int2* __restrict__ tilemap = nullptr, int atomic_rmw = 0) {
  // atomic_rmw (round 3): a tile that only has to become T - C R^T (no panel copy, no diagonal copy) sends its 128 x 128 products
  // to memory as fp64 atomic adds of -acc (global_atomic_add_f64, no return value) instead of load / subtract / store: the
  // read-modify-write then happens in the L2 while the wavefronts are already in the next tile's K loop -- the epilogue's memory
  // latency was not overlapped with anything before (the two resident blocks of a CU run their phases in step).  T + (-x) rounds
  // exactly like T - x and every entry receives one update per launch: same bits.
  // tilemap (nullable; round 3): position -> (iblk, jblk) of parts 0 / 2 / 4, written by the host (update_tile_map): the tiles in
  // the order of 8 x 8 super-blocks, so that the ~64 tiles an XCD works on at a time share 8 + 8 operand panels (2 MB of its 4 MB
  // L2) instead of a whole block column's worth (one C panel per tile: 8 MB at N = 8000, re-fetched over the fabric every column)
  // Pout (nullable): the updated block column `col` is ALSO written as the next step's column panel
  // P[row][0..127] (what k_copy_panel / k_copy_panel_sym would read back out of T: tiles of the block row `col` go in transposed),
  // so that the next panel chain starts with the diagonal inverse instead of a copy kernel.
  // Dcopy (nullable): the updated diagonal tile (col + 1, col + 1) is also left there (128 x 128): the early-diagonal chain
  // builds the diagonal block after next from it without touching T.
  // done_cnt (nullable): every block of the launch adds one when its stores are out (release): the other stream waits for
  // gridDim.x of them instead of for an event
  __shared__ GemmSmem sm;
  auto finish = [&]() {
    if (done_cnt) {
      __syncthreads();
      if (threadIdx.x == 0) __hip_atomic_fetch_add(done_cnt, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  // part = 3 / 4 (round 3, the panel stream of the sweep): part 1 plus the diagonal tile (col + 1, col + 1) as block `nblk` of
  // the launch -- everything the next panel kernel AND the chain of the diagonal inverse after next read (Pout, Dcopy) -- /
  // part 2 without that tile.
  int iblk, jblk;
  if (part == 3 && (int)blockIdx.x == nblk) {
    if (col + 1 >= nblk) return finish();
    iblk = jblk = col + 1;
  } else if (part == 1 || part == 3) {
    if ((int)blockIdx.x >= nblk) return finish();
    if (SYM && (int)blockIdx.x > col) {
      iblk = col;
      jblk = blockIdx.x;
    } else {
      iblk = blockIdx.x;
      jblk = col;
    }
  } else if (tilemap) {
    const long L = xcd_tile(SYM ? (long)nblk * (nblk + 1) / 2 : (long)nblk * nblk);
    if (L < 0) return finish();
    const int2 ij = tilemap[L];
    iblk = ij.x;
    jblk = ij.y;
    if ((part == 2 || part == 4) && (jblk == col || (SYM && iblk == col))) return finish();
    if (part == 4 && iblk == col + 1 && jblk == col + 1) return finish();
  } else if (SYM) {
    // (atomic_rmw bit 1, option "update_rev": odd steps walk every XCD's tile range from its end -- the whole upper triangle is streamed
    // once per step, cyclically; a memory-side cache smaller than it keeps nothing of a cyclic stream, but most of a back-and-forth one)
    const long L = ((atomic_rmw & 2) && (kb & 1)) ? xcd_tile_rev((long)nblk * (nblk + 1) / 2) : xcd_tile((long)nblk * (nblk + 1) / 2);
    if (L < 0) return finish();
    jblk = (int)((sqrt(8.0 * (double)L + 1.0) - 1.0) * 0.5);
    while ((long)jblk * (jblk + 1) / 2 > L) --jblk;            // guard the float estimate
    while ((long)(jblk + 1) * (jblk + 2) / 2 <= L) ++jblk;
    iblk = (int)(L - (long)jblk * (jblk + 1) / 2);             // i <= j: the upper block triangle
    if ((part == 2 || part == 4) && (iblk == col || jblk == col)) return finish();
    if (part == 4 && iblk == col + 1 && jblk == col + 1) return finish();
  } else {
    const long L = xcd_tile((long)nblk * nblk);
    if (L < 0) return finish();
    iblk = (int)(L / nblk);
    jblk = (int)(L % nblk);
    if ((part == 2 || part == 4) && jblk == col) return finish();
    if (part == 4 && iblk == col + 1 && jblk == col + 1) return finish();
  }
  const int i0 = iblk * MIK_BM, j0 = jblk * MIK_BN, k0 = kb * 128;
  const bool ptrans = SYM && iblk == col && jblk != col;  // a tile of the block ROW col: panel rows = its columns
  double* P = (jblk == col || (SYM && iblk == col)) ? Pout : nullptr;
  double* DC = (iblk == col + 1 && jblk == col + 1) ? Dcopy : nullptr;
  if (iblk == kb || jblk == kb) {
    for (int e = threadIdx.x; e < 128 * 128; e += 64 * 2 * (8 / NAI)) {
      const int r = e >> 7, c = e & 127;
      double v;
      if (iblk == kb && jblk == kb) v = Dinv[e];
      else if (jblk == kb) v = Cnew[(long)(i0 + r) * 128 + c];
      else v = Rt[(long)(j0 + c) * 128 + r];
      T[(long)(i0 + r) * ld + j0 + c] = v;
      if (P) {
        if (ptrans) P[(long)(j0 + c) * 128 + r] = v;
        else P[(long)(i0 + r) * 128 + c] = v;
      }
    }
    return finish();
  }
  d4 acc[NAI][4];
#pragma unroll
  for (int x = 0; x < NAI; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
  gemm_core<NAI>(Cold + (long)i0 * 128, 128, Rt + (long)j0 * 128, 128, 0, 128, acc, sm);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, lq = lane >> 4, lc = lane & 15;
  constexpr int WR = 16 * NAI;  // rows of the wave tile
  // read-modify-write in batches of 16 independent loads, the NEXT batch's loads in flight while this one is subtracted and
  // stored (two register sets): the epilogue pays the memory latency once, not four times
  constexpr int NTV = MIK_UPD_NTV(NAI);  // register sets of the epilogue (the 8-wave form has a 128-VGPR budget for 4 waves per SIMD)
  double tv[NTV][4][4];
  auto tile_ptr = [&](int ai) { return T + (long)(i0 + wm * WR + ai * 16 + lq) * ld + j0 + wn * 64 + lc; };
  if ((atomic_rmw & 1) && !P && !DC) {  // block-uniform
#pragma unroll
    for (int ai = 0; ai < NAI; ++ai) {
      double* tp = tile_ptr(ai);
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          (void)__hip_atomic_fetch_add(tp + (long)(4 * r) * ld + bi * 16, -acc[ai][bi][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return finish();
  }
  auto load_batch = [&](int ai, double (&dst)[4][4]) {
    const double* tp = tile_ptr(ai);
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[bi][r] = tp[(long)(4 * r) * ld + bi * 16];
  };
  load_batch(0, tv[0]);
#pragma unroll
  for (int ai = 0; ai < NAI; ++ai) {
    if (NTV == 1 && ai > 0) load_batch(ai, tv[0]);
    if (NTV == 2 && ai + 1 < NAI) load_batch(ai + 1, tv[(ai + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    double* tp = tile_ptr(ai);
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const double v = tv[NTV == 2 ? (ai & 1) : 0][bi][r] - acc[ai][bi][r];
        tp[(long)(4 * r) * ld + bi * 16] = v;
        if (P) {
          const int row = wm * WR + ai * 16 + lq + 4 * r, cc = wn * 64 + bi * 16 + lc;  // position inside the tile
          if (ptrans) P[(long)(j0 + cc) * 128 + row] = v;
          else P[(long)(i0 + row) * 128 + cc] = v;
        }
        if (DC) DC[(wm * WR + ai * 16 + lq + 4 * r) * 128 + wn * 64 + bi * 16 + lc] = v;
      }
    __builtin_amdgcn_sched_barrier(0);
  }
  finish();
}

// Early-diagonal chain: the diagonal block kb + 1 as step kb's update will leave it, Dnext = Dsrc - Cb . Rb^T, from the 128 panel
// rows of that block alone (Cb = rows of the column panel, Rb = the matching rows of R^T, see k_panel's blk0) -- the same tile
// loop, operands and subtraction as k_update uses for this tile, hence the same bits.  One block.
__global__ void __launch_bounds__(256, 2)
k_next_diag(const double* __restrict__ Dsrc, long ldsrc, const double* __restrict__ Cb, const double* __restrict__ Rb,
            double* __restrict__ Dnext) {
  __shared__ GemmSmem sm;
  d4 acc[4][4];
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
  gemm_core<4>(Cb, 128, Rb, 128, 0, 128, acc, sm);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, lq = lane >> 4, lc = lane & 15;
#pragma unroll
  for (int ai = 0; ai < 4; ++ai)
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + ai * 16 + lq + 4 * r, cc = wn * 64 + bi * 16 + lc;
        Dnext[row * 128 + cc] = Dsrc[(long)row * ldsrc + cc] - acc[ai][bi][r];
      }
}

// One 128 x 128 x 128 product C = A . Bt^T spread over the chip: 256 wavefronts (64 blocks), each ONE accumulator stream of
// gemm_core's tile loop -- 4 rows x 16 columns, v_mfma_f64_4x4x4_4b, K tiles of 16 from the top down, within a tile the k
// quadruples {8m + 2kq + h} in the order (m, h) = (0,0) (0,1) (1,0) (1,1) -- so every entry is accumulated in exactly the order
// k_panel / k_update use and comes out with the same bits, but in ~4 us instead of the 22 us one 256-thread block needs for
// the tile (a CU's MFMA rate).  All 32 operand fragments of a lane are loaded up front (one memory latency).
//   MODE 0: Out = -(alpha * acc)  (R^T rows of a block below the pivot block, what k_panel's RtOut holds for them)
//   MODE 1: Out = Dsrc - acc      (k_update's tile)
// A, Bt, Out: 128 x 128, row stride 128; Dsrc: row stride ldsrc.
template <int MODE>
__global__ void __launch_bounds__(256) k_gemm128(const double* __restrict__ A, const double* __restrict__ Bt, double alpha,
                                                 const double* __restrict__ Dsrc, long ldsrc, double* __restrict__ Out) {
  const int lane = threadIdx.x & 63, w = blockIdx.x * 4 + (threadIdx.x >> 6);  // 0 .. 255
  const int R = w >> 3, Cg = w & 7, kq = lane >> 4;
  const double* ap = A + (long)(4 * R + (lane & 3)) * 128 + 2 * kq;
  const double* bp = Bt + (long)(16 * Cg + (lane & 15)) * 128 + 2 * kq;
  double2 fa[16], fb[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {  // t = 2 * tile + m: k = 8 t + 2 kq + h
    fa[t] = *reinterpret_cast<const double2*>(ap + 8 * t);
    fb[t] = *reinterpret_cast<const double2*>(bp + 8 * t);
  }
  double acc = 0.0;
#pragma unroll
  for (int tile = 7; tile >= 0; --tile)
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      acc = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[2 * tile + m].x, fb[2 * tile + m].x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[2 * tile + m].y, fb[2 * tile + m].y, acc, 0, 0, 0);
    }
  const int row = 4 * R + (lane >> 4), col = 16 * Cg + (lane & 15);
  if (MODE == 0) {
    const double v = alpha * acc;
    Out[row * 128 + col] = -v;
  } else {
    Out[row * 128 + col] = Dsrc[(long)row * ldsrc + col] - acc;
  }
}

// The diagonal inverse is the head of the sweep's serial chain: 128 barrier-separated pivot steps, 88 us on a CU of its own and
// 120 - 200 us on a CU it shares with a trailing-update block (measured, profiles/r02_inverse_timeline.txt).  In the look-ahead
// sweep it therefore gets a CU of its own: the big trailing update of a step is held back by k_gate until the diagonal inverse
// of the next step HAS STARTED (flag[1 + block] is raised as its first action) -- it then sits on an empty CU --, and the
// inverse is launched with ~100 KB of dynamic LDS it never touches, so that no 64-KB update block can join it there.
// 1 / p for the pivots of the diagonal-block inverse: hardware reciprocal estimate + two Newton steps (5 dependent operations)
// instead of the ~35-instruction IEEE division sequence -- it sits on the serial path of every one of the 128 pivot steps.
// Within 1 ulp of the correctly rounded quotient; zero / non-finite pivots are flagged by the callers before the result is used.
__device__ __forceinline__ double pivot_recip(double p) {
  double r = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-p, r, 1.0);
  return __builtin_fma(r, e, r);
}

// 128x128 in-register Gauss-Jordan inverse of the diagonal block, one 1024-thread workgroup.
// Thread (w = wave 0..15, lane) owns rows 8w..8w+7, columns lane and lane+64.  Per elimination step
// the owners publish the pivot row and pivot column through double-buffered LDS; one barrier per step.
// flag bit0: zero / non-finite pivot (singular); bit1: non-positive pivot inside the station block
// (the shifted matrix was not positive definite -> the unpivoted path is not trustworthy).
__global__ void __launch_bounds__(1024) k_diag_inv(const double* __restrict__ T, long ld, int k0, int nspd,
                                                   double* __restrict__ Dinv, double* __restrict__ DinvT,
                                                   int* __restrict__ flag) {
  __shared__ double rowk[2][128], colk[2][128];
  diag_started(flag, k0);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  double al[8], ah[8];  // columns lane / lane+64 of this thread's 8 rows (two arrays: never indexed dynamically)
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    al[r] = T[(long)(k0 + w * 8 + r) * ld + k0 + lane];
    ah[r] = T[(long)(k0 + w * 8 + r) * ld + k0 + lane + 64];
  }
  int bad = 0;
  // k = 8*kb + kr with kr unrolled: the pivot row's owner is wave kb and its local row index kr is a
  // compile-time constant, so a[][] is only ever indexed statically (no scratch).
#pragma unroll 1
  for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
    for (int kr = 0; kr < 8; ++kr) {
      const int k = kb * 8 + kr;
      const int pb = kr & 1;
      if (kb == w) {
        rowk[pb][lane] = al[kr];
        rowk[pb][lane + 64] = ah[kr];
      }
      if (lane == (k & 63)) {
#pragma unroll
        for (int r = 0; r < 8; ++r) colk[pb][w * 8 + r] = (kb < 8) ? al[r] : ah[r];
      }
      __syncthreads();
      const double piv = rowk[pb][k];
      if (!(fabs(piv) > 1e-300) || !isfinite(piv)) bad |= 1;
      if ((k0 + k) < nspd && !(piv > 0.0)) bad |= 2;
      const double pinv = pivot_recip(piv);
      const double rk0 = rowk[pb][lane] * pinv, rk1 = rowk[pb][lane + 64] * pinv;
      const bool c0 = (lane == k), c1 = (lane + 64 == k);
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const double f = colk[pb][w * 8 + r];
        const double n0 = c0 ? -f * pinv : al[r] - f * rk0;
        const double n1 = c1 ? -f * pinv : ah[r] - f * rk1;
        const bool prow = (kb == w) && (r == kr);
        al[r] = prow ? (c0 ? pinv : rk0) : n0;
        ah[r] = prow ? (c1 ? pinv : rk1) : n1;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const int i = w * 8 + r;
    Dinv[i * 128 + lane] = al[r];
    Dinv[i * 128 + lane + 64] = ah[r];
    DinvT[lane * 128 + i] = al[r];
    DinvT[(lane + 64) * 128 + i] = ah[r];
  }
  if (bad && threadIdx.x == 0) atomicOr(flag, bad);
  diag_done(flag, k0);
}

// The same 128x128 in-place Gauss-Jordan inverse on a NT-thread workgroup laid out as a GY x GX grid with a cyclic
// (128/GY) x (128/GX) register tile per thread (rows ty + GY i, columns tx + GX j): fewer wavefronts per barrier and the
// pivot row / column indices inside a thread are compile-time constants (kb outer, unrolled).  One barrier per step.
template <int GY, int GX>
__global__ void __launch_bounds__(GY * GX) k_diag_inv_t(const double* __restrict__ T, long ld, int k0, int nspd,
                                                         double* __restrict__ Dinv, double* __restrict__ DinvT,
                                                         int* __restrict__ flag) {
  constexpr int RI = 128 / GY, CJ = 128 / GX, KBN = GY;  // steps per unrolled group
  static_assert(GY <= GX && GX % GY == 0, "row groups nest in column groups");
  // pivot row / column in OWNER-MAJOR order ([tx][j], [ty][i]): a thread's CJ + RI reads per step are contiguous (ds_read_b128)
  __shared__ double rowk[2][128], colk[2][128];
  diag_started(flag, k0);
  const int ty = threadIdx.x / GX, tx = threadIdx.x % GX;
  double a[RI][CJ];
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < CJ; ++j) a[i][j] = T[(long)(k0 + ty + GY * i) * ld + k0 + tx + GX * j];
  int bad = 0;
  // step k = GY * kb + kr: pivot row k is local row kb of the threads with ty == kr; pivot column k is local column
  // jb = k / GX (constant within the group) of the threads with tx == k % GX
#pragma unroll
  for (int kb = 0; kb < 128 / KBN; ++kb) {
    const int jb = (GY * kb) / GX, cbase = (GY * kb) % GX;  // compile-time after unrolling
#pragma unroll 1
    for (int kr = 0; kr < KBN; ++kr) {
      const int pb = kr & 1, pc = cbase + kr;  // pc = k % GX: the tx that owns pivot column k
      if (ty == kr) {
#pragma unroll
        for (int j = 0; j < CJ; ++j) rowk[pb][tx * CJ + j] = a[kb][j];
      }
      if (tx == pc) {
#pragma unroll
        for (int i = 0; i < RI; ++i) {
#pragma unroll
          for (int j = 0; j < CJ; ++j)
            if (j == jb) colk[pb][ty * RI + i] = a[i][j];
        }
      }
      __syncthreads();
      const double piv = rowk[pb][pc * CJ + jb];  // element (k, k)
      if (!(fabs(piv) > 1e-300) || !isfinite(piv)) bad |= 1;
      if ((k0 + KBN * kb + kr) < nspd && !(piv > 0.0)) bad |= 2;
      const double pinv = pivot_recip(piv);
      double rk[CJ], ck[RI];
#pragma unroll
      for (int j = 0; j < CJ; ++j) rk[j] = rowk[pb][tx * CJ + j] * pinv;
#pragma unroll
      for (int i = 0; i < RI; ++i) ck[i] = colk[pb][ty * RI + i];
      const bool prow = (ty == kr), pcol = (tx == pc);
#pragma unroll
      for (int i = 0; i < RI; ++i) {
#pragma unroll
        for (int j = 0; j < CJ; ++j) {
          double v = a[i][j] - ck[i] * rk[j];
          if (j == jb) v = pcol ? -ck[i] * pinv : v;               // pivot column: -a_ik / a_kk
          if (i == kb) v = prow ? ((j == jb && pcol) ? pinv : rk[j]) : v;  // pivot row: a_kj / a_kk, corner 1 / a_kk
          a[i][j] = v;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < RI; ++i)
#pragma unroll
    for (int j = 0; j < CJ; ++j) {
      const int r = ty + GY * i, c = tx + GX * j;
      Dinv[r * 128 + c] = a[i][j];
      DinvT[c * 128 + r] = a[i][j];
    }
  if (bad && threadIdx.x == 0) atomicOr(flag, bad);
  diag_done(flag, k0);
}

// ------------------------------------------------------------------------------------------------
// Round 3: the diagonal-block inverse BLOCKED -- 8 sub-steps of 16 pivots instead of 128 barrier-separated rank-1 steps.
// k_diag_inv_t spends half of every pivot step (~750 of 1430 cycles, profiles/r02_diag_probe.txt) in "publish the pivot row and
// column -> barrier -> read them back", 128 times.  Here the 128 x 128 block lives in the MFMA accumulator layout of gemm_core
// (4 waves as 2 x 2, wave tile 64 x 64: acc[ai][bi][r] <-> row 64 wm + 16 ai + 4 r + (lane >> 4), column 64 wn + 16 bi + (lane & 15))
// and a sub-step s (pivots 16 s .. 16 s + 15) is
//   1. the owners publish the raw column block (128 x 16) and the raw row block (16 x 128, transposed) as K tiles in LDS; barrier
//   2. the 16 x 16 diagonal sub-block is inverted by Gauss-Jordan INSIDE ONE WAVEFRONT (lane = 4 i + jq holds D[i][4 jq .. 4 jq + 3];
//      pivot row / column / pivot travel by cross-lane reads, no LDS round trip, no barrier), redundantly by all four waves (they
//      sit on four SIMDs; nothing else could run meanwhile); wave 0 leaves Dinv (and -Dinv^T) as B tiles; barrier
//   3. Cnew = -Craw . Dinv (128 x 16) and Rnew^T = Rraw^T . Dinv^T (128 x 16) on the matrix cores, 32 rows per wave; barrier
//   4. the rank-16 update  M += Cnew . Rraw  of the whole block: ONE K tile of gemm_core's loop (256 MFMAs per wave), then the
//      column block, row block and diagonal sub-block are overwritten with Cnew, Rnew, Dinv (Gauss-Jordan in place).
// The same elimination order as k_diag_inv_t (no pivoting either way), sums grouped differently: equal to rounding, not bit
// for bit.  ~86 KB of LDS (dynamic), which also keeps trailing-update blocks off this block's CU (see k_gate).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ int tile_a_idx(int row, int k) { return row * 16 + ((((k >> 1) ^ (row & 2))) << 1) + (k & 1); }
__device__ __forceinline__ int tile_b_idx(int row, int k) { return row * 16 + ((((k >> 1) ^ ((row >> 1) & 7))) << 1) + (k & 1); }
#define MIK_DIAGB_LDS_DOUBLES (2048 + 2 * 2048 + 2048 + 2048 + 256 + 256)

// ABL (tools/diag_probe only; 0 in the library): 1 = no pivot loop, 2 = no rank-16 update, 4 = no panel products, 8 = no publish /
// overwrite, 16 = no barriers -- results are then wrong, only the clock is read.
template <int ABL = 0>
__global__ void __launch_bounds__(256) k_diag_inv_b(const double* __restrict__ T, long ld, int k0, int nspd,
                                                     double* __restrict__ Dinv, double* __restrict__ DinvT,
                                                     int* __restrict__ flag) {
  extern __shared__ double diagb_lds[];
  double* const Craw = diagb_lds;          // [128][16], A swizzle: the raw column block
  double* const Rt0 = diagb_lds + 2048;    // 2 x [128][16], B swizzle: the raw row block, transposed (alternating)
  double* const Cn = diagb_lds + 6144;     // [128][16], A swizzle: Cnew
  double* const Rn = diagb_lds + 8192;     // [128][16], B swizzle: Rn[col][k] = Rnew[k][col]
  double* const Bd1 = diagb_lds + 10240;   // [16][16], B swizzle: Bd1[c][q] = -Dinv[q][c]
  double* const Bd2 = Bd1 + 256;           // [16][16], B swizzle: Bd2[k][q] =  Dinv[k][q]
  diag_started(flag, k0);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, lq = lane >> 4, lc = lane & 15;
  const int kq = lane >> 4, ia = lane & 3, jb = lane & 15;  // operand-fragment coordinates (gemm_core)
  d4 acc[4][4];
#pragma unroll
  for (int ai = 0; ai < 4; ++ai)
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[ai][bi][r] = T[(long)(k0 + wm * 64 + 16 * ai + 4 * r + lq) * ld + k0 + wn * 64 + 16 * bi + lc];
  int bad = 0;
  // broadcast inside each quad of lanes (DPP quad_perm: no LDS crossbar), and a lane's double read into SGPRs
  auto quad_bcast = [](double v, auto qc) {
    constexpr int q = decltype(qc)::value, ctrl = q | (q << 2) | (q << 4) | (q << 6);
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
  };
  auto lane_value = [](double v, int l) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
  };
#pragma unroll 1
  for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
    for (int sq = 0; sq < 4; ++sq) {  // unrolled: the accumulator registers of block column / row sq are named at compile time
      const int s = 4 * sb + sq;
      double* const Rt = Rt0 + (sq & 1) * 2048;
      // 1. publish the raw column block and the raw row block
      if (!(ABL & 8) && wn == sb) {
#pragma unroll
        for (int ai = 0; ai < 4; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r) Craw[tile_a_idx(wm * 64 + 16 * ai + 4 * r + lq, lc)] = acc[ai][sq][r];
      }
      if (!(ABL & 8) && wm == sb) {
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
          for (int r = 0; r < 4; ++r) Rt[tile_b_idx(wn * 64 + 16 * bi + lc, 4 * r + lq)] = acc[sq][bi][r];
      }
      if (!(ABL & 16)) __syncthreads();
      // 2. the 16 x 16 diagonal sub-block, inverted inside the wavefront.  The pivot of step p + 1 is known to every lane one
      // step early (three more uniform values of the current state), so its reciprocal -- five dependent operations -- is formed
      // while the cross-lane reads of step p + 1 are in flight instead of after them.
      {
        const int i = lane >> 2, jq = lane & 3;
        double a[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) a[c] = Craw[tile_a_idx(16 * s + i, 4 * jq + c)];
        auto check = [&](double piv, int p) {
          if (!(fabs(piv) > 1e-300) || !isfinite(piv)) bad |= 1;
          if ((k0 + 16 * s + p) < nspd && !(piv > 0.0)) bad |= 2;
        };
        double pinv = 0.0;
        if (!(ABL & 1)) {
          const double piv0 = lane_value(a[0], 0);
          check(piv0, 0);
          pinv = pivot_recip(piv0);
        }
#pragma unroll
        for (int p = 0; p < ((ABL & 1) ? 0 : 16); ++p) {
          const int pr = p & 3, pq = p >> 2;
          double f;  // D[i][p]
          switch (pq) {
            case 0: f = quad_bcast(a[pr], std::integral_constant<int, 0>{}); break;
            case 1: f = quad_bcast(a[pr], std::integral_constant<int, 1>{}); break;
            case 2: f = quad_bcast(a[pr], std::integral_constant<int, 2>{}); break;
            default: f = quad_bcast(a[pr], std::integral_constant<int, 3>{}); break;
          }
          double rk[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) rk[c] = __shfl(a[c], 4 * p + jq);  // D[p][4 jq + c]
          double pinv_next = 0.0;
          if (p < 15) {
            const int p1 = p + 1, r1 = p1 & 3, q1 = p1 >> 2;
            const double d11 = lane_value(a[r1], 4 * p1 + q1);  // D[p+1][p+1]
            const double d10 = lane_value(a[pr], 4 * p1 + pq);  // D[p+1][p]
            const double d01 = lane_value(a[r1], 4 * p + q1);   // D[p][p+1]
            const double pivn = __builtin_fma(-d10, d01 * pinv, d11);  // what the update below leaves at (p+1, p+1), same operations
            check(pivn, p1);
            pinv_next = pivot_recip(pivn);
          }
#pragma unroll
          for (int c = 0; c < 4; ++c) rk[c] *= pinv;
          const bool prow = (i == p), pcol = (jq == pq);
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            double v = __builtin_fma(-f, rk[c], a[c]);
            if (c == pr) v = pcol ? -f * pinv : v;                // pivot column: -a_ip / a_pp
            v = prow ? ((c == pr && pcol) ? pinv : rk[c]) : v;    // pivot row: a_pj / a_pp, corner 1 / a_pp
            a[c] = v;
          }
          pinv = pinv_next;
        }
        if (wave == 0) {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            Bd2[tile_b_idx(i, 4 * jq + c)] = a[c];
            Bd1[tile_b_idx(4 * jq + c, i)] = -a[c];
          }
        }
      }
      if (!(ABL & 16)) __syncthreads();
      // 3. Cnew (rows 32 wave ..) and Rnew^T (columns 32 wave ..): 8 groups of 4 rows each, K = 16
      if (!(ABL & 4)) {
        const int R0 = 32 * wave;
#pragma unroll
        for (int which = 0; which < 2; ++which) {
          const double* src = which ? Rt : Craw;
          const double* bd = which ? Bd2 : Bd1;
          double pc[8];
#pragma unroll
          for (int g = 0; g < 8; ++g) pc[g] = 0.0;
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            const double2 fbd = *reinterpret_cast<const double2*>(bd + tile_b_idx(jb, 8 * m + 2 * kq));
            double2 fc[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) {
              const int row = R0 + 4 * g + ia;
              fc[g] = *reinterpret_cast<const double2*>(src + (which ? tile_b_idx(row, 8 * m + 2 * kq) : tile_a_idx(row, 8 * m + 2 * kq)));
            }
#pragma unroll
            for (int g = 0; g < 8; ++g) pc[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[g].x, fbd.x, pc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < 8; ++g) pc[g] = __builtin_amdgcn_mfma_f64_4x4x4f64(fc[g].y, fbd.y, pc[g], 0, 0, 0);
          }
#pragma unroll
          for (int g = 0; g < 8; ++g) {
            const int row = R0 + 4 * g + lq;
            if (which) Rn[tile_b_idx(row, lc)] = pc[g];
            else Cn[tile_a_idx(row, lc)] = pc[g];
          }
        }
      }
      if (!(ABL & 16)) __syncthreads();
      // 4. M += Cnew . Rraw: one K tile of gemm_core's loop
#pragma unroll
      for (int m = 0; m < ((ABL & 2) ? 0 : 2); ++m) {
        double2 fa[16], fb[4];
#pragma unroll
        for (int x = 0; x < 16; ++x) fa[x] = *reinterpret_cast<const double2*>(Cn + tile_a_idx(wm * 64 + 4 * x + ia, 8 * m + 2 * kq));
#pragma unroll
        for (int x = 0; x < 4; ++x) fb[x] = *reinterpret_cast<const double2*>(Rt + tile_b_idx(wn * 64 + 16 * x + jb, 8 * m + 2 * kq));
#pragma unroll
        for (int ai = 0; ai < 4; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int bi = 0; bi < 4; ++bi)
              acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[4 * ai + r].x, fb[bi].x, acc[ai][bi][r], 0, 0, 0);
#pragma unroll
        for (int ai = 0; ai < 4; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int bi = 0; bi < 4; ++bi)
              acc[ai][bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[4 * ai + r].y, fb[bi].y, acc[ai][bi][r], 0, 0, 0);
      }
      // Gauss-Jordan in place: column block <- Cnew, row block <- Rnew, diagonal sub-block <- Dinv
      if (!(ABL & 8) && wn == sb) {
#pragma unroll
        for (int ai = 0; ai < 4; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[ai][sq][r] = Cn[tile_a_idx(wm * 64 + 16 * ai + 4 * r + lq, lc)];
      }
      if (!(ABL & 8) && wm == sb) {
#pragma unroll
        for (int bi = 0; bi < 4; ++bi)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[sq][bi][r] = Rn[tile_b_idx(wn * 64 + 16 * bi + lc, 4 * r + lq)];
        if (wn == sb) {
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[sq][sq][r] = Bd2[tile_b_idx(4 * r + lq, lc)];
        }
      }
    }
  }
#pragma unroll
  for (int ai = 0; ai < 4; ++ai)
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wm * 64 + 16 * ai + 4 * r + lq, col = wn * 64 + 16 * bi + lc;
        Dinv[row * 128 + col] = acc[ai][bi][r];
        DinvT[col * 128 + row] = acc[ai][bi][r];
      }
  if (bad && lane == 0) atomicOr(flag, bad);
  diag_done(flag, k0);
}

// Out[j][m] = T[k0+m][j]   (transpose of a 128-row panel; general path)
__global__ void __launch_bounds__(256) k_transpose_rows(const double* __restrict__ T, long ld, int k0, int Mp,
                                                        double* __restrict__ Out) {
  __shared__ double tile[64][65];
  const int j0 = blockIdx.x * 64, m0 = blockIdx.y * 64;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int m = e >> 6, j = e & 63;
    tile[m][j] = T[(long)(k0 + m0 + m) * ld + j0 + j];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int j = e >> 6, m = e & 63;
    Out[(long)(j0 + j) * 128 + m0 + m] = tile[m][j];
  }
}

// ------------------------------------------------------------------------------------------------
// Pivot search for the pivoted path (partial pivoting, LAPACK dgetf2 order) on a scratch copy of the
// column panel.  One launch per panel column c; ping-pong buffers Pin -> Pout (Mp x 128, ld 128):
//   every block first reduces the previous launch's per-block candidates to the pivot row `pr` of
//   column c, then rewrites its rows with rows (k0+c) and pr exchanged and column c eliminated from
//   the rows below k0+c, and finally emits its candidate (max |.| over active rows) for column c+1.
// Rows < k0 (already pivots of earlier blocks) and rows >= M (padding) never take part.
// ------------------------------------------------------------------------------------------------
struct PivCand {
  double v;
  int row;
  int pad;
};

__global__ void __launch_bounds__(64)
k_piv_first(const double* __restrict__ P, int k0, int M, int Mp, PivCand* __restrict__ cand) {
  // candidate of column 0 for one block of MIK_PIV_ROWS rows (same block granularity as k_piv_step)
  const int row = blockIdx.x * 32 + threadIdx.x;
  double v = -1.0;
  int r = 0x7fffffff;
  if (threadIdx.x < 32 && row >= k0 && row < M) { v = fabs(P[(long)row * 128]); r = row; }
  for (int o = 16; o > 0; o >>= 1) {
    const double v2 = __shfl_xor(v, o);
    const int r2 = __shfl_xor(r, o);
    if (v2 > v || (v2 == v && r2 < r)) { v = v2; r = r2; }
  }
  if (threadIdx.x == 0) {
    cand[blockIdx.x].v = v;
    cand[blockIdx.x].row = r;
  }
}

// One block = 32 rows of the scratch panel; thread (col = tid & 127, ty = tid >> 7) walks the block's rows
// two at a time, so every access is a coalesced 1-KiB row.  Rows < k0 + c and columns <= c are dead for the
// pivot search and are not copied.
#define MIK_PIV_ROWS 32
__global__ void __launch_bounds__(256)
k_piv_step(const double* __restrict__ Pin, double* __restrict__ Pout, int k0, int c, int M, int Mp,
           const PivCand* __restrict__ cand_in, PivCand* __restrict__ cand_out, int ncand,
           int* __restrict__ pivrow /* 128 entries of this panel */, int* __restrict__ flag) {
  __shared__ double sv[256];
  __shared__ int sr[256];
  __shared__ int s_pr;
  // 1. pivot row of column c from the candidates of the previous launch
  {
    double v = -2.0;
    int r = 0x7fffffff;
    for (int e = threadIdx.x; e < ncand; e += 256) {
      const double v2 = cand_in[e].v;
      const int r2 = cand_in[e].row;
      if (v2 > v || (v2 == v && r2 < r)) { v = v2; r = r2; }
    }
    sv[threadIdx.x] = v;
    sr[threadIdx.x] = r;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        const double v2 = sv[threadIdx.x + o];
        const int r2 = sr[threadIdx.x + o];
        if (v2 > sv[threadIdx.x] || (v2 == sv[threadIdx.x] && r2 < sr[threadIdx.x])) {
          sv[threadIdx.x] = v2;
          sr[threadIdx.x] = r2;
        }
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      int pr = sr[0];
      if (!(sv[0] > 0.0)) {  // nothing usable left in this column: singular (or only padding rows left)
        pr = k0 + c;
        if (k0 + c < M) atomicOr(flag, 1);
      }
      s_pr = pr;
      if (blockIdx.x == 0) pivrow[c] = pr;
    }
    __syncthreads();
  }
  const int pr = s_pr, kr = k0 + c;
  const int col = threadIdx.x & 127, ty = threadIdx.x >> 7;
  const double pcol = Pin[(long)pr * 128 + col];  // pivot row, this thread's column
  const double pinv = 1.0 / Pin[(long)pr * 128 + c];
  // 2. rows of this block: exchange kr <-> pr, eliminate column c from the rows below kr (columns > c only)
  double nextv = -1.0;
  int nextr = 0x7fffffff;
  const int r0 = blockIdx.x * MIK_PIV_ROWS;
  for (int rr = ty; rr < MIK_PIV_ROWS; rr += 2) {
    const int row = r0 + rr;
    if (row < kr || row >= Mp) continue;
    const int src = (row == kr) ? pr : ((row == pr) ? kr : row);
    const double x = Pin[(long)src * 128 + col];
    double y = x;
    if (row > kr && row < M) {
      const double f = Pin[(long)src * 128 + c] * pinv;  // broadcast load
      if (col > c) y = x - f * pcol;
      if (col == c + 1) {
        const double ay = fabs(y);
        if (ay > nextv) { nextv = ay; nextr = row; }  // rows ascend: the first maximum is kept
      }
    }
    if (col > c || row == kr) Pout[(long)row * 128 + col] = y;
  }
  // 3. this block's candidate for column c+1: held by the threads with col == c+1 (one per ty)
  sv[threadIdx.x] = nextv;
  sr[threadIdx.x] = nextr;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = -1.0;
    int r = 0x7fffffff;
    if (c + 1 < 128) {
      for (int t = 0; t < 2; ++t) {
        const double v2 = sv[t * 128 + c + 1];
        const int r2 = sr[t * 128 + c + 1];
        if (v2 > v || (v2 == v && r2 < r)) { v = v2; r = r2; }
      }
    }
    cand_out[blockIdx.x].v = v;
    cand_out[blockIdx.x].row = r;
  }
}

// apply the panel's 128 row interchanges (in order) to all of T; one block per 256 columns
__global__ void __launch_bounds__(256)
k_swap_rows(double* __restrict__ T, long ld, int k0, const int* __restrict__ pivrow, int Mp) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  for (int c = 0; c < 128; ++c) {
    const int pr = pivrow[c], kr = k0 + c;
    if (pr != kr) {
      const double a = T[(long)kr * ld + j], b = T[(long)pr * ld + j];
      T[(long)kr * ld + j] = b;
      T[(long)pr * ld + j] = a;
    }
  }
}

// undo the row interchanges as column interchanges in reverse order: A^-1 = (P A)^-1 P
__global__ void __launch_bounds__(256)
k_swap_cols(double* __restrict__ T, long ld, const int* __restrict__ pivall, int nswap, int Mp) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Mp) return;
  double* row = T + (long)i * ld;
  for (int s = nswap - 1; s >= 0; --s) {
    const int pr = pivall[s];
    if (pr != s) {
      const double a = row[s], b = row[pr];
      row[s] = b;
      row[pr] = a;
    }
  }
}

// Symmetric sweep: column panel of block K from the upper block triangle.  Rows at / above the block are read in place;
// rows below it (none swept yet, like K itself: plain symmetry) come from the block ROW K, P[r][c] = T[k0 + c][r], through
// an LDS transpose so that both the reads and the writes stay coalesced.  One 64-row slab per block.
__global__ void __launch_bounds__(256) k_copy_panel_sym(const double* __restrict__ T, long ld, int k0, int Mp,
                                                        double* __restrict__ P) {
  __shared__ double tile[64][65];
  const int r0 = blockIdx.x * 64;
  if (r0 < k0 + 128) {
    for (int e = threadIdx.x; e < 64 * 128; e += 256) {
      const int r = e >> 7, c = e & 127;
      P[(long)(r0 + r) * 128 + c] = T[(long)(r0 + r) * ld + k0 + c];
    }
    return;
  }
  for (int half = 0; half < 2; ++half) {  // 64 of the 128 panel columns at a time
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int c = e >> 6, r = e & 63;  // consecutive threads walk along a row of T
      tile[c][r] = T[(long)(k0 + half * 64 + c) * ld + r0 + r];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < 64 * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      P[(long)(r0 + r) * 128 + half * 64 + c] = tile[c][r];
    }
    __syncthreads();
  }
}

// after the symmetric sweep every block is swept: T is symmetric, fill the lower block triangle from the upper one
__global__ void __launch_bounds__(256) k_mirror_upper(double* __restrict__ T, long ld, int nblk64) {
  __shared__ double tile[64][65];
  const int bi = blockIdx.y, bj = blockIdx.x;  // 64 x 64 tiles; source tile (bi, bj) with bi <= bj, destination (bj, bi)
  if (bi > bj) return;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    tile[r][c] = T[(long)(bi * 64 + r) * ld + bj * 64 + c];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    if (bi < bj || c < r) T[(long)(bj * 64 + r) * ld + bi * 64 + c] = tile[c][r];
  }
}

// after a FULL sweep (or the pivoted elimination): T <- (T + T^T) / 2.  The inverse of the symmetric kriging matrix is symmetric;
// the computed one is so only up to rounding (cond . eps), and the symmetric contraction reads one triangle: on an ill-conditioned
// system (power variogram + drift terms) the two triangles differ by more than the sigma^2 bar at exact-hit points, where
// b^T X b is a difference of large terms.  A quadratic form sees only the symmetric part of X, so with the average in both
// triangles the half product equals the full one to rounding (round 3).  64 x 64 tile pairs, like k_mirror_upper.
__global__ void __launch_bounds__(256) k_symmetrize(double* __restrict__ T, long ld, int nblk64) {
  __shared__ double up[64][65], lo[64][65];
  const int bi = blockIdx.y, bj = blockIdx.x;
  if (bi > bj) return;
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    up[r][c] = T[(long)(bi * 64 + r) * ld + bj * 64 + c];
    lo[r][c] = T[(long)(bj * 64 + r) * ld + bi * 64 + c];
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 64 * 64; e += 256) {
    const int r = e >> 6, c = e & 63;
    T[(long)(bi * 64 + r) * ld + bj * 64 + c] = 0.5 * (up[r][c] + lo[c][r]);
    if (bi < bj) T[(long)(bj * 64 + r) * ld + bi * 64 + c] = 0.5 * (up[c][r] + lo[r][c]);
  }
}

// copy a column panel T[:, k0:k0+128] -> P (Mp x 128)
__global__ void __launch_bounds__(256) k_copy_panel(const double* __restrict__ T, long ld, int k0, int Mp,
                                                    double* __restrict__ P) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)Mp * 128) return;
  const long i = idx >> 7;
  const int m = (int)(idx & 127);
  P[idx] = T[i * ld + k0 + m];
}

// ------------------------------------------------------------------------------------------------
// fragment-layout self test: D = A(16x4) . B(4x16) with asymmetric integer data
// ------------------------------------------------------------------------------------------------
__global__ void k_selftest_mfma(double* out /*16x16 row-major*/) {
  const int l = threadIdx.x;
  const double a = (double)((l & 15) * 7 + (l >> 4) * 3 + 1);    // A[i=l&15][k=l>>4]
  const double b = (double)((l >> 4) * 11 + (l & 15) * 5 + 2);   // B[k=l>>4][j=l&15]
  d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
  acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}


// v_mfma_f64_4x4x4_4b_f64 as the kernels use it: A replicated over blocks, B = 4 x 16 columns
__global__ void k_selftest_mfma4(double* out /*4x16 row-major*/) {
  const int l = threadIdx.x;
  const double a = (double)((l & 3) * 7 + (l >> 4) * 3 + 1);    // A[i=l&3][k=l>>4], same for every block
  const double b = (double)((l >> 4) * 11 + (l & 15) * 5 + 2);  // B[k=l>>4][col=l&15]
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[(l >> 4) * 16 + (l & 15)] = d;                            // D[i=l>>4][col=l&15]
}

// ------------------------------------------------------------------------------------------------
// Pseudo-inverse of the kriging matrix (pseudo_inv=True: P_INV[type](a), core.py:33 -> scipy.linalg.pinv / pinvh), for
// matrices made singular by duplicated stations.  One-sided (Hestenes) Jacobi on the ROWS of the symmetric matrix:
// plane rotations W = prod J make the rows of B = W A mutually orthogonal, so A = W^T diag(sigma) Q^T with q_i = b_i/sigma_i
// and pinv(A) = sum_{sigma_i > cut} b_i^T w_i / sigma_i^2 = B^T D W, cut = M eps sigma_max (SciPy's default rtol for
// both pinv and pinvh; on a symmetric matrix the two coincide: singular values = |eigenvalues|).
//   k_jac_step  : one round of the round-robin tournament: block b rotates rows (p, q) of B and W (disjoint pairs);
//                 rows below dead2 = (0.1 M eps)^2 |A|_F^2 / M (<= a hundredth of the cut-off, squared) are left alone
//   k_rownorm2  : sigma_i^2
//   k_pinv_gemm : out = B^T diag(d) W, 64 x 64 tiles
// ------------------------------------------------------------------------------------------------
// pseudo-inverse, fast path: T[i][j] += sign * val for a short coordinate list (the projector onto the null space spanned
// by duplicated stations), and a plain row-per-wavefront mat-vec for the probes that verify the result
__global__ void __launch_bounds__(256) k_coo_add(double* __restrict__ T, long ld, const int* __restrict__ ij,
                                                 const double* __restrict__ val, int n, double sign) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < n) T[(long)ij[2 * e] * ld + ij[2 * e + 1]] += sign * val[e];
}
// T[i][i] += v for i < m
__global__ void __launch_bounds__(256) k_shift_diag(double* __restrict__ T, long ld, int m, double v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < m) T[(long)i * ld + i] += v;
}
// T += sign * sum_k n_k n_k^T over the leading m x m block; the r vectors n_k are the rows of Nv (row length ldn)
__global__ void __launch_bounds__(256) k_lowrank_add(double* __restrict__ T, long ld, int m, const double* __restrict__ Nv, long ldn, int r,
                                                     double sign) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i0 = blockIdx.y * 64;
  if (j >= m) return;
  for (int ii = threadIdx.x >> 6; ii < 64; ii += 4) {
    const int i = i0 + ii;
    if (i >= m) break;
    double s = 0.0;
    for (int k = 0; k < r; ++k) s += Nv[(long)k * ldn + i] * Nv[(long)k * ldn + j];
    T[(long)i * ld + j] += sign * s;
  }
}

__global__ void __launch_bounds__(256) k_matvec(const double* __restrict__ A, long ld, int m, const double* __restrict__ x,
                                                double* __restrict__ y) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  const double* r = A + (long)row * ld;
  double s = 0.0;
  for (int b = lane; b < m; b += 64) s += r[b] * x[b];
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) y[row] = s;
}

// three matrix-vector products in one pass over the matrix (the probe columns of verify_inverse)
__global__ void __launch_bounds__(256) k_matvec3(const double* __restrict__ A, long ld, int m, const double* __restrict__ x0,
                                                 const double* __restrict__ x1, const double* __restrict__ x2, double* __restrict__ y0,
                                                 double* __restrict__ y1, double* __restrict__ y2) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  const double* r = A + (long)row * ld;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0;
  for (int b = lane; b < m; b += 64) {
    const double a = r[b];
    s0 += a * x0[b];
    s1 += a * x1[b];
    s2 += a * x2[b];
  }
  for (int o = 32; o > 0; o >>= 1) {
    s0 += __shfl_xor(s0, o);
    s1 += __shfl_xor(s1, o);
    s2 += __shfl_xor(s2, o);
  }
  if (lane == 0) y0[row] = s0, y1[row] = s1, y2[row] = s2;
}

__global__ void __launch_bounds__(256) k_set_identity(double* __restrict__ W, long ld, int n) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)n * ld) return;
  const long r = e / ld, c = e - r * ld;
  W[e] = (r == c) ? 1.0 : 0.0;
}

__global__ void __launch_bounds__(256)
k_jac_step(double* __restrict__ B, double* __restrict__ W, long ld, int n, int m, int step, double dead2,
           unsigned long long* maxoff) {
  // tournament over m (even) players: player m-1 stays, the others rotate; round `step` pairs (step+b) with (step-b)
  const int b = blockIdx.x;
  int i = step, j = m - 1;
  if (b > 0) {
    i = (step + b) % (m - 1);
    j = (step - b + (m - 1)) % (m - 1);
  }
  const int p = i < j ? i : j, q = i < j ? j : i;
  if (q >= n) return;  // the padding player of an odd n
  double* bp = B + (long)p * ld;
  double* bq = B + (long)q * ld;
  double al = 0.0, be = 0.0, ga = 0.0;
  for (long c = threadIdx.x; c < ld; c += 256) {
    const double x = bp[c], y = bq[c];
    al += x * x;
    be += y * y;
    ga += x * y;
  }
  __shared__ double red[3][4];
  __shared__ double cs[2];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    al += __shfl_xor(al, o, 64);
    be += __shfl_xor(be, o, 64);
    ga += __shfl_xor(ga, o, 64);
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = al;
    red[1][threadIdx.x >> 6] = be;
    red[2][threadIdx.x >> 6] = ga;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    al = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    be = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    ga = red[2][0] + red[2][1] + red[2][2] + red[2][3];
    double c = 1.0, sn = 0.0;
    const double scale = sqrt(al * be);
    // rows whose norm has fallen far below the pseudo-inverse cut-off are numerically zero (the null space of a
    // rank-deficient matrix): their direction is rounding noise, rotating against them would never settle
    if (al > dead2 && be > dead2 && fabs(ga) > 1e-17 * scale) {
      const double off = fabs(ga) / scale;
      atomicMax(maxoff, (unsigned long long)__double_as_longlong(off));
      const double zeta = (be - al) / (2.0 * ga);
      const double t = ((zeta >= 0.0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
      c = 1.0 / sqrt(1.0 + t * t);
      sn = c * t;
    }
    cs[0] = c;
    cs[1] = sn;
  }
  __syncthreads();
  const double c = cs[0], sn = cs[1];
  if (sn == 0.0) return;
  double* wp = W + (long)p * ld;
  double* wq = W + (long)q * ld;
  for (long k = threadIdx.x; k < ld; k += 256) {
    const double x = bp[k], y = bq[k];
    bp[k] = c * x - sn * y;
    bq[k] = sn * x + c * y;
    const double u = wp[k], v = wq[k];
    wp[k] = c * u - sn * v;
    wq[k] = sn * u + c * v;
  }
}

// ---- BLOCK one-sided Jacobi (round 4): the general pseudo-inverse without a pass over the matrix per row pair ---------------------
// The scalar form above streams B and W once per round of the tournament, M - 1 rounds per sweep: 9.3 s at M = 4000.  Here the rows
// are taken in blocks of MIK_BJ_B = 32 (sorted by norm at the start of every sweep: de Rijk's ordering, which the Gram route needs for
// its accuracy -- scripts/prototype_block_jacobi.py, profiles/r03_block_jacobi_prototype_cpu.txt); a round pairs the blocks off, and
// for every pair X (64 rows x M)
//   k_bj_gram      G = X X^T in one pass over the 64 rows (column slices on separate workgroups),
//   k_bj_eig       if some pair of live rows is further from orthogonal than `tol`, the 64 x 64 symmetric eigenproblem
//                  G = Q diag Q^T by a two-sided cyclic Jacobi in LDS (relative accuracy on graded matrices: a norm-wise
//                  eigensolver loses the small singular values the pseudo-inverse is made of),
//   k_bj_rotate    X <- Q^T X for the rows of B and of W (second pass),
// so a sweep streams the matrix ~3 (M / 32 - 1) times instead of ~2 (M - 1) times, and pairs already orthogonal cost one pass.
// order[] = row numbers sorted by norm, padded with -1 to whole blocks (and to an even number of blocks).
#define MIK_BJ_B 32
#define MIK_BJ_LD 65  // LDS row stride of the 64 x 64 matrices (odd: rows and columns are both walked)
// the pair of blocks (or of rows) that slot `b` of round `r` of a round-robin tournament over m (even) players holds
__device__ __forceinline__ void bj_pair(int m, int r, int b, int& lo, int& hi) {
  int i = r, j = m - 1;
  if (b > 0) {
    i = (r + b) % (m - 1);
    j = (r - b + (m - 1)) % (m - 1);
  }
  lo = i < j ? i : j;
  hi = i < j ? j : i;
}
// G = X X^T of a pair's 64 rows over ONE slice of the columns (grid: pairs x slices; the slices' partial sums are added in a fixed order
// by k_bj_eig: deterministic, no atomics): 16 x 16 threads, 4 x 4 entries each, the slice staged 64 columns at a time (column-major in LDS)
__global__ void __launch_bounds__(64)
k_bj_gram(const double* __restrict__ B, long ld, int n, const int* __restrict__ order, int nb, int round, int nslice,
          double* __restrict__ Gpart) {
  // ONE wavefront per (pair, slice): 8 x 8 threads with 8 x 8 entries each -- 16 LDS reads per 64 multiply-adds (the 16 x 16 x (4 x 4)
  // form of the first version read 8 per 16 and was bound by the LDS pipe: 254 us per round at M = 4000, now ~2 x less)
  __shared__ double Xs[64 * MIK_BJ_LD];
  __shared__ int idx[64];
  const int t = threadIdx.x, ty = t >> 3, tx = t & 7;
  int bi, bj;
  bj_pair(nb, round, blockIdx.x, bi, bj);
  idx[t] = order[(t < 32 ? bi : bj) * MIK_BJ_B + (t & 31)];
  __syncthreads();
  const int per = (((n + nslice - 1) / nslice + 63) / 64) * 64;
  const int cbeg = blockIdx.y * per, cend = min(n, cbeg + per);
  double acc[8][8] = {};
  for (int c0 = cbeg; c0 < cend; c0 += 64) {
    for (int r = 0; r < 64; ++r) {  // one row per pass: 64 consecutive columns (coalesced)
      const int row = idx[r], c = c0 + t;
      Xs[t * MIK_BJ_LD + r] = (row >= 0 && c < cend) ? B[(long)row * ld + c] : 0.0;
    }
    __syncthreads();
#pragma unroll 4
    for (int c = 0; c < 64; ++c) {
      double a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = Xs[c * MIK_BJ_LD + 8 * ty + u];
        b[u] = Xs[c * MIK_BJ_LD + 8 * tx + u];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int w = 0; w < 8; ++w) acc[u][w] += a[u] * b[w];
    }
    __syncthreads();
  }
  double* go = Gpart + ((long)blockIdx.x * nslice + blockIdx.y) * 64 * 64;
#pragma unroll
  for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int w = 0; w < 8; ++w) go[(8 * ty + u) * 64 + 8 * tx + w] = acc[u][w];
}

// the pair's Gram matrix (sum of the slices), the test whether its live rows are orthogonal already, and if not its eigenvectors
__global__ void __launch_bounds__(256)
k_bj_eig(const double* __restrict__ Gpart, int nslice, double dead2, double tol, int max_inner, double* __restrict__ Qbuf,
         int* __restrict__ active, unsigned long long* __restrict__ worst) {
  extern __shared__ double bj_lds[];
  double* G = bj_lds;                     // [64][65]
  double* Q = G + 64 * MIK_BJ_LD;         // [64][65]
  __shared__ double cs[32][2];
  __shared__ int pq[32][2];
  __shared__ double red[4];
  __shared__ int flag;
  const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
  double acc[4][4];
  {
    const double* gi = Gpart + (long)blockIdx.x * nslice * 64 * 64;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        double v = 0.0;
        for (int sl = 0; sl < nslice; ++sl) v += gi[(long)sl * 64 * 64 + (4 * ty + u) * 64 + 4 * tx + w];
        acc[u][w] = v;
      }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      G[(4 * ty + u) * MIK_BJ_LD + 4 * tx + w] = acc[u][w];
      Q[(4 * ty + u) * MIK_BJ_LD + 4 * tx + w] = (4 * ty + u == 4 * tx + w) ? 1.0 : 0.0;
    }
  __syncthreads();
  // how far from orthogonal are the live rows of this pair?  (rows below a hundredth of the cut-off are the null space: noise)
  double far = 0.0;
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int p = 4 * ty + u, q = 4 * tx + w;
      const double gp = G[p * MIK_BJ_LD + p], gq = G[q * MIK_BJ_LD + q];
      if (p != q && gp > dead2 && gq > dead2) far = fmax(far, fabs(acc[u][w]) / sqrt(gp * gq));
    }
  for (int o = 32; o > 0; o >>= 1) far = fmax(far, __shfl_xor(far, o));
  if ((t & 63) == 0) red[t >> 6] = far;
  __syncthreads();
  if (t == 0) {
    far = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    atomicMax(worst, (unsigned long long)__double_as_longlong(far));
    flag = far > tol ? 1 : 0;
    active[blockIdx.x] = flag;
  }
  __syncthreads();
  if (!flag) return;
  // two-sided cyclic Jacobi on G (64 players, 63 rounds of 32 disjoint rotations per sweep), eigenvectors accumulated in Q.  A round:
  // 32 threads form the rotations; then G <- J^T G J as 32 x 32 independent 2 x 2 blocks (block (k1, k2) = rows of pair k1, columns of
  // pair k2: R_k1^T [..] R_k2, in place) and Q <- Q J column pair by column pair -- one barrier-separated phase, not two.  Rotations are
  // applied down to 1e-16 relative; the sweeps end when none exceeded 1e-15 (below that they chase rounding noise for ever).
  // max_inner: while the rows are still far from orthogonal the outer iteration does not need the eigenvectors of THIS Gram matrix
  // to full accuracy -- any orthogonal Q is a valid step, and the first sweeps of a Jacobi iteration do most of the work
  for (int sweep = 0; sweep < max_inner; ++sweep) {
    if (t == 0) flag = 0;
    __syncthreads();
    for (int r = 0; r < 63; ++r) {
      if (t < 32) {
        int p, q;
        bj_pair(64, r, t, p, q);
        const double app = G[p * MIK_BJ_LD + p], aqq = G[q * MIK_BJ_LD + q], apq = G[p * MIK_BJ_LD + q];
        const double den = sqrt(fabs(app * aqq));
        double c = 1.0, sn = 0.0;
        if (den > 0.0 && fabs(apq) > 1e-16 * den) {
          const double zeta = (aqq - app) / (2.0 * apq);
          const double tt = ((zeta >= 0.0) ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
          c = 1.0 / sqrt(1.0 + tt * tt);
          sn = c * tt;
          if (fabs(apq) > 1e-15 * den) flag = 1;
        }
        pq[t][0] = p, pq[t][1] = q;
        cs[t][0] = c, cs[t][1] = sn;
      }
      __syncthreads();
      for (int e = t; e < 32 * 32; e += 256) {  // 2 x 2 blocks of G
        const int k1 = e >> 5, k2 = e & 31;
        const int p1 = pq[k1][0], q1 = pq[k1][1], p2 = pq[k2][0], q2 = pq[k2][1];
        const double c1 = cs[k1][0], s1 = cs[k1][1], c2 = cs[k2][0], s2 = cs[k2][1];
        const double gpp = G[p1 * MIK_BJ_LD + p2], gpq = G[p1 * MIK_BJ_LD + q2], gqp = G[q1 * MIK_BJ_LD + p2], gqq = G[q1 * MIK_BJ_LD + q2];
        // rows:  [p1; q1] <- [c1 -s1; s1 c1] [p1; q1]
        const double rpp = c1 * gpp - s1 * gqp, rpq = c1 * gpq - s1 * gqq, rqp = s1 * gpp + c1 * gqp, rqq = s1 * gpq + c1 * gqq;
        // columns: [p2 q2] <- [p2 q2] [c2 s2; -s2 c2]
        G[p1 * MIK_BJ_LD + p2] = c2 * rpp - s2 * rpq;
        G[p1 * MIK_BJ_LD + q2] = s2 * rpp + c2 * rpq;
        G[q1 * MIK_BJ_LD + p2] = c2 * rqp - s2 * rqq;
        G[q1 * MIK_BJ_LD + q2] = s2 * rqp + c2 * rqq;
      }
      for (int e = t; e < 32 * 64; e += 256) {  // Q <- Q J
        const int k = e >> 6, row = e & 63, p = pq[k][0], q = pq[k][1];
        const double c = cs[k][0], sn = cs[k][1];
        const double qp = Q[row * MIK_BJ_LD + p], qq = Q[row * MIK_BJ_LD + q];
        Q[row * MIK_BJ_LD + p] = c * qp - sn * qq;
        Q[row * MIK_BJ_LD + q] = sn * qp + c * qq;
      }
      __syncthreads();
    }
    if (!flag) break;  // (everyone reads it between the last barrier above and the next one)
    __syncthreads();
  }
  double* qo = Qbuf + (long)blockIdx.x * 64 * 64;
  for (int e = t; e < 64 * 64; e += 256) qo[e] = Q[(e >> 6) * MIK_BJ_LD + (e & 63)];
}

// X <- Q^T X for the 64 rows of an active pair and a chunk of 64 columns, both matrices (z = 0: B, 1: W): a 64 x 64 x 64 product from
// LDS, 16 x 16 threads with 4 x 4 outputs each (rows 4 ty .., columns 4 tx ..).  (One wavefront with 8 x 8 outputs per thread, the form
// that sped up k_bj_gram, was measured 2.7 x SLOWER here -- 661 against 242 us per round at M = 4000: Q and the X chunk are 64 KB of LDS
// per wavefront, two wavefronts per CU.)
__global__ void __launch_bounds__(256)
k_bj_rotate(double* __restrict__ Bm, double* __restrict__ Wm, long ld, int ncols, const int* __restrict__ order, int nb, int round,
            const double* __restrict__ Qbuf, const int* __restrict__ active) {
  if (!active[blockIdx.x]) return;
  __shared__ double Qs[64 * 64];  // Q[k][i]
  __shared__ double Xs[64 * 64];  // X[k][c]
  __shared__ int idx[64];
  const int t = threadIdx.x, ty = t >> 4, tx = t & 15;
  int bi, bj;
  bj_pair(nb, round, blockIdx.x, bi, bj);
  if (t < 64) idx[t] = order[(t < 32 ? bi : bj) * MIK_BJ_B + (t & 31)];
  const double* qi = Qbuf + (long)blockIdx.x * 64 * 64;
  for (int e = t; e < 64 * 64; e += 256) Qs[e] = qi[e];
  __syncthreads();
  double* X = blockIdx.z ? Wm : Bm;
  const int c0 = blockIdx.y * 64;
  for (int e = t; e < 64 * 64; e += 256) {
    const int k = e >> 6, c = c0 + (e & 63), row = idx[k];
    Xs[e] = (row >= 0 && c < ncols) ? X[(long)row * ld + c] : 0.0;
  }
  __syncthreads();
  double acc[4][4] = {};
#pragma unroll 8
  for (int k = 0; k < 64; ++k) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a[u] = Qs[k * 64 + 4 * ty + u];
      b[u] = Xs[k * 64 + 4 * tx + u];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int w = 0; w < 4; ++w) acc[u][w] += a[u] * b[w];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int row = idx[4 * ty + u];
    if (row < 0) continue;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int c = c0 + 4 * tx + w;
      if (c < ncols) X[(long)row * ld + c] = acc[u][w];
    }
  }
}

__global__ void __launch_bounds__(256) k_rownorm2(const double* __restrict__ B, long ld, int n, double* __restrict__ out) {
  const int r = blockIdx.x;
  double acc = 0.0;
  for (long c = threadIdx.x; c < ld; c += 256) {
    const double x = B[(long)r * ld + c];
    acc += x * x;
  }
  __shared__ double red[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[r] = red[0] + red[1] + red[2] + red[3];
}

// out[i][j] = sum_k B[k][i] d[k] W[k][j]  (i, j < n); both operands are read along their contiguous rows
__global__ void __launch_bounds__(256)
k_pinv_gemm(const double* __restrict__ B, const double* __restrict__ W, const double* __restrict__ d, long ld, int n,
            double* __restrict__ out) {
  __shared__ double sb[16][64], sw[16][64];
  const int i0 = blockIdx.y * 64, j0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // 16 x 16 threads, 4 x 4 outputs each
  double acc[4][4] = {};
  for (int k0 = 0; k0 < n; k0 += 16) {
    for (int e = threadIdx.x; e < 16 * 64; e += 256) {
      const int kk = e >> 6, c = e & 63, k = k0 + kk;
      const bool in = k < n;
      sb[kk][c] = (in && i0 + c < n) ? B[(long)k * ld + i0 + c] * d[k] : 0.0;
      sw[kk][c] = (in && j0 + c < n) ? W[(long)k * ld + j0 + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = sb[kk][ty * 4 + u];
        b[u] = sw[kk][tx * 4 + u];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int w = 0; w < 4; ++w) acc[u][w] += a[u] * b[w];
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int i = i0 + ty * 4 + u, j = j0 + tx * 4 + w;
      if (i < n && j < n) out[(long)i * ld + j] = acc[u][w];
    }
}

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
__device__ __forceinline__ double mw_entry_t(const Vario& v, int mode, double x1, double y1, double z1, double x2, double y2, double z2) {
  if (MODEL < 0) return mw_entry(v, mode, x1, y1, z1, x2, y2, z2);
  const double dx = x1 - x2, dy = y1 - y2, dz = z1 - z2;  // z = 0 in 2-D
  const double d2 = dx * dx + dy * dy + dz * dz;
  return -vario<(MODEL < 0 ? 0 : MODEL), false>(v, sqrt(d2), d2);
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
  double m[RI][RI], rhs[RI];
#pragma unroll
  for (int i = 0; i < RI; ++i) {
    const int row = ty + G * i;
#pragma unroll
    for (int j = 0; j <= i; ++j) {
      const int col = tx + G * j;
      double v = (row == col) ? 1.0 : 0.0;  // padding rows / columns: identity
      if (row < K && col < K)
        v = (row == col) ? shift : shift + mw_entry_t<MODEL>(a.v, a.mode, csx[row], csy[row], csz[row], csx[col], csy[col], csz[col]);
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
#pragma unroll
        for (int i = cc; i < RI; ++i) ab[ty + G * i] = m[i][cc];
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
      if (!MIK_MWC_LEAN(G, RI) && ty <= cx) u[cc] = 0.0;  // rows / columns <= c of the diagonal local tile are finished
      if (tx <= cx) w[cc] = 0.0;
      const double ur = (ty < 3 ? ab[NB + ty] : 0.0) * inv;
      // the five inner products z and sigma^2 are made of, sum_c y_p(c) y_q(c) / d(c), are formed ONCE at the end from this log
      // (every thread used to accumulate all five in every step)
      if (lt < 4) ylog[lt * NB + c] = (lt < 3) ? ab[NB + lt] : inv;
      if (MIK_MWC_LEAN(G, RI)) {  // the largest one-wavefront tiles: the row factors are read as they are used (RI fewer live doubles)
#pragma unroll
        for (int i = cc; i < RI; ++i) {
          double ui = ab[ty + G * i] * inv;
          if (i == cc && ty <= cx) ui = 0.0;
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

// ------------------------------------------------------------------------------------------------
// Variogram-fit statistics (core.py:759-836 _find_statistics -> core.py:654-756 _krige): station i is kriged from
// stations 0..i-1 for i = 1..N-1.  The reference solves N-1 growing dense systems (O(N^4)); here the inverse
// of the bordered matrix [[0, 1^T], [1, -Gamma_i]] (Lagrange row FIRST so a new station appends a row/column) is
// grown by the bordering identity: with u = [1; -gamma(d(i, 0..i-1))] (row i of the assembled matrix),
// x = Minv u is the kriging solution itself (k_i = x[1:].y, ss_i = -x.u) and
//   Minv' = [[Minv + x x^T / s, -x / s], [-x^T / s, 1 / s]],  s = 0 - u.x = ss_i
// so each step is one mat-vec, one tiny reduction and one rank-1 update: O(N^3) flops, 24 N^3 / 3 bytes in total.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_stat_matvec(const double* __restrict__ S, long ld, int m, const double* __restrict__ Trow /* T[i][0..i-1] */,
              double* __restrict__ x) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= m) return;
  const double* r = S + (long)row * ld;
  double s = 0.0;
  for (int b = lane; b < m; b += 64) s += r[b] * (b == 0 ? 1.0 : Trow[b - 1]);
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) x[row] = s;
}

// one block: k = sum_j x[1+j] y[j], ss = -(x[0] + sum_j x[1+j] Trow[j]); out[0] = k, out[1] = ss, out[2] = 1/ss
__global__ void __launch_bounds__(256)
k_stat_reduce(const double* __restrict__ x, int m, const double* __restrict__ Trow, const double* __restrict__ y,
              double* __restrict__ kout, double* __restrict__ ssout, double* __restrict__ scal) {
  __shared__ double sk[256], su[256];
  double k = 0.0, u = 0.0;
  for (int j = threadIdx.x; j < m - 1; j += 256) {
    const double xv = x[1 + j];
    k += xv * y[j];
    u += xv * Trow[j];
  }
  sk[threadIdx.x] = k;
  su[threadIdx.x] = u;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      sk[threadIdx.x] += sk[threadIdx.x + o];
      su[threadIdx.x] += su[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double ss = -(x[0] + su[0]);
    *kout = sk[0];
    *ssout = ss;
    scal[0] = 1.0 / ss;
  }
}

__global__ void __launch_bounds__(256)
k_stat_update(double* __restrict__ S, long ld, int m, const double* __restrict__ x, const double* __restrict__ scal) {
  const int b = blockIdx.x * 64 + (threadIdx.x & 63);
  const double sinv = scal[0];
  if (b > m) return;
  const double xb = (b < m) ? x[b] : 0.0;
  for (int a = blockIdx.y * 64 + (threadIdx.x >> 6); a < blockIdx.y * 64 + 64 && a <= m; a += 4) {
    double* p = S + (long)a * ld + b;
    if (a < m && b < m) *p += x[a] * xb * sinv;
    else if (a == m && b == m) *p = sinv;
    else *p = -((a == m) ? xb : x[a]) * sinv;
  }
}

// first station pair closer than 1e-10 (the reference's solve would be singular): flag = 1
template <int NDIM>
__global__ void __launch_bounds__(256)
k_stat_dupes(const double* __restrict__ xs, const double* __restrict__ ys, const double* __restrict__ zs, int N,
             int* __restrict__ flag) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const double x = xs[i], y = ys[i], z = (NDIM == 3) ? zs[i] : 0.0;
  for (int j = 0; j < i; ++j) {
    const double dx = x - xs[j], dy = y - ys[j], dz = (NDIM == 3) ? z - zs[j] : 0.0;
    if (sqrt(dx * dx + dy * dy + dz * dz) <= 1e-10) { atomicOr(flag, 1); return; }
  }
}

// tools/kernel_bench only: the full (non-symmetric) contraction loop with components removed (see gemm_core ABL)
template <int NAI, int ABL>
__global__ void __launch_bounds__(64 * 2 * (8 / NAI), 2 * (4 / NAI))
k_contract_ablate(const double* __restrict__ Ainv, long lda, const double* __restrict__ Bt, long ldb,
                  double* __restrict__ part, int palloc, int nIblk, int kend) {
  __shared__ GemmSmem sm;
  int iblk, tblk;
  if (!super_tile(nIblk, palloc / MIK_BN, iblk, tblk)) return;
  d4 acc[NAI][4];
#pragma unroll
  for (int x = 0; x < NAI; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
  // ABL & 16: ragged K range as in the symmetric form (k >= i0), without the doubling step
  gemm_core<NAI, (ABL & 47)>(Ainv + (long)iblk * MIK_BM * lda, lda, Bt + (long)tblk * MIK_BN * ldb, ldb,
                             (ABL & 16) ? iblk * MIK_BM : 0, kend, acc, sm);
  double s = 0.0;
#pragma unroll
  for (int x = 0; x < NAI; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) s += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
  if (s == 1.2345e-300) part[(long)iblk * palloc + tblk * MIK_BN + threadIdx.x % 128] = s;
}

// tools/kernel_bench only (round 3, the tile-shape experiment): the symmetric contraction with a 256 (rows of A_inv) x 128 (points)
// block tile -- 16 wavefronts of 32 x 64 on the same MFMA loop, ONE 1024-thread block per CU, 96 KB of LDS, persistent over the
// same per-XCD tile queue.  Per tile step it requests (256 + 128) x 16 operand doubles for 256 x 128 x 16 multiply-adds, 25 % less
// than two 128 x 128 tiles.  part[] has one row per 256-row block.  (A 256 x 256 tile does not exist for this register tiling:
// 16 waves of 32 x 64 cover 256 x 128; 32 x 64 per wave at 128 VGPRs is what lets 4 waves share a SIMD.)
template <bool SYM>
__global__ void __launch_bounds__(1024, 1)
k_contract256(const double* __restrict__ Ainv, long lda, const double* __restrict__ Bt, long ldb, double* __restrict__ part, int palloc,
              int nIblk /* 256-row blocks */, int kend, unsigned long long* __restrict__ queue) {
  constexpr int NAI = 2, BM = 256, WROWS = 32, NWM = BM / WROWS;
  extern __shared__ double smem256[];
  GemmSmemT<BM>& sm = *reinterpret_cast<GemmSmemT<BM>*>(smem256);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int xcd = (int)(xcc & 7);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, lq = lane >> 4, lc = lane & 15;
  int steal = 0;
  for (;;) {
    int iblk, tblk;
    const int xq = (xcd + steal) & 7;
    if (threadIdx.x == 0) sm.next = (long)__hip_atomic_fetch_add(&queue[xq], 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const long seq = sm.next;
    const int kind = super_tile_at(nIblk, palloc / MIK_BN, xq, seq, iblk, tblk);
    __syncthreads();
    if (kind == 2) {
      if (++steal == 8) return;
      continue;
    }
    if (kind == 1) continue;
    const int i0 = iblk * BM, t0 = tblk * MIK_BN;
    const double* Ag = Ainv + (long)i0 * lda;
    const double* Bg = Bt + (long)t0 * ldb;
    d4 acc[NAI][4];
#pragma unroll
    for (int x = 0; x < NAI; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
    if (SYM) {
      const int kd = (i0 + BM) < kend ? (i0 + BM) : kend;
      gemm_core<NAI, 0, BM>(Ag, lda, Bg, ldb, i0, kend, acc, sm, kd - MIK_BK);
    } else {
      gemm_core<NAI, 0, BM>(Ag, lda, Bg, ldb, 0, kend, acc, sm);
    }
    double cs[4];
#pragma unroll
    for (int bp = 0; bp < 2; ++bp) {
      double bv[2][4 * NAI];
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const long t = t0 + wn * 64 + (2 * bp + b2) * 16 + lc;
        const double* brow = Bt + t * ldb + i0 + wm * WROWS + lq;
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r) bv[b2][ai * 4 + r] = brow[ai * 16 + 4 * r];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int bi = 2 * bp + b2;
        double sacc = 0.0;
#pragma unroll
        for (int ai = 0; ai < NAI; ++ai)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc += bv[b2][ai * 4 + r] * acc[ai][bi][r];
        sacc += __shfl_xor(sacc, 16);
        sacc += __shfl_xor(sacc, 32);
        cs[bi] = sacc;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    double* red = &sm.As[0][0][0];
    if (lq == 0) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) red[wm * 128 + wn * 64 + bi * 16 + lc] = cs[bi];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
      double v = 0.0;
#pragma unroll
      for (int w = 0; w < NWM; ++w) v += red[w * 128 + threadIdx.x];
      part[(long)iblk * palloc + t0 + threadIdx.x] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Experimental semivariogram of the constructor (core.py:432-505): all station pairs i < j, distance d_ij and
// g_ij = (z_i - z_j)^2 / 2, equal-width lag bins between min d and max d + 0.001.  Pass 1: min / max of d per block;
// pass 2: per-block sums of d, g and counts per bin (LDS atomics), reduced on the host.  One 64 x 64 pair tile per block.
// ------------------------------------------------------------------------------------------------
template <int NDIM>
__device__ __forceinline__ double pair_dist(const double* xs, const double* ys, const double* zs, int i, int j) {
  if (NDIM == 1) {
    const double la1 = ys[i] * MIK_PI / 180.0, la2 = ys[j] * MIK_PI / 180.0;
    // core.py:441-451: great_circle_distance(x1, y1, x2, y2) on meshgrids, pairs kept where row > column, i.e.
    // point 1 = the smaller station index, point 2 = the larger
    return gc_dist(xs[i], cos(la1), sin(la1), xs[j], cos(la2), sin(la2));
  }
  const double dx = xs[i] - xs[j], dy = ys[i] - ys[j];
  double s2 = dx * dx + dy * dy;
  if (NDIM == 3) {
    const double dz = zs[i] - zs[j];
    s2 += dz * dz;
  }
  return sqrt(s2);
}

template <int NDIM>
__global__ void __launch_bounds__(256)
k_vg_minmax(const double* __restrict__ xs, const double* __restrict__ ys, const double* __restrict__ zs, int N,
            double* __restrict__ out /* 2 per block */) {
  __shared__ double smin[256], smax[256];
  double lo = 1e300, hi = -1e300;
  if (blockIdx.x <= blockIdx.y) {  // tile (rows i of blockIdx.x, columns j of blockIdx.y), pairs i < j
    const int j = blockIdx.y * 64 + (threadIdx.x & 63);
    for (int r = threadIdx.x >> 6; r < 64; r += 4) {
      const int i = blockIdx.x * 64 + r;
      if (i < j && j < N) {
        const double d = pair_dist<NDIM>(xs, ys, zs, i, j);
        lo = d < lo ? d : lo;
        hi = d > hi ? d : hi;
      }
    }
  }
  smin[threadIdx.x] = lo;
  smax[threadIdx.x] = hi;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      smin[threadIdx.x] = fmin(smin[threadIdx.x], smin[threadIdx.x + o]);
      smax[threadIdx.x] = fmax(smax[threadIdx.x], smax[threadIdx.x + o]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const long b = (long)blockIdx.y * gridDim.x + blockIdx.x;
    out[2 * b] = smin[0];
    out[2 * b + 1] = smax[0];
  }
}

#define MIK_VG_MAXLAGS 64
template <int NDIM>
__global__ void __launch_bounds__(256)
k_vg_bin(const double* __restrict__ xs, const double* __restrict__ ys, const double* __restrict__ zs,
         const double* __restrict__ vals, int N, int nlags, const double* __restrict__ edges /* nlags + 1 */,
         double* __restrict__ out /* per block: nlags x 3 */) {
  __shared__ double sd[MIK_VG_MAXLAGS], sg[MIK_VG_MAXLAGS], sc[MIK_VG_MAXLAGS], se[MIK_VG_MAXLAGS + 1];
  if (threadIdx.x < nlags) sd[threadIdx.x] = sg[threadIdx.x] = sc[threadIdx.x] = 0.0;
  if (threadIdx.x <= nlags) se[threadIdx.x] = edges[threadIdx.x];
  __syncthreads();
  if (blockIdx.x <= blockIdx.y) {
    const int j = blockIdx.y * 64 + (threadIdx.x & 63);
    const double inv = (se[1] > se[0]) ? 1.0 / (se[1] - se[0]) : 0.0;
    for (int r = threadIdx.x >> 6; r < 64; r += 4) {
      const int i = blockIdx.x * 64 + r;
      if (i < j && j < N) {
        const double d = pair_dist<NDIM>(xs, ys, zs, i, j);
        const double dz = vals[i] - vals[j];
        int b = (int)((d - se[0]) * inv);
        b = b < 0 ? 0 : (b > nlags - 1 ? nlags - 1 : b);
        while (b > 0 && d < se[b]) --b;                  // the reference's own tests: bins[n] <= d < bins[n+1]
        while (b < nlags - 1 && d >= se[b + 1]) ++b;
        if (d >= se[b] && d < se[b + 1]) {
          atomicAdd(&sd[b], d);
          atomicAdd(&sg[b], 0.5 * dz * dz);
          atomicAdd(&sc[b], 1.0);
        }
      }
    }
  }
  __syncthreads();
  if (threadIdx.x < nlags) {
    const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
    double* o = out + blk * 3 * nlags;
    o[threadIdx.x] = sd[threadIdx.x];
    o[nlags + threadIdx.x] = sg[threadIdx.x];
    o[2 * nlags + threadIdx.x] = sc[threadIdx.x];
  }
}

}  // namespace mik
