// mik_k_core.h -- K1 assembly, prediction points from axes / masks, MFMA self test, variogram-fit statistics, experimental semivariogram
// (one of the section headers mik_kernels.h is the umbrella of; every section is included by exactly one translation unit of the library)
#pragma once
#include "mik_dev.h"

namespace mik {

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

// The factor exchange moves the UPPER BLOCK TRIANGLE of the inverse only (round 6): an inverse the device computed is exactly symmetric, the
// receiver mirrors the lower block triangle (k_mirror_upper) and holds the sender's matrix bit for bit.  Packed layout: block row I (128 rows) keeps its columns [128 I, Mp), row-major, block
// rows one after the other: tri_len(Mp) = Mp (Mp + 128) / 2 doubles (N = 8000: 264 MB instead of 520 MB).  One block per matrix row;
// every segment starts on a 1 KB boundary and has an even length: double2 copies.  unpack = the way back into T.
__host__ __device__ inline size_t tri_off(size_t Mp, size_t I) { return 128 * I * Mp - 8192 * (I * (I > 0 ? I - 1 : 0)); }
__host__ __device__ inline size_t tri_len(size_t Mp) { return tri_off(Mp, Mp / 128); }
__global__ void __launch_bounds__(256) k_tri_pack(double* __restrict__ T, size_t Mp, double* __restrict__ P, int unpack) {
  const size_t row = blockIdx.x, I = row >> 7, len = Mp - 128 * I;
  double* t = T + row * Mp + 128 * I;
  double* p = P + tri_off(Mp, I) + (row & 127) * len;
  for (size_t i = 2 * threadIdx.x; i < len; i += 512) {
    if (unpack) *reinterpret_cast<double2*>(t + i) = *reinterpret_cast<const double2*>(p + i);
    else *reinterpret_cast<double2*>(p + i) = *reinterpret_cast<const double2*>(t + i);
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
// exp_neg_lean (mik_dev.h) on n arguments: the moving window's 19-instruction exponential against the caller's reference
__global__ void __launch_bounds__(256) k_selftest_exp(const double* __restrict__ x, double* __restrict__ out, int n) {
  const ExpTab tab = exp_tab_load();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = exp_neg_lean(x[i], tab);
}
__global__ void k_selftest_mfma4(double* out /*4x16 row-major*/) {
  const int l = threadIdx.x;
  const double a = (double)((l & 3) * 7 + (l >> 4) * 3 + 1);    // A[i=l&3][k=l>>4], same for every block
  const double b = (double)((l >> 4) * 11 + (l & 15) * 5 + 2);  // B[k=l>>4][col=l&15]
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
  out[(l >> 4) * 16 + (l & 15)] = d;                            // D[i=l>>4][col=l&15]
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
