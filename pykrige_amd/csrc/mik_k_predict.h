// mik_k_predict.h -- K3a right-hand sides, K3b dense and range-aware contraction, point sort
// (one of the section headers mik_kernels.h is the umbrella of; every section is included by exactly one translation unit of the library)
#pragma once
#include "mik_dev.h"

namespace mik {

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
  int nKf;               // SP, F8 (round 5, option "sparse_ktile" 8): the flags are per 8 stations, nKf = Mp / 8 their row stride.  The candidates
                         // stay per 16 stations: right-hand sides are computed and stored in whole 128-byte lines (per 8 stations k_rhs was
                         // 12 % slower: half-line stores)
  const unsigned* perm;  // SP, nullable: the chunk's points in sorted order -- point t of the chunk is point perm[t] of the WHOLE list;
                         // px / py / pz / extra / zout are then the list's base pointers, not the chunk's (option "sort_points")
};

template <int MODEL, int NDIM, bool SP = false, bool F8 = false>  // F8 (SP only): flags per 8 stations (row stride nKf) instead of per 16
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
  // SP (round 5): the block walks the COMPACT list of its point block's candidate K tiles, sixteen tiles per iteration -- every lane has
  // a station to work on.  (Striding over all Mp columns and skipping the non-candidates, as rounds 4 did, left three quarters of the
  // lanes of an iteration idle at config 5 -- 10 % of the tiles are candidates -- and paid a dependent flag load per iteration.)
  __shared__ unsigned short slist[SP ? MIK_SP_MAXK16 : 16];
  __shared__ int sncand;
  if (SP) {
    if (threadIdx.x < 64) {  // wavefront 0: ascending list by ballot / popcount
      const unsigned char* crow = a.cand + (long)(t0 >> 7) * a.nK16;
      int nc = 0;
      for (int base = 0; base < a.nK16; base += 64) {
        const int k = base + (int)threadIdx.x;
        const bool on = k < a.nK16 && crow[k] != 0;
        const unsigned long long m = __ballot(on);
        if (on) slist[nc + __popcll(m & ((1ULL << threadIdx.x) - 1ULL))] = (unsigned short)k;
        nc += __popcll(m);
      }
      if (threadIdx.x == 0) sncand = nc;
    }
    __syncthreads();
  }
  const int nit = SP ? sncand : a.Mp;

  for (int it = SP ? (int)(threadIdx.x >> 4) : (int)threadIdx.x; it < nit; it += SP ? 16 : 256) {
    const int j = SP ? 16 * (int)slist[it] + (int)(threadIdx.x & 15) : it;
    double val[MIK_TP];
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
    if (SP) {  // 16 (8) lanes = one K tile; every writer writes the same 1
      const unsigned long long m = __ballot(nz);
      const int l = threadIdx.x & 63;
      if (F8) {
        if ((l & 7) == 0 && ((m >> l) & 0xffULL) != 0) a.flags[(long)(t0 >> 7) * a.nKf + (j >> 3)] = 1;
      } else if ((l & 15) == 0 && ((m >> l) & 0xffffULL) != 0) {
        a.flags[(long)(t0 >> 7) * a.nK16 + (j >> 4)] = 1;
      }
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
                                                 unsigned char* __restrict__ cand, const unsigned* __restrict__ perm, int whole128,
                                                 unsigned char* __restrict__ flags, int nKf) {
  __shared__ double red[6][2];
  const int tb = blockIdx.x, t = tb * 128 + threadIdx.x;
  // the point block's row of flags (one byte per K tile of nKf, set by k_rhs behind this kernel on the stream): cleared here instead of by a
  // memset of its own (round 6: one dispatch less on every launch's chain; nKf is a multiple of 8, the rows are 8-byte aligned)
  for (int w = threadIdx.x; w < nKf / 8; w += 128) reinterpret_cast<unsigned long long*>(flags + (long)tb * nKf)[w] = 0ULL;
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

// geographic points (lon, lat in degrees) as unit vectors: what k_sp_cand's boxes and the point sort's keys are built from
// (coordinates_type = 'geographic' with the range-aware contraction, round 5)
__global__ void __launch_bounds__(256) k_geo_unit_p(const double* __restrict__ lon, const double* __restrict__ lat, long n,
                                                    double* __restrict__ ux, double* __restrict__ uy, double* __restrict__ uz) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  if (t >= n) return;
  const double lo = lon[t] * MIK_PI / 180.0, la = lat[t] * MIK_PI / 180.0;
  const double c = cos(la);
  ux[t] = c * cos(lo);
  uy[t] = c * sin(lo);
  uz[t] = sin(la);
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
// h8 (round 5, "sparse_ktile" 8): the flags and the list are per 8 stations; a list POSITION of the contraction is then a PAIR of
// list-adjacent 8-station tiles (entries 2 w, 2 w + 1 -- one dword), an odd last entry is paired with 0xffff; kcount = positions.
__global__ void __launch_bounds__(64) k_sp_lists_g(const unsigned char* __restrict__ flags, int nK16,
                                                   unsigned short* __restrict__ klist, int* __restrict__ kcount,
                                                   int* __restrict__ ntiles, int h8 = 0) {
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
    if (h8) {
      if (nk & 1) kl[nk] = (unsigned short)0xffffu;  // (nk odd < nK16, which is even: the slot exists)
      nk = (nk + 1) >> 1;
    }
    kcount[tb] = nk;
    ntiles[tb] = (nk + 7) / 8;
  }
}

// Tile records of k_contract_spg, in k_sp_tiles' order (groups of MIK_ST point blocks, group g on XCD g % 8, inside a group tile
// position ascending = longest K loops first, point block fast).  Record (two uint4):
//   [0] = {tblk << 10 | r, nk, klist[nk - 1], klist[nk - 2]}      [1] = the eight row groups klist[8 r .. 8 r + 7] (u16 each)
// stats: [0] tiles, [1] off-diagonal K tiles summed over the tiles, [2] (row group, K tile) products of the triangular parts.
// H8: a position is a dword of the list (two 8-station tiles), the record has three uint4: [1], [2] = the eight positions of the tile.
// Round 6: one block per GROUP (round 5: one 1024-thread block for the whole launch, every thread writing its point block's records one after the
// other -- 40 us alone, 280 us beside the other lane's k_rhs, on the critical chain of every launch: profiles/r06_predict_timeline_c5.txt).  Every
// block recomputes the group offsets (8 KB of counts, L2-resident), then writes its own group's records, one (point block, tile) pair per thread;
// block 0 also writes xoff / stats and zeroes the tile queues (the memset that used to sit between this kernel and the contraction).
template <bool H8 = false>
__global__ void __launch_bounds__(256) k_sp_tiles_g(const int* __restrict__ ntiles, const int* __restrict__ kcount,
                                                    const unsigned short* __restrict__ klist, int nK16, int nTblk,
                                                    uint4* __restrict__ recs, int* __restrict__ xoff,
                                                    unsigned long long* __restrict__ stats, int st, unsigned long long* __restrict__ queue) {
  // st = point blocks per group (option "sparse_group", 1 .. 16; 16 by default since round 5)
  __shared__ int gcnt[1024 + 1], goff[1024 + 1], xtot[9], nr[16], rowbase[256 + 1];
  __shared__ unsigned long long ksum, dsum;
  const int nG = (nTblk + st - 1) / st;
  const int gg = blockIdx.x;
  if (threadIdx.x == 0) ksum = 0ULL, dsum = 0ULL;
  for (int g = threadIdx.x; g < nG; g += 256) {
    int c = 0;
    for (int q = 0; q < st; ++q) {
      const int tb = g * st + q;
      if (tb >= nTblk) break;
      c += ntiles[tb];
    }
    gcnt[g] = c;
  }
  if (threadIdx.x < 16) {
    const int t2 = gg * st + (int)threadIdx.x;
    nr[threadIdx.x] = ((int)threadIdx.x < st && t2 < nTblk) ? ntiles[t2] : 0;
  }
  __syncthreads();
  if (threadIdx.x < 8) {  // exclusive scan of the groups of XCD x
    int sum = 0;
    for (int q = threadIdx.x; q < nG; q += 8) {
      goff[q] = sum;
      sum += gcnt[q];
    }
    xtot[threadIdx.x] = sum;
  }
  int maxnr = 0;
#pragma unroll
  for (int qq = 0; qq < 16; ++qq) maxnr = nr[qq] > maxnr ? nr[qq] : maxnr;
  if (maxnr > 256) maxnr = 256;  // (ntiles <= Mp / 128 <= 181 while 32-bit DMA offsets address the inverse: the gathered form's own limit)
  for (int r = threadIdx.x; r <= maxnr; r += 256) {  // rowbase[r] = records of this group in the tile positions below r
    int b = 0;
    for (int rr = 0; rr < r; ++rr)
#pragma unroll
      for (int qq = 0; qq < 16; ++qq) b += nr[qq] > rr ? 1 : 0;
    rowbase[r] = b;
  }
  __syncthreads();
  if (blockIdx.x == 0) {
    unsigned long long ks = 0ULL, ds = 0ULL;
    for (int tb = threadIdx.x; tb < nTblk; tb += 256) {
      const int nk = kcount[tb], full = nk / 8, rem = nk - 8 * full;
      ks += (unsigned long long)((long)full * nk - 4L * full * (full + 1));  // sum over full tiles r of nk - 8 (r + 1)
      ds += (unsigned long long)(36 * full + rem * (rem + 1) / 2);
    }
    atomicAdd(&ksum, ks);
    atomicAdd(&dsum, ds);
    if (threadIdx.x < 8) queue[threadIdx.x] = 0ULL;
    __syncthreads();
    if (threadIdx.x == 0) {
      int sum = 0;
      for (int x = 0; x < 8; ++x) {
        xoff[x] = sum;
        sum += xtot[x];
      }
      xoff[8] = sum;
      stats[0] = (unsigned long long)sum;
      stats[1] = ksum;
      stats[2] = dsum;
    }
  }
  int w0 = goff[gg];
  for (int x = 0; x < (gg & 7); ++x) w0 += xtot[x];
  for (int idx = threadIdx.x; idx < 16 * maxnr; idx += 256) {  // one (point block q of the group, tile r) pair per thread, point block fast
    const int q = idx & 15, r = idx >> 4;
    if (r >= nr[q]) continue;
    const int tb = gg * st + q;
    int before = 0;
#pragma unroll
    for (int qq = 0; qq < 16; ++qq) before += (qq < q && nr[qq] > r) ? 1 : 0;
    const int nk = kcount[tb];
    const unsigned short* kl = klist + (long)tb * nK16;
    const unsigned* kl32 = reinterpret_cast<const unsigned*>(kl);
    const unsigned k1 = nk >= 1 ? (H8 ? kl32[nk - 1] : (unsigned)kl[nk - 1]) : 0u, k2 = nk >= 2 ? (H8 ? kl32[nk - 2] : (unsigned)kl[nk - 2]) : 0u;
    uint4* out = recs + (H8 ? 3L : 2L) * (w0 + rowbase[r] + before);
    out[0] = make_uint4(((unsigned)tb << 10) | (unsigned)r, (unsigned)nk, k1, k2);
    if (H8) {
      out[1] = *reinterpret_cast<const uint4*>(kl + 16 * r);  // 32-byte aligned: the list stride is a multiple of 16
      out[2] = *reinterpret_cast<const uint4*>(kl + 16 * r + 8);
    } else {
      out[1] = *reinterpret_cast<const uint4*>(kl + 8 * r);  // 16-byte aligned: nK16 is a multiple of 8
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
  unsigned long long* prof;     // PROF: [block][wave][10] cycle sums per phase (diagnostic instantiation only)
};

// EPI (option "sparse_epilogue", default 1 in the 8-station form): a group's term of part[r][t] = sum_i delta_ti W_it is formed at the K
// step of the group's own 16 x 16 square -- its accumulators are final there, and the delta it needs IS that step's B tile in LDS --
// instead of from global memory after the K loop: no operand reads in the epilogue (a tenth of the kernel's fabric traffic, two memory
// round trips per tile during which the block issues no matrix instruction).  Round 4 measured it 1.7 % slower (its LDS reads were
// issued one by one, each waited for); round 5 (reads in one batch, the queue look-ahead out of the first K step): config 5 contraction
// 38.0 -> 36.3 ms per 2.1 M points.  PROF: the diagnostic instantiation (MIK_SPG_PROF=1: cycle sums per phase of the tile loop and per
// triangle step, printed per launch; profiles/r05_spg_tile_phases.txt).  Last: two of a tile's barriers are gone -- the queue record needs none
// on the fast path (it was written barriers ago), and the wave-rows' sums wait in LDS of their own (sred) for the barrier at the top of
// the next tile, where wavefronts 0 and 1 add and store them: 36.1 - 36.5 -> 36.7 - 37.2 M points/s at config 5.
// H8 (round 5, option "sparse_ktile" 8): the list is per 8 stations and a list POSITION is a dword = a pair (h0, h1) of list-adjacent
// 8-station tiles: a K step stages columns 8 h0 .. + 7 into the lower half of the 16-wide LDS tile and 8 h1 .. + 7 into the upper half
// (the per-lane DMA offset of the lanes that feed the upper half is shifted by 8 (h1 - h0) columns: one v_add + one v_cndmask by a
// constant lane mask per DMA), a 16-row group is two gathered 8-row groups, and an odd last entry (h1 = 0xffff) is a K step of its
// lower half alone (the m = 1 MFMAs are skipped) and a row group whose upper eight rows are left out of the epilogue.  Everything else
// -- positions, tiles of eight positions, the triangle, the queue -- is the 16-station form's.  -15 % work at BASELINE config 5 by the
// CPU model (profiles/r04b_sparse_granularity_model_cpu.txt: 718 instead of 780 stations in active tiles, work ~ n^2).
template <int NAI, bool EPI = false, bool H8 = false, bool PROF = false>
__global__ void __launch_bounds__(64 * 2 * (8 / NAI), 2 * (4 / NAI)) k_contract_spg(SpgArgs a) {
  static_assert(NAI == 2, "8 waves: 4 wave-rows of two 16-row groups x 2 wave-columns of 64 points");
  constexpr int WROWS = 16 * NAI, NWM = 128 / WROWS;
  constexpr int NTHR = 64 * 2 * (MIK_BM / WROWS), PROWS = NTHR / 8, NPASS = MIK_BM / PROWS;
  constexpr unsigned LDS_PASS = PROWS * MIK_BK * 8, LDS_BUF = MIK_BM * MIK_BK * 8;
  __shared__ GemmSmem sm;
  constexpr int RW = H8 ? 3 : 2;  // uint4 per tile record
  __shared__ uint4 srec[2 * RW];  // two tile records: the current tile's and the next one's
  __shared__ int sst[4];     // the queue owner's state: [0] sequences tried, [1] first record and [2] record count of the current sequence
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
    // (H8: the source slot without its half bit, as for aoffb -- see stage())
    boffb = (unsigned)(((long)lrow * a.ldb + (((slot ^ ((lrow >> 1) & 7)) & (H8 ? 3 : 7)) << 1)) * 8);
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
  auto pair_of = [](const uint4& r1, const uint4& r2, int gi) -> unsigned {  // H8: the dword (h0 | h1 << 16) of position gi of the tile
    return gi == 0 ? r1.x : gi == 1 ? r1.y : gi == 2 ? r1.z : gi == 3 ? r1.w : gi == 4 ? r2.x : gi == 5 ? r2.y : gi == 6 ? r2.z : r2.w;
  };
  // lanes whose DMA piece lands in the UPPER half (k 8 .. 15) of a row of the LDS tile: source slot c = slot ^ swizzle >= 4.  A image:
  // swizzle = row & 2 -> bit 2 of the lane; B image: swizzle = (row >> 1) & 7 -> bit 2 of the lane XOR bit 0 of the wave (row bit 3)
  const unsigned long long upA = 0xF0F0F0F0F0F0F0F0ULL, upB = (wave & 1) ? 0x0F0F0F0F0F0F0F0FULL : 0xF0F0F0F0F0F0F0F0ULL;
  auto sel = [](unsigned lo, unsigned hi, unsigned long long m) -> unsigned {  // per lane: m bit set ? hi : lo
    unsigned r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(lo), "v"(hi), "s"(m));  // (not volatile: a pure function of its operands)
    return r;
  };
  // kc: the K position -- 16 k (16-station form) or the pair code h0 | h1 << 16 (H8)
  auto stage = [&](const double* Bgu, int kc, int b) {
    int k = kc;
    unsigned dk = 0u;  // H8: byte shift of the upper-half lanes' source (their offsets carry no half bit): 8 (h1 - h0) columns; an odd
                       // tail's h1 = 0xffff: the lower half once more (its m = 1 products are skipped)
    if (H8) {
      const unsigned h0 = (unsigned)kc & 0xffffu, h1 = (unsigned)kc >> 16;
      k = 8 * (int)h0;
      dk = (h1 == 0xffffu) ? 0u : 64u * (h1 - h0);
    }
    const double* abase = uniform_ptr(Agu + k);
#pragma unroll
    for (int p = 0; p < NPASS; ++p) {
      const double* bbase = uniform_ptr(Bgu + (long)(PROWS * p) * a.ldb + k);
      const unsigned la = ldsA + b * LDS_BUF + p * LDS_PASS, lb = ldsB + b * LDS_BUF + p * LDS_PASS;
      unsigned va = aoffb[p], vb = boffb;
      if (H8) {
        va = sel(va, va + dk, upA);
        vb = sel(vb, vb + dk, upB);
      }
      if (p == 0) {
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(va), "s"(abase), "s"(la) : "memory");
        asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vb), "s"(bbase), "s"(lb) : "memory");
      } else {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(va), "s"(abase), "s"(la) : "memory");
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vb), "s"(bbase), "s"(lb) : "memory");
      }
    }
  };
  // One thread owns the queue, one tile ahead: lane 0 of wavefront 6.  fetch_next() pops the position of the NEXT tile (atomic) and reads
  // that tile's record into the other half of srec: two L2 round trips.  It runs in triangle step min(n - 1, 2) of the current tile, where
  // wave-row 3 (positions 3 and 7) has no products left: the wait costs no matrix instruction, and the step's barrier comes when the
  // other wave-rows have finished theirs.  (Round 4 had thread 0 do it in the tile's first K step: wavefront 0 reached that step's
  // barrier two round trips late and the other seven waited; 40 - 44 us per tile beyond its K steps at BASELINE config 5.  Keeping the
  // atomic's result in a register until the tile ends does not work: hipcc waits for it at once and spills it.)  acquire(), after the
  // K loop, then finds the record in LDS; only when a sequence has run out does it walk on to the next XCD's (a few times per block
  // and launch).
  constexpr unsigned REC_END = 0xffffffffu, REC_MORE = 0xfffffffeu;
  constexpr unsigned QOWNER = 64 * 6;  // lane 0 of wavefront 6 (wave-row 3)
  auto fetch = [&](int xq) { return __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(&a.queue[xq]), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  int cur = 1;  // srec[2 cur], srec[2 cur + 1] = the current tile's record
  auto fetch_next = [&]() {
    if (threadIdx.x == QOWNER) {
      const int steal = sst[0];
      uint4 r0 = make_uint4(REC_END, 0u, 0u, 0u), r1 = make_uint4(0u, 0u, 0u, 0u), r2 = make_uint4(0u, 0u, 0u, 0u);
      if (steal < 8) {
        const unsigned seq = fetch((xcd + steal) & 7);
        if (seq < (unsigned)sst[2]) {
          const uint4* rp = a.recs + (long)RW * (sst[1] + (long)seq);
          r0 = rp[0];
          r1 = rp[1];
          if (H8) r2 = rp[2];
        } else {
          r0.x = REC_MORE;
        }
      }
      srec[RW * (cur ^ 1)] = r0;
      srec[RW * (cur ^ 1) + 1] = r1;
      if (H8) srec[RW * (cur ^ 1) + 2] = r2;
    }
  };
  // (round 5: the record was written in a triangle step, several barriers ago: only the first call -- fetch_next() right before it --
  // and the walk to another XCD's sequence need a barrier of their own; the test on the LDS word is the same for every thread)
  auto acquire = [&](bool first) -> bool {  // block-uniform result
    const bool slow = first || __builtin_amdgcn_readfirstlane(srec[RW * (cur ^ 1)].x) == REC_MORE;
    if (slow) __syncthreads();  // every wavefront has read the word (the owner rewrites the record below) / the first record is published
    if (slow && threadIdx.x == QOWNER && srec[RW * (cur ^ 1)].x == REC_MORE) {
      uint4 r0 = make_uint4(REC_END, 0u, 0u, 0u), r1 = make_uint4(0u, 0u, 0u, 0u), r2 = make_uint4(0u, 0u, 0u, 0u);
      int steal = sst[0];
      while (++steal < 8) {
        const int xq = (xcd + steal) & 7;  // help the next XCD's sequence
        const int qlo = a.xoff[xq], qcnt = a.xoff[xq + 1] - qlo;
        const unsigned seq = fetch(xq);
        if (seq < (unsigned)qcnt) {
          const uint4* rp = a.recs + (long)RW * (qlo + (long)seq);
          r0 = rp[0];
          r1 = rp[1];
          if (H8) r2 = rp[2];
          sst[1] = qlo;
          sst[2] = qcnt;
          break;
        }
      }
      sst[0] = steal;
      srec[RW * (cur ^ 1)] = r0;
      srec[RW * (cur ^ 1) + 1] = r1;
      if (H8) srec[RW * (cur ^ 1) + 2] = r2;
    }
    if (slow) __syncthreads();
    cur ^= 1;
    return __builtin_amdgcn_readfirstlane(srec[RW * cur].x) != REC_END;
  };
  // the current tile's state: block- or wave-uniform values in scalar registers
  int tblk, rpos, n, ksec, erow[NAI];
  int erow1[NAI];    // H8: first row of the upper eight rows of group ai (8 h1), -1 when the position's h1 is missing; erow = 8 h0
  bool half_last;    // H8: the list's last position (w = n - 1) holds one 8-station tile only
  const mik_cu32_t* ksrc;  // the point block's list from this tile's first group on (16-byte aligned), as dwords in the CONSTANT address
                           // space: a uniform load from there is a scalar load (s_load_dword: no vector registers, no vmcnt); the list
                           // was written by an earlier kernel and is not modified during this one
  const double* Bgu;
  auto list_at = [&](int i) -> int { return H8 ? (int)ksrc[i] : (int)((ksrc[i >> 1] >> (16 * (i & 1))) & 0xffffu); };
  auto adopt = [&]() {  // srec -> the state above, first K tile into buffer 1 (nothing is waited for)
    const uint4 r0 = srec[RW * cur], r1 = srec[RW * cur + 1], r2 = H8 ? srec[RW * cur + 2] : srec[RW * cur + 1];
    const unsigned tile = __builtin_amdgcn_readfirstlane(r0.x);
    tblk = (int)(tile >> 10);
    rpos = (int)(tile & 1023u);
    const int nk = __builtin_amdgcn_readfirstlane((int)r0.y), g0 = 8 * rpos;
    const int kfirst = __builtin_amdgcn_readfirstlane((int)r0.z);
    n = nk - g0;  // K tiles of this tile: n - 8 off-diagonal ones, then its own min(n, 8) groups
    ksec = __builtin_amdgcn_readfirstlane((int)r0.w);
    const int ng = n < 8 ? n : 8;
    half_last = H8 && ((unsigned)kfirst >> 16) == 0xffffu;
    {  // byte offsets (relative to A_inv) of this thread's two staged rows
      const int tid = wave * 64 + lane_now(), lrow = tid >> 3, slot = tid & 7;
#pragma unroll
      for (int p = 0; p < NPASS; ++p) {
        const int R = lrow + PROWS * p, s = R >> 4;
        int gi = (s >> 1) + 4 * (s & 1);
        gi = gi < ng ? gi : ng - 1;  // a short last tile: the missing groups alias its last one (their accumulators stay zero)
        long grow;
        if (H8) {
          const unsigned pc = pair_of(r1, r2, gi), h0 = pc & 0xffffu, h1 = pc >> 16;
          grow = 8L * (long)(((R & 8) && h1 != 0xffffu) ? h1 : h0) + (R & 7);  // (a missing upper half aliases the lower one: left out of the epilogue)
        } else {
          grow = 16L * (long)group_of(r1, gi) + (R & 15);
        }
        // H8: the source slot WITHOUT its half bit -- stage() adds the upper half's column shift to the lanes of the upper half
        aoffb[p] = (unsigned)((grow * a.lda + (((H8 ? (slot & 3) : slot) ^ (lrow & 2)) << 1)) * 8);
      }
    }
#pragma unroll
    for (int ai = 0; ai < NAI; ++ai) {
      int gi = wm + 4 * ai;
      gi = gi < ng ? gi : ng - 1;
      if (H8) {
        const unsigned pc = pair_of(r1, r2, gi), h0 = pc & 0xffffu, h1 = pc >> 16;
        erow[ai] = __builtin_amdgcn_readfirstlane(8 * (int)h0);
        erow1[ai] = __builtin_amdgcn_readfirstlane(h1 == 0xffffu ? -1 : 8 * (int)h1);
      } else {
        erow[ai] = __builtin_amdgcn_readfirstlane(16 * (int)group_of(r1, gi));  // wave-uniform (wm)
      }
    }
    ksrc = (const mik_cu32_t*)(uintptr_t)(a.klist + (long)tblk * a.nK16 + (H8 ? 2 * g0 : g0));
    Bgu = uniform_ptr(a.Bt + (long)tblk * MIK_BN * a.ldb);
    stage(Bgu, H8 ? kfirst : 16 * kfirst, 1);
  };
  if (threadIdx.x == QOWNER) {
    const int lo = a.xoff[xcd];
    sst[0] = 0;
    sst[1] = lo;
    sst[2] = a.xoff[xcd + 1] - lo;
  }
  fetch_next();
  bool have = acquire(true);
  if (have) adopt();
  long pend_off = -1;  // the finished tile's row of part + its point block's first point: its sums wait in sred until the next barrier
  __shared__ double sred[NWM * 128];
  auto flush_sums = [&]() {
    if (wave < 2 && pend_off >= 0) {
      const int c = wave * 64 + lane_now();
      double v = 0.0;
#pragma unroll
      for (int x = 0; x < NWM; ++x) v += sred[x * 128 + c];
      a.part[pend_off + c] = v;
    }
  };
  __shared__ unsigned long long sprof[PROF ? 16 : 1];  // PROF: cycles and visits per triangle step w (wavefront 0's view)
  unsigned long long tstep = 0ULL;
  if (PROF && threadIdx.x < 16) sprof[threadIdx.x] = 0ULL;
  unsigned long long tp[10] = {0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL, 0ULL}, tm0 = 0ULL;
  auto mark = [&](int i) {  // PROF: the cycles since the previous mark go to phase i
    if (PROF) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      tp[i] += t - tm0;
      tm0 = t;
    }
  };
  if (PROF) tm0 = __builtin_amdgcn_s_memtime();
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
    flush_sums();
    const int fstep = n - 1 < 2 ? n - 1 : 2;  // the triangle step at which the queue's owner looks ahead (see fetch_next)
    mark(0);
    if (PROF) tp[8] += 1ULL, tp[9] += (unsigned long long)(n > 8 ? n - 8 : 0);
    for (; w >= 8; --w) {
      stage(Bgu, H8 ? kn : 16 * kn, buf ^ 1);
      int kn2 = 0;
      if (w >= 2) kn2 = list_at(w - 2);  // scalar load, in flight during this step's MFMAs
      const bool skip_hi = H8 && half_last && w == n - 1;  // an odd tail: only its lower eight stations exist
      const double* as = &sm.As[buf][0][0] + aoff;
      const double* bs = &sm.Bs[buf][0][0];
      const int bo[2] = {boff, boff_hi()};
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        if (H8 && m == 1 && skip_hi) continue;
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
    mark(1);
    if (PROF) tstep = __builtin_amdgcn_s_memtime();
    double cs[4] = {0.0, 0.0, 0.0, 0.0};  // EPI: this lane's sums over its rows of delta_ti W_it, points wn * 64 + bi * 16 + (lane & 15)
    for (; w >= 0; --w) {
      if (w >= 1) stage(Bgu, H8 ? kn : 16 * kn, buf ^ 1);
      int kn2 = 0;
      if (w >= 2) kn2 = list_at(w - 2);
      if (w == fstep) fetch_next();
      const bool skip_hi = H8 && half_last && w == n - 1;
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
          if (H8 && m == 1 && skip_hi) continue;
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
            // (sixteen LDS reads in one batch, into the registers the step's fragments have left, then four independent chains of four
            // multiply-adds: read - wait - multiply-add one by one cost this step 4000 cycles with everybody else at the barrier)
            const int ln = lane_now(), lq2 = ln >> 4, lc2 = ln & 15;
            double dv[4][4];
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
              const int pnt = wn * 64 + bi * 16 + lc2, sw = (pnt >> 1) & 7;  // B image: element (point, k) in slot (k >> 1) ^ sw of its row
              const double* brow = bs + pnt * MIK_BK;
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int k = 4 * r + lq2;
                dv[bi][r] = brow[(((k >> 1) ^ sw) << 1) | (k & 1)];
              }
            }
            // an odd tail has no upper eight rows: their K columns and their rows alias the lower eight (finite numbers), weight 0
            const double up = (H8 && erow1[ai] < 0) ? 0.0 : 1.0;
#pragma unroll
            for (int bi = 0; bi < 4; ++bi) {
              const double lo2 = dv[bi][0] * acc[ai][bi][0] + dv[bi][1] * acc[ai][bi][1];
              const double hi2 = dv[bi][2] * acc[ai][bi][2] + dv[bi][3] * acc[ai][bi][3];
              cs[bi] += lo2 + up * hi2;
            }
          }
      }
      drain();
      __syncthreads();
      buf ^= 1;
      kn = kn2;
      if (PROF && threadIdx.x == 0) {
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        sprof[w] += t - tstep;
        sprof[8 + w] += 1ULL;
        tstep = t;
      }
    }
    mark(2);
    // the next tile: record -> LDS (one barrier), its first K tile on the way to buffer 1 while this tile's epilogue runs
    const int t0 = tblk * MIK_BN, rp = rpos;
    int er[NAI], er1[NAI];
#pragma unroll
    for (int ai = 0; ai < NAI; ++ai) er[ai] = erow[ai], er1[ai] = H8 ? erow1[ai] : 0;
    have = acquire(false);
    mark(3);
    if (have) adopt();
    mark(4);
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
          for (int r = 0; r < 4; ++r) {
            if (!H8) bv[b2][ai * 4 + r] = brow[er[ai] + 4 * r];
            else if (r < 2) bv[b2][ai * 4 + r] = brow[er[ai] + 4 * r];                           // rows lq + 4 r < 8: the lower eight (8 h0 ..)
            else bv[b2][ai * 4 + r] = er1[ai] >= 0 ? brow[er1[ai] + 4 * r - 8] : 0.0;           // the upper eight (8 h1 ..), or none
          }
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
    if (PROF) {
      double keep = cs[0] + cs[1] + cs[2] + cs[3];
      asm volatile("" : "+v"(keep));  // (the sums exist before the mark)
    }
    mark(5);
    // the four wave-rows' sums meet in LDS; wavefronts 0 and 1 add them and store the tile's row of part BEHIND the next barrier the block
    // passes anyway (the top of the next tile, or the one behind the loop): no barrier and no wait of the other six for this
    if (lq == 0) {
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) sred[wm * 128 + wn * 64 + bi * 16 + lc] = cs[bi];
    }
    pend_off = (long)rp * a.palloc + t0;
    mark(6);
  }
  __syncthreads();
  flush_sums();
  if (PROF && (threadIdx.x & 63) == 0) {
#pragma unroll
    for (int i = 0; i < 10; ++i) a.prof[((long)blockIdx.x * 8 + wave) * 10 + i] = tp[i];
  }
  if (PROF && threadIdx.x < 16) a.prof[(long)gridDim.x * 80 + (long)blockIdx.x * 16 + threadIdx.x] = sprof[threadIdx.x];
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

}  // namespace mik
