// mik_k_inverse.h -- K2 block Gauss-Jordan inverse, pivoted path, pseudo-inverses
// (one of the section headers mik_kernels.h is the umbrella of; every section is included by exactly one translation unit of the library)
#pragma once
#include "mik_dev.h"

namespace mik {

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
// The streams of the sweep are ordered by events only.  (Rounds 2-5 also carried flag-ordered schedules -- "early_diag" 4 / 5: a kernel
// of one stream waiting for a kernel of another -- which tools that serialise dispatches break; they lost their A/B at N >= 5000 and
// left the library in round 6 with their done-flags and counters: DESIGN_HISTORY.md.)
#define MIK_F_STRIDE 4096  // block columns a sweep can have (N x N matrices end long before 524 288 stations)
#define MIK_F_START 1
#define MIK_F_INTS (2 + MIK_F_STRIDE)

__device__ __forceinline__ void diag_started(int* flag, int k0) {
  if (threadIdx.x == 0) __hip_atomic_store(flag + MIK_F_START + k0 / 128, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_gate(const int* __restrict__ flag, int idx, int max_polls) {
  for (int i = 0; i < max_polls; ++i) {  // bounded: a late chain only costs this kernel's time, never a hang
    if (__hip_atomic_load(flag + MIK_F_START + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
    __builtin_amdgcn_s_sleep(8);
  }
}
// Out[i][n] = alpha * sum_m A[i][m] * Bt[n][m],  i over Mp rows, n < 128, m < 128 (one tile column)
// NAI: a block does 32 * NAI rows (4 waves as 2 x 2, wave tile 16 NAI x 64).  NAI = 4 is one 128 x 128 tile per block: 22 us, a
// CU's MFMA rate, whatever Mp is; NAI = 1 (round 3; the only form the library instantiates since round 6) spreads the same
// accumulation streams over 4 x the blocks -- the panel kernel sits on the update stream's critical path once per step.  Same k
// order per entry: same bits.
template <int NAI = 1>
__global__ void __launch_bounds__(256, 2)
k_panel(const double* __restrict__ A, long lda, const double* __restrict__ Bt, double alpha,
        double* __restrict__ Out, double* __restrict__ RtOut = nullptr, int k0 = 0,
        int* __restrict__ flag = nullptr, int gate_diag = -1) {
  // RtOut (symmetric sweep): also Rt[i][:] = -sigma_i Out[i][:], sigma_i = -1 for row blocks already swept
  // flag (early-diagonal schedule): gate_diag >= 0 -- block 0 leaves only when diagonal inverse gate_diag has started (k_gate's
  // hint without its launch: the update behind this kernel then finds that inverse already on its CU)
  constexpr int BMR = 32 * NAI;  // rows per block
  __shared__ GemmSmemT<BMR> sm;
  const int i0 = blockIdx.x * BMR;
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
        Out[(long)i * 128 + n] = v;
        if (RtOut) RtOut[(long)i * 128 + n] = (i < k0) ? v : -v;
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
// NAI = 4: 4 waves per block, wave tile 64 x 64 (228 VGPRs, 2 waves per SIMD); NAI = 2 (round 3; the only form the library
// instantiates since round 6): 8 waves, wave tile 32 x 64 (<= 128 VGPRs, 4 waves per SIMD to cover the short K loop and the
// read-modify-write epilogue).  Same accumulation order per entry: bit-identical results.
// register sets of the read-modify-write epilogue: two for the 4-wave form; the 8-wave form (128-VGPR budget) keeps ONE -- with two
// it spills 25 registers and the inverse is 12-14 % slower (N=5000 4.33 -> 4.90 ms; profiles/r03_k2_panel_stream_ab.txt)
#ifndef MIK_UPD_NTV
#define MIK_UPD_NTV(NAI) ((NAI) == 4 ? 2 : 1)
#endif
#ifdef MIK_UPD_PROF
__device__ unsigned long long* mik_upd_prof = nullptr;
#endif
template <bool SYM, int NAI = 2>
__global__ void __launch_bounds__(64 * 2 * (8 / NAI), NAI == 2 ? 4 : 2)
k_update(double* __restrict__ T, long ld, int nblk, int kb, const double* __restrict__ Cold,
         const double* __restrict__ Cnew, const double* __restrict__ Rt, const double* __restrict__ Dinv, int part, int col,
         double* __restrict__ Pout, double* __restrict__ Dcopy = nullptr, int rev = 0) {
  // rev (option "update_rev"): odd steps of the half sweep walk every XCD's tile range from its end -- the whole upper triangle is
  // streamed once per step, cyclically; a memory-side cache smaller than it keeps nothing of a cyclic stream, but most of a
  // back-and-forth one.
  // (Rounds 3-5 carried four more schedules of this kernel -- fp64 atomic adds instead of the read-modify-write, a host-written tile
  // map of 8 x 8 super-blocks, a per-CU token that made the two resident blocks alternate, a count of finished blocks for a waiting
  // kernel of the other stream.  All bit-identical, none faster: DESIGN_HISTORY.md.)
  // Pout (nullable): the updated block column `col` is ALSO written as the next step's column panel
  // P[row][0..127] (what k_copy_panel / k_copy_panel_sym would read back out of T: tiles of the block row `col` go in transposed),
  // so that the next panel chain starts with the diagonal inverse instead of a copy kernel.
  // Dcopy (nullable): the updated diagonal tile (col + 1, col + 1) is also left there (128 x 128): the early-diagonal chain
  // builds the diagonal block after next from it without touching T.
  __shared__ GemmSmem sm;
#ifdef MIK_UPD_PROF
  const unsigned long long pts = __builtin_amdgcn_s_memtime();
  const unsigned long long prs = __builtin_amdgcn_s_memrealtime();
#endif
  // part = 3 / 4 (round 3, the panel stream of the sweep): part 1 plus the diagonal tile (col + 1, col + 1) as block `nblk` of
  // the launch -- everything the next panel kernel AND the chain of the diagonal inverse after next read (Pout, Dcopy) -- /
  // part 2 without that tile.
  int iblk, jblk;
  if (part == 3 && (int)blockIdx.x == nblk) {
    if (col + 1 >= nblk) return;
    iblk = jblk = col + 1;
  } else if (part == 1 || part == 3) {
    if ((int)blockIdx.x >= nblk) return;
    if (SYM && (int)blockIdx.x > col) {
      iblk = col;
      jblk = blockIdx.x;
    } else {
      iblk = blockIdx.x;
      jblk = col;
    }
  } else if (SYM) {
    const long L = (rev && (kb & 1)) ? xcd_tile_rev((long)nblk * (nblk + 1) / 2) : xcd_tile((long)nblk * (nblk + 1) / 2);
    if (L < 0) return;
    jblk = (int)((sqrt(8.0 * (double)L + 1.0) - 1.0) * 0.5);
    while ((long)jblk * (jblk + 1) / 2 > L) --jblk;            // guard the float estimate
    while ((long)(jblk + 1) * (jblk + 2) / 2 <= L) ++jblk;
    iblk = (int)(L - (long)jblk * (jblk + 1) / 2);             // i <= j: the upper block triangle
    if ((part == 2 || part == 4) && (iblk == col || jblk == col)) return;
    if (part == 4 && iblk == col + 1 && jblk == col + 1) return;
  } else {
    const long L = xcd_tile((long)nblk * nblk);
    if (L < 0) return;
    iblk = (int)(L / nblk);
    jblk = (int)(L % nblk);
    if ((part == 2 || part == 4) && jblk == col) return;
    if (part == 4 && iblk == col + 1 && jblk == col + 1) return;
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
    return;
  }
  d4 acc[NAI][4];
#pragma unroll
  for (int x = 0; x < NAI; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[x][y] = (d4){0.0, 0.0, 0.0, 0.0};
#ifdef MIK_UPD_PROF  // tools/update_bench only: s_memtime at the phase boundaries of one tile, per block
  const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
  gemm_core<NAI>(Cold + (long)i0 * 128, 128, Rt + (long)j0 * 128, 128, 0, 128, acc, sm);
#ifdef MIK_UPD_PROF
  const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
#endif
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wm = wave >> 1, wn = wave & 1, lq = lane >> 4, lc = lane & 15;
  constexpr int WR = 16 * NAI;  // rows of the wave tile
  // read-modify-write in batches of 16 independent loads, the NEXT batch's loads in flight while this one is subtracted and
  // stored (two register sets): the epilogue pays the memory latency once, not four times
  constexpr int NTV = MIK_UPD_NTV(NAI);  // register sets of the epilogue (the 8-wave form has a 128-VGPR budget for 4 waves per SIMD)
  double tv[NTV][4][4];
  auto tile_ptr = [&](int ai) { return T + (long)(i0 + wm * WR + ai * 16 + lq) * ld + j0 + wn * 64 + lc; };
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
#ifdef MIK_UPD_PROF
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long pt2 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && mik_upd_prof) {
    unsigned long long* o = mik_upd_prof + 4L * blockIdx.x;
    unsigned xcc_id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
    o[0] = pt0, o[1] = pt1, o[2] = pt2, o[3] = ((pt0 - pts) << 8) | (xcc_id & 7);
    if (blockIdx.x == 0) {  // calibration: s_memrealtime counts at 100 MHz
      const unsigned long long pre = __builtin_amdgcn_s_memrealtime();
      mik_upd_prof[4L * gridDim.x] = pt2 - pts, mik_upd_prof[4L * gridDim.x + 1] = pre - prs;
      mik_upd_prof[4L * gridDim.x + 2] = prs;
    }
  }
#endif
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


}  // namespace mik
