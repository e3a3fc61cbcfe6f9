// tools/mik_k_experiments.h -- kernels that lost their A/B and left libmikrige.so in round 6 (the library instantiates none of them;
// tools/update_bench, tools/diag_probe and tools/kernel_bench still time them).  Include after mik_kernels.h / mik_k_inverse.h.
#pragma once
namespace mik {

// ---- Trailing update, DEEP form (round 5; option "update_deep") -------------------------------------------------------------------
// k_update's tile is a K = 128 MFMA loop followed by a read-modify-write of the 128 x 128 tile of T, and the two phases ADD: the T
// loads are issued after the loop (no registers for them at 128 VGPRs / 4 wavefronts per SIMD) and the two resident blocks of a CU
// fall into step.  Measured (profiles/r03_k2_panel_stream_ab.txt): 57 us per round of 512 resident blocks = 27 us of matrix-core
// time + 32 us of memory time.  This form gives a CU ONE block of 16 wavefronts (wave tile 16 x 64: 32 accumulator registers) and
//   * loads the block's T tile into registers BEFORE the K loop (32 more VGPRs): the epilogue is subtract + store, no load latency;
//   * stages the operands through FOUR LDS buffers (128 KB) with three K tiles in flight -- counted s_waitcnt vmcnt(N), raw
//     s_barrier: with one block per CU nothing else covers a DMA's latency;
//   * runs up to `tpb` tiles per block as ONE pipeline: the first K tiles of the next tile are in flight while the current one
//     finishes, its stores drain under the next tile's loop.
// Same K order per entry (K tiles from the top down, within a tile k = 8m + 2kq + h in the order (m, h)) and the same final
// subtraction as k_update: BIT-IDENTICAL inverses.  Only tiles that take a rank-128 update are handled here (part 4 / 0 without the
// pivot's own block row / column, which are copies: k_update part 5).
#define MIK_UD_NST 4
#define MIK_UD_LDS_BYTES (MIK_UD_NST * 2 * 128 * MIK_BK * 8)
__device__ __forceinline__ void ud_wait_vm(int n) {  // counted wait; n is block-uniform and one of a few values
  switch (n) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(36)" ::: "memory"); break;  // 4 + 16 stores + 16 T loads issued after the awaited stage
  }
}
template <bool SYM, int ABL = 0>
__global__ void __launch_bounds__(1024)
k_update_deep(double* __restrict__ T, long ld, int nblk, int kb, const double* __restrict__ Cold, const double* __restrict__ Cnew,
              const double* __restrict__ Rt, const double* __restrict__ Dinv, int part, int col, int tpb, int rev, int gdeep) {
  constexpr int abl = ABL;  // tools/update_bench only (0 in the library): 1 no T loads, 2 no LDS-DMA, 4 no MFMAs, 8 no stagger, 16 no fragment reads, 32 no stores
  extern __shared__ double ud_lds[];  // As[NST][128][16] then Bs[NST][128][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if ((int)blockIdx.x >= gdeep) {
    // the pivot's own block column / row (and Dinv): copies of the panels, one tile per block -- k_update's write-back path
    const int t = (int)blockIdx.x - gdeep;
    int iblk, jblk;
    if (SYM) {
      if (t >= nblk) return;
      iblk = t <= kb ? t : kb;
      jblk = t <= kb ? kb : t;
    } else if (t < nblk) {
      iblk = t;
      jblk = kb;
    } else {
      int j = t - nblk;
      if (j >= kb) ++j;
      if (j >= nblk) return;
      iblk = kb;
      jblk = j;
    }
    if (part == 4 && (jblk == col || (SYM && iblk == col))) return;  // the column part (k_update part 3) has written those
    const int i0 = iblk * 128, j0 = jblk * 128;
    for (int e = tid; e < 128 * 128; e += 1024) {
      const int r = e >> 7, c = e & 127;
      double v;
      if (iblk == kb && jblk == kb) v = Dinv[e];
      else if (jblk == kb) v = Cnew[(long)(i0 + r) * 128 + c];
      else v = Rt[(long)(j0 + c) * 128 + r];
      T[(long)(i0 + r) * ld + j0 + c] = v;
    }
    return;
  }
  const int wm = wave >> 1, wn = wave & 1, lq = lane >> 4, lc = lane & 15;
  const long total = SYM ? (long)nblk * (nblk + 1) / 2 : (long)nblk * nblk;
  const long per = (total + 7) / 8;
  const long xbeg = (long)(blockIdx.x % 8) * per, xend = xbeg + per < total ? xbeg + per : total;
  const long first = (long)(blockIdx.x / 8) * tpb;  // offset inside the XCD's range
  // position p (0 .. tpb - 1) of this block -> logical tile L (or -1), walked from the range's end when rev
  auto tile_at = [&](int p, int& iblk, int& jblk) -> bool {
    const long o = first + p;
    if (p >= tpb || xbeg + o >= xend) return false;
    const long L = rev ? xend - 1 - o : xbeg + o;
    if (SYM) {
      int j = (int)((sqrt(8.0 * (double)L + 1.0) - 1.0) * 0.5);
      while ((long)j * (j + 1) / 2 > L) --j;
      while ((long)(j + 1) * (j + 2) / 2 <= L) ++j;
      jblk = __builtin_amdgcn_readfirstlane(j);
      iblk = __builtin_amdgcn_readfirstlane((int)(L - (long)j * (j + 1) / 2));
    } else {
      iblk = __builtin_amdgcn_readfirstlane((int)(L / nblk));
      jblk = __builtin_amdgcn_readfirstlane((int)(L % nblk));
    }
    if (iblk == kb || jblk == kb) return false;                                                   // copies: k_update part 5
    if (part == 4 && (jblk == col || (SYM && iblk == col) || (iblk == col + 1 && jblk == col + 1))) return false;  // column part
    return true;
  };
  // the next valid position at or after p (tpb if none)
  auto next_valid = [&](int p, int& iblk, int& jblk) -> int {
    while (p < tpb && !tile_at(p, iblk, jblk)) ++p;
    return p;
  };
  // ---- staging (gemm_core's thread -> (row, slot) map, swizzles and LDS-DMA form; 1024 threads = one pass per operand)
  const int lrow = tid >> 3, slot = tid & 7;
  const unsigned aoffb = (unsigned)(((long)lrow * 128 + ((slot ^ (lrow & 2)) << 1)) * 8);
  const unsigned boffb = (unsigned)(((long)lrow * 128 + ((slot ^ ((lrow >> 1) & 7)) << 1)) * 8);
  constexpr unsigned BUF = 128 * MIK_BK * 8;  // bytes of one operand tile
  const unsigned ldsA = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(ud_lds + (size_t)wave * 8 * MIK_BK));
  const unsigned ldsB = ldsA + MIK_UD_NST * BUF;
  auto uniform_ptr = [](const double* q) {
    const unsigned long long v = (unsigned long long)(uintptr_t)q;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return reinterpret_cast<const double*>((uintptr_t)(((unsigned long long)hi << 32) | lo));
  };
  auto stage = [&](int iblk, int jblk, int kk, int buf) {  // K tile kk (0 = the top one, k = 112) of tile (iblk, jblk) into buffer buf
    const int k = 128 - MIK_BK * (kk + 1);
    const double* abase = uniform_ptr(Cold + (long)iblk * 128 * 128 + k);
    const double* bbase = uniform_ptr(Rt + (long)jblk * 128 * 128 + k);
    const unsigned la = ldsA + buf * BUF, lb = ldsB + buf * BUF;
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(aoffb), "s"(abase), "s"(la) : "memory");
    asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(boffb), "s"(bbase), "s"(lb) : "memory");
  };
  // ---- fragment offsets (doubles, inside one operand tile)
  const int kq = lane >> 4, ia = lane & 3, jb = lane & 15;
  int aoff[2], boff[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    aoff[m] = (wm * 16 + ia) * MIK_BK + (((4 * m + kq) ^ (ia & 2)) << 1);
    boff[m] = (wn * 64 + jb) * MIK_BK + (((4 * m + kq) ^ ((jb >> 1) & 7)) << 1);
  }
  const double* As = ud_lds;
  const double* Bs = ud_lds + (size_t)MIK_UD_NST * 128 * MIK_BK;
  // ---- two cursors over the block's sequence of (tile, K tile): the DMA runs NST - 1 steps ahead of the matrix cores
  int ci, cj, cp = next_valid(0, ci, cj);  // compute cursor: position, tile
  if (cp >= tpb) return;                   // block-uniform
  int si = ci, sj = cj, sp = cp, skk = 0;  // staging cursor
  int issued = 0;                          // steps staged so far
  auto stage_next = [&]() {                // stage the staging cursor's step and advance it (no-op at the end of the sequence)
    if (sp >= tpb) return;
    if (!(abl & 2)) stage(si, sj, skk, issued % MIK_UD_NST);
    ++issued;
    if (++skk == 128 / MIK_BK) {
      skk = 0;
      sp = next_valid(sp + 1, si, sj);
    }
  };
  auto tile_ptr = [&](int iblk, int jblk) { return T + (long)(iblk * 128 + wm * 16 + lq) * ld + jblk * 128 + wn * 64 + lc; };
  double tv[4][4] = {};
  auto load_t = [&](int iblk, int jblk) {
    const double* tp = tile_ptr(iblk, jblk);
#pragma unroll
    for (int bi = 0; bi < 4; ++bi)
#pragma unroll
      for (int r = 0; r < 4; ++r) tv[bi][r] = tp[(long)(4 * r) * ld + bi * 16];
  };
  if (!(abl & 1)) load_t(ci, cj);
#pragma unroll
  for (int s = 0; s < MIK_UD_NST - 1; ++s) stage_next();
  int step = 0;     // steps computed so far
  bool firsttile = true;
  const bool late = (wave >> 2) & 1;  // wave-uniform; SIMD = wave % 4 holds two early and two late wavefronts
  double2 fa[4], fb[4];
  d4 acc[4];
  auto read_frags = [&](const double* as, const double* bs, int m) {
#pragma unroll
    for (int x = 0; x < 4; ++x) fa[x] = *reinterpret_cast<const double2*>(as + aoff[m] + 4 * x * MIK_BK);
#pragma unroll
    for (int x = 0; x < 4; ++x) fb[x] = *reinterpret_cast<const double2*>(bs + boff[m] + 16 * x * MIK_BK);
  };
  auto mfma_all = [&]() {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) acc[bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[r].x, fb[bi].x, acc[bi][r], 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int bi = 0; bi < 4; ++bi) acc[bi][r] = __builtin_amdgcn_mfma_f64_4x4x4f64(fa[r].y, fb[bi].y, acc[bi][r], 0, 0, 0);
  };
  while (cp < tpb) {
#pragma unroll
    for (int y = 0; y < 4; ++y) acc[y] = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int kk = 0; kk < 128 / MIK_BK; ++kk, ++step) {
      // operations issued after the stage this step reads: the later stages in flight (2 DMAs each) and -- for the first three
      // steps of a tile that follows another -- that tile's 16 stores and this tile's 16 T loads
      const int ahead = issued - step - 1;  // 0 .. NST - 2
      ud_wait_vm((!firsttile && kk < MIK_UD_NST - 1) ? 36 : 2 * ahead);
      asm volatile("s_barrier" ::: "memory");  // every wavefront's share of the stage has landed; buffer (step - 1) % NST is free
      stage_next();
      const double* as = As + (size_t)(step % MIK_UD_NST) * 128 * MIK_BK;
      const double* bs = Bs + (size_t)(step % MIK_UD_NST) * 128 * MIK_BK;
      // STAGGER: with one barrier per K tile the 16 wavefronts of the block fall into step -- all read fragments, then all feed the
      // matrix cores, and the two phases add (measured: 32 us per tile against 13.7 of matrix-core time).  Half of the wavefronts
      // (two of the four on every SIMD) therefore run HALF A STEP LATE: they read the second half's fragments before the barrier
      // and contract them after it, so that one group reads while the other multiplies.  The order of the products of an entry
      // does not change.
      if (abl & 8) {
        if (!(abl & 16)) read_frags(as, bs, 0);
        if (!(abl & 4)) mfma_all();
        if (!(abl & 16)) read_frags(as, bs, 1);
        if (!(abl & 4)) mfma_all();
      } else {
      if (late && kk > 0) mfma_all();
      read_frags(as, bs, 0);
      mfma_all();
      read_frags(as, bs, 1);
      if (!late || kk == 128 / MIK_BK - 1) mfma_all();
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wavefront's fragment reads are done before it reaches the next barrier
    }
    // epilogue: T tile (in registers since before the K loop) - acc
    {
      double* tp = tile_ptr(ci, cj);
#pragma unroll
      for (int bi = 0; bi < 4; ++bi)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (!(abl & 32)) tp[(long)(4 * r) * ld + bi * 16] = tv[bi][r] - acc[bi][r];
    }
    cp = next_valid(cp + 1, ci, cj);
    firsttile = false;
    if (cp < tpb && !(abl & 1)) load_t(ci, cj);
  }
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

}  // namespace mik
