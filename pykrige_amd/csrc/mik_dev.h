// mik_kernels.h -- gfx950 (CDNA4) device code of the kriging execute() path.  fp64 throughout.
//
//   K1  k_assemble            kriging matrix A (or its SPD-shifted form) from station coordinates
//   K2  k_diag_inv_b, k_panel, k_update (+ k_piv_* for the pivoted path)   block Gauss-Jordan inverse, in place
//   K3a k_rhs                 right-hand sides b_g for a chunk of points (+ z_g = c.b_g), written point-major
//   K3b k_contract            sigma^2_g = -b_g^T A_inv b_g as a dense contraction on v_mfma_f64_4x4x4_4b_f64
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

// exp(x) for x <= 0 in 19 instructions (round 5): Cody-Waite reduction by ln 2 (the high part has 21 trailing zero bits: n ln2_hi is exact
// for |n| < 2^11), degree-13 Taylor polynomial on |r| <= ln 2 / 2 (remainder r^14 / 14! < 5e-18), v_ldexp_f64.  Within 1 ulp of the
// library's exp (tests/test_hip_parity.py::test_lean_exp...); that one is ~45 instructions plus its 64-bit literals, and the moving
// window's set-up evaluates it once per register-tile element (5 800 times per point at k = 100: half of the kernel's instructions).
// (The constants are handed to the FMAs as SCALAR register pairs -- inline asm, "s" operands: a v_fma_f64 takes no 64-bit literal, and left
// to itself hipcc materialises every coefficient with two v_mov_b32 per use: 26 vector moves per call beside 16 FMAs.)
__device__ __forceinline__ double fma_sc(double a, double b, double c_scalar) {  // a * b + c, c wave-uniform
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(c_scalar));
  return d;
}
// The sixteen constants, read ONCE per kernel from constant memory (scalar loads) and kept in scalar registers: materialised per call they were
// 28 s_mov_b32 beside the 19 vector instructions.
struct ExpTab {
  double log2e, ln2hi, ln2lo, c[13];  // c[k] = 1 / (13 - k)!  for k = 0 .. 12  (c[11] = 1/2, c[12] = 1 is folded: see below)
};
static __constant__ double MIK_EXP_TAB[16] = {1.44269504088896338700e+00, -6.93147180369123816490e-01, -1.90821492927058770002e-10,
                                              1.6059043836821613e-10, 2.08767569878681e-09, 2.505210838544172e-08, 2.755731922398589e-07,
                                              2.7557319223985893e-06, 2.48015873015873e-05, 1.984126984126984e-04, 1.388888888888889e-03,
                                              8.333333333333333e-03, 4.1666666666666664e-02, 1.6666666666666666e-01, 0.5, 1.0};
__device__ __forceinline__ ExpTab exp_tab_load() {
  ExpTab t;
  const double* q = MIK_EXP_TAB;
  t.log2e = q[0], t.ln2hi = q[1], t.ln2lo = q[2];
#pragma unroll
  for (int k = 0; k < 13; ++k) t.c[k] = q[3 + k];
  // opaque to the optimiser: otherwise it folds the known initialisers back into literals and re-materialises them at every use
  asm volatile("" : "+s"(t.log2e), "+s"(t.ln2hi), "+s"(t.ln2lo));
#pragma unroll
  for (int k = 0; k < 12; ++k) asm volatile("" : "+s"(t.c[k]));
  return t;
}
__device__ __forceinline__ double exp_neg_lean(double x, const ExpTab& T) {
  // one v_max_f64: far-apart / huge coordinates give d or d^2 = inf or 1e300 -> x = -inf would make the reduction inf - inf = NaN and (int)n is
  // undefined beyond 2^31; at -800 the result has already underflowed to 0 (gamma = sill, as libm's exp and the reference give)
  x = __builtin_fmax(x, -800.0);
  double t;
  asm("v_mul_f64 %0, %1, %2" : "=v"(t) : "s"(T.log2e), "v"(x));
  const double n = __builtin_rint(t);
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(n), "s"(T.ln2hi), "v"(x));
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(n), "s"(T.ln2lo), "v"(r));
  double p;
  asm("v_mul_f64 %0, %1, %2" : "=v"(p) : "s"(T.c[0]), "v"(r));   // r / 13!  (one scalar operand per instruction)
  asm("v_add_f64 %0, %1, %2" : "=v"(p) : "s"(T.c[1]), "v"(p));   // + 1 / 12!
#pragma unroll
  for (int k = 2; k < 12; ++k) p = fma_sc(p, r, T.c[k]);         // 1 / 11! .. 1 / 2!
  p = __builtin_fma(p, r, 1.0);                                    // (1.0 is an inline constant)
  p = __builtin_fma(p, r, 1.0);
  return ldexp(p, (int)n);
}
__device__ __forceinline__ double exp_neg_lean(double x) { return exp_neg_lean(x, exp_tab_load()); }

// FAST = the per-point right-hand-side path (5e9 evaluations at config 2, VALU-bound): divisions by
// the model constants become multiplications by their host-computed reciprocals (<= 1 ulp change of the
// exp argument; 1e-16 relative on gamma, tolerance is 1e-8).  FAST = false keeps the reference's operation
// order and is used where it is free (the N x N matrix assembly).
// LEAN (with FAST; the moving window's matrix set-up, round 5): exp_neg_lean for the two models that take an exponential.
template <int MODEL, bool FAST, bool LEAN = false>
__device__ __forceinline__ double vario(const Vario& v, double d, double d2, const ExpTab* tab = nullptr) {
  if (MODEL == 0) return v.p0 * d + v.p1;                               // linear   :25-29
  if (MODEL == 1) return v.p0 * pow(d, v.p1) + v.p2;                    // power    :32-37
  if (MODEL == 2) {                                                     // gaussian :40-45 (needs d^2 only)
    if (LEAN) return v.p0 * (1.0 - (tab ? exp_neg_lean(-d2 * v.c0inv, *tab) : exp_neg_lean(-d2 * v.c0inv))) + v.p2;
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
  if (MODEL == 4 && LEAN) return v.p0 * (1.0 - (tab ? exp_neg_lean(-d * v.c0inv, *tab) : exp_neg_lean(-d * v.c0inv))) + v.p2;
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

#define MIK_SP_MAXK16 4096  // K tiles a point block's list can hold in LDS (Mp <= 65536)

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

}  // namespace mik
