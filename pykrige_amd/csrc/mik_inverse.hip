// mik_inverse.hip -- K2: the factor path (block Gauss-Jordan sweep and its schedules, probes, deflated / null-space / Jacobi pseudo-inverses)
// One translation unit of libmikrige.so (pykrige_amd/build.py compiles them in parallel).
#include "mik_k_inverse.h"
#include "mik_host.h"


// Moore-Penrose pseudo-inverse of the assembled matrix in T (leading M x M block, row length Mp; the padding columns of
// those rows are zero), in place.  Cyclic one-sided Jacobi until every row pair is orthogonal to 1e-15, then B^T D W.
// The general pseudo-inverse by a BLOCK one-sided Jacobi (round 4; kernels and algebra: mik_kernels.h k_bj_*).  Same result as
// run_pseudo_inverse_scalar below -- B = W A with mutually orthogonal rows, pinv(A) = B^T diag(1 / sigma_i^2 | sigma_i > M eps sigma_max) W
// -- from ~3 M / 32 passes over the matrix per sweep instead of ~2 M.
static int run_pseudo_inverse_scalar(mik_handle* h);
static int run_pseudo_inverse(mik_handle* h) {
  // measured (profiles/r04_pseudo_inverse_block_jacobi.txt): M = 501 100 ms against 55 ms scalar, M = 1001 277 / 255, M = 2001 615 / 1180,
  // M = 4001 1.5 s / 9.3 s -- the block form from 1536 rows on unless the caller says otherwise
  if (h->opt_pinv_block == 0 || (h->opt_pinv_block < 0 && h->M < 1536)) return run_pseudo_inverse_scalar(h);
  const int n = h->M;
  const long ld = h->Mp;
  int nb = (n + MIK_BJ_B - 1) / MIK_BJ_B;
  nb += nb & 1;
  if (nb < 2) nb = 2;
  const int npairs = nb / 2;
  DevBuf W, out, sig, worst, order, qbuf, active;
  MIKC(W.ensure(sizeof(double) * (size_t)n * ld));
  MIKC(out.ensure(sizeof(double) * (size_t)n * ld));
  MIKC(sig.ensure(sizeof(double) * (size_t)n));
  MIKC(worst.ensure(sizeof(unsigned long long)));
  MIKC(order.ensure(sizeof(int) * (size_t)nb * MIK_BJ_B));
  MIKC(qbuf.ensure(sizeof(double) * 64 * 64 * (size_t)npairs));
  MIKC(active.ensure(sizeof(int) * (size_t)npairs));
  double* B = h->T.as<double>();
  hipLaunchKernelGGL(k_set_identity, dim3((unsigned)(((long)n * ld + 255) / 256)), dim3(256), 0, h->stream, W.as<double>(), ld, n);
  std::vector<double> s2(n), d(n);
  std::vector<int> ord((size_t)nb * MIK_BJ_B);
  const double eps = 2.220446049250313e-16;
  const size_t lds = sizeof(double) * 2 * 64 * MIK_BJ_LD;
  HIPC(hipFuncSetAttribute((const void*)k_bj_eig, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int nslice = std::max(1, std::min(16, n / 256));  // column slices of the Gram pass: 62 pairs alone leave three quarters of the CUs idle
  DevBuf gpart;
  MIKC(gpart.ensure(sizeof(double) * 64 * 64 * (size_t)npairs * nslice));
  bool converged = false;
  int sweeps = 0;
  double last_off = 1.0;
  // orthogonal to 4e-15: the cosines themselves are 4000-term sums -- their rounding noise sits at 1e-15 and the iteration would
  // chase it for sweeps (measured at M = 4001: 1.4e-15, 1.0e-15, 0.999e-15 in the last three of 19 sweeps;
  // 2.0e-15, 1.98e-15 after the Gram sums were regrouped)
  const double bj_tol = 4e-15;
  for (int sweep = 0; sweep < 40 && !converged; ++sweep, ++sweeps) {
    const int max_inner = last_off > 1e-3 ? 3 : 30;
    hipLaunchKernelGGL(k_rownorm2, dim3(n), dim3(256), 0, h->stream, (const double*)B, ld, n, sig.as<double>());
    HIPC(hipMemcpyAsync(s2.data(), sig.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    double smax2 = 0.0;
    for (double v : s2) smax2 = std::max(smax2, std::isfinite(v) ? v : 0.0);
    if (!(smax2 > 0.0)) return fail(MIK_ESINGULAR, "pseudo-inverse: the matrix is zero or not finite");
    // rows sorted by norm, largest first (ties by index: deterministic), cut into blocks of 32, padded with -1
    for (int i = 0; i < n; ++i) ord[(size_t)i] = i;
    std::stable_sort(ord.begin(), ord.begin() + n, [&](int a, int b) { return s2[(size_t)a] > s2[(size_t)b]; });
    for (size_t i = (size_t)n; i < ord.size(); ++i) ord[i] = -1;
    HIPC(hipMemcpyAsync(order.p, ord.data(), sizeof(int) * ord.size(), hipMemcpyHostToDevice, h->stream));
    HIPC(hipMemsetAsync(worst.p, 0, sizeof(unsigned long long), h->stream));
    // rows below a hundredth of the cut-off M eps sigma_max are the null space: their angles are rounding noise
    const double dead2 = (0.01 * (double)n * eps) * (0.01 * (double)n * eps) * smax2;
    for (int round = 0; round < nb - 1; ++round) {
      hipLaunchKernelGGL(k_bj_gram, dim3(npairs, nslice), dim3(64), 0, h->stream, (const double*)B, ld, n, (const int*)order.as<int>(), nb, round,
                         nslice, gpart.as<double>());
      hipLaunchKernelGGL(k_bj_eig, dim3(npairs), dim3(256), lds, h->stream, (const double*)gpart.as<double>(), nslice, dead2, bj_tol, max_inner,
                         qbuf.as<double>(), active.as<int>(), worst.as<unsigned long long>());
      hipLaunchKernelGGL(k_bj_rotate, dim3(npairs, (unsigned)((n + 63) / 64), 2), dim3(256), 0, h->stream, B, W.as<double>(), ld, n,
                         (const int*)order.as<int>(), nb, round, (const double*)qbuf.as<double>(), (const int*)active.as<int>());
    }
    HIPC(hipGetLastError());
    unsigned long long bits = 0;
    HIPC(hipMemcpyAsync(&bits, worst.p, sizeof bits, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    double off;
    memcpy(&off, &bits, sizeof off);
    converged = off < bj_tol;
    last_off = off;
    if (getenv("MIK_DEBUG_PINV")) fprintf(stderr, "block Jacobi sweep %d: largest cosine between live rows %.3e (inner sweeps <= %d)\n", sweep, off, max_inner);
  }
  if (!converged) return fail(MIK_ESINGULAR, "pseudo-inverse: block Jacobi iteration did not converge");
  h->tm.null_dim = 0;
  hipLaunchKernelGGL(k_rownorm2, dim3(n), dim3(256), 0, h->stream, (const double*)B, ld, n, sig.as<double>());
  HIPC(hipMemcpyAsync(s2.data(), sig.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  double smax = 0.0;
  for (double v : s2) smax = std::max(smax, sqrt(v));
  const double cut = (double)n * eps * smax;  // scipy.linalg.pinv / pinvh: rtol = max(M, N) * eps
  for (int i = 0; i < n; ++i) d[i] = (sqrt(s2[i]) > cut) ? 1.0 / s2[i] : 0.0;
  HIPC(hipMemcpyAsync(sig.p, d.data(), sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  const unsigned tiles = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL(k_pinv_gemm, dim3(tiles, tiles), dim3(256), 0, h->stream, (const double*)B, (const double*)W.as<double>(),
                     (const double*)sig.as<double>(), ld, n, out.as<double>());
  HIPC(hipMemsetAsync(h->T.p, 0, h->T.bytes, h->stream));
  HIPC(hipMemcpy2DAsync(h->T.p, sizeof(double) * ld, out.p, sizeof(double) * ld, sizeof(double) * n, n, hipMemcpyDeviceToDevice,
                        h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  HIPC(hipGetLastError());
  return MIK_OK;
}

// the scalar form (rounds 1-3; option "pinv_block" 0): one row pair per block, one launch per round of the tournament
static int run_pseudo_inverse_scalar(mik_handle* h) {
  const int n = h->M, m = n + (n & 1);
  const long ld = h->Mp;
  DevBuf W, out, sig, maxoff;
  MIKC(W.ensure(sizeof(double) * (size_t)n * ld));
  MIKC(out.ensure(sizeof(double) * (size_t)n * ld));
  MIKC(sig.ensure(sizeof(double) * (size_t)n));
  MIKC(maxoff.ensure(sizeof(unsigned long long)));
  double* B = h->T.as<double>();
  hipLaunchKernelGGL(k_set_identity, dim3((unsigned)(((long)n * ld + 255) / 256)), dim3(256), 0, h->stream, W.as<double>(), ld, n);
  std::vector<double> s2(n), d(n);
  hipLaunchKernelGGL(k_rownorm2, dim3(n), dim3(256), 0, h->stream, (const double*)B, ld, n, sig.as<double>());
  HIPC(hipMemcpyAsync(s2.data(), sig.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  double fro2 = 0.0;
  for (double v : s2) fro2 += v;  // |A|_F^2 = sum sigma_i^2, invariant under the rotations; sigma_max^2 >= fro2 / n
  const double eps = 2.220446049250313e-16;
  const double dead2 = 0.01 * ((double)n * eps) * ((double)n * eps) * fro2 / (double)n;
  bool converged = false;
  for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
    HIPC(hipMemsetAsync(maxoff.p, 0, sizeof(unsigned long long), h->stream));
    for (int step = 0; step < m - 1; ++step)
      hipLaunchKernelGGL(k_jac_step, dim3(m / 2), dim3(256), 0, h->stream, B, W.as<double>(), ld, n, m, step, dead2,
                         maxoff.as<unsigned long long>());
    unsigned long long bits = 0;
    HIPC(hipMemcpyAsync(&bits, maxoff.p, sizeof bits, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    double off;
    memcpy(&off, &bits, sizeof off);
    converged = off < 1e-15;
  }
  if (!converged) return fail(MIK_ESINGULAR, "pseudo-inverse: Jacobi iteration did not converge");
  hipLaunchKernelGGL(k_rownorm2, dim3(n), dim3(256), 0, h->stream, (const double*)B, ld, n, sig.as<double>());
  HIPC(hipMemcpyAsync(s2.data(), sig.p, sizeof(double) * n, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  double smax = 0.0;
  for (double v : s2) smax = std::max(smax, sqrt(v));
  const double cut = (double)n * eps * smax;  // scipy.linalg.pinv / pinvh: rtol = max(M, N) * eps
  for (int i = 0; i < n; ++i) d[i] = (sqrt(s2[i]) > cut) ? 1.0 / s2[i] : 0.0;
  HIPC(hipMemcpyAsync(sig.p, d.data(), sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  const unsigned tiles = (unsigned)((n + 63) / 64);
  hipLaunchKernelGGL(k_pinv_gemm, dim3(tiles, tiles), dim3(256), 0, h->stream, (const double*)B, (const double*)W.as<double>(),
                     (const double*)sig.as<double>(), ld, n, out.as<double>());
  HIPC(hipMemsetAsync(h->T.p, 0, h->T.bytes, h->stream));
  HIPC(hipMemcpy2DAsync(h->T.p, sizeof(double) * ld, out.p, sizeof(double) * ld, sizeof(double) * n, n, hipMemcpyDeviceToDevice,
                        h->stream));
  HIPC(hipStreamSynchronize(h->stream));  // W / out / sig are released at scope exit
  HIPC(hipGetLastError());
  return MIK_OK;
}

// T's lower block triangle from its upper one, on `st` of the current device: a group member / rank that received the packed upper block
// triangle of an EXACTLY symmetric inverse (the half sweep mirrors its triangle, every other device path ends in k_symmetrize) rebuilds the
// matrix the leader holds, bit for bit
int launch_mirror_upper(double* T, long Mp, hipStream_t st) {
  hipLaunchKernelGGL(k_mirror_upper, dim3((unsigned)(Mp / 64), (unsigned)(Mp / 64)), dim3(256), 0, st, T, Mp, (int)(Mp / 64));
  HIPC(hipGetLastError());
  return MIK_OK;
}

int ensure_factor_buffers(mik_handle* h) {
  const size_t Mp = h->Mp;
  MIKC(h->T.ensure(sizeof(double) * Mp * Mp));
  MIKC(h->cvec.ensure(sizeof(double) * Mp));
  return MIK_OK;
}

static void launch_diag_inv(mik_handle* h, hipStream_t st, const double* T, long ld, int k0, int nspd, double* dinv, double* dinvT,
                            bool own_cu = false) {
  // k_diag_inv_b (round 3): 86 KB of LDS of its own, padded to 100 KB when it wants the CU to itself (no 64-KB update block fits beside it)
  const int lds = own_cu ? 100 * 1024 : (int)(sizeof(double) * MIK_DIAGB_LDS_DOUBLES);
  // per launch: the attribute belongs to the function object of the CURRENT device (device groups factor on several)
  (void)hipFuncSetAttribute((const void*)k_diag_inv_b<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipLaunchKernelGGL(k_diag_inv_b<0>, dim3(1), dim3(256), lds, st, T, ld, k0, nspd, dinv, dinvT, h->flag.as<int>());
}

// unpivoted (path 1) or pivoted (path 2) block Gauss-Jordan on T in place
static int run_block_inverse(mik_handle* h, bool pivoted, int nspd, int* flag_out) {
  const int Mp = h->Mp, nblk = Mp / 128;
  const size_t panel = sizeof(double) * (size_t)Mp * 128;
  MIKC(h->Cold.ensure(panel));
  MIKC(h->Cnew.ensure(panel));
  MIKC(h->Rt.ensure(panel));
  MIKC(h->Dinv.ensure(sizeof(double) * 128 * 128));
  MIKC(h->DinvT.ensure(sizeof(double) * 128 * 128));
  if (nblk > MIK_F_STRIDE) return fail(MIK_EINVAL, "more block columns than the sweep's flag layout holds");
  MIKC(h->flag.ensure(sizeof(int) * (size_t)MIK_F_INTS));  // layout: MIK_F_* in mik_kernels.h
  HIPC(hipMemsetAsync(h->flag.p, 0, sizeof(int) * (size_t)MIK_F_INTS, h->stream));
  const int ncand = Mp / 32;  // one candidate per 32-row block of the pivot-search panel (MIK_PIV_ROWS)
  if (pivoted) {
    MIKC(h->TKt.ensure(panel));
    MIKC(h->P0.ensure(panel));
    MIKC(h->P1.ensure(panel));
    MIKC(h->cand0.ensure(sizeof(PivCand) * ncand));
    MIKC(h->cand1.ensure(sizeof(PivCand) * ncand));
    MIKC(h->pivall.ensure(sizeof(int) * Mp));
  }
  double* T = h->T.as<double>();
  const long ld = Mp;
  const long tiles = (long)nblk * nblk;
  const unsigned pgrid = (unsigned)(((long)Mp * 128 + 255) / 256);
  // measured (scripts/inverse_lookahead_ab.py): +16 % at 16 block columns (the second stream's waits cost more than the
  // overlap returns), -12 % at 40, -17 % at 63
  // half sweep (upper block triangle only): on request, or by itself for the two variograms whose measured error stays three
  // orders inside the 1e-8 / 1e-6 bar (exponential, spherical: profiles/r02_sweep_vs_pivoted_vs_half_sweep_accuracy.txt and the
  // full-size fixtures) and from 24 block columns on, where it pays
  const bool symsweep = !pivoted && (h->opt_symsweep > 0 || (h->opt_symsweep < 0 && !h->no_half_sweep && (h->model == 3 || h->model == 4) && nblk >= 24));
  h->last_half_sweep = symsweep;
  const long ltiles = symsweep ? (long)nblk * (nblk + 1) / 2 : tiles;
  const unsigned ug = (unsigned)(8 * ((ltiles + 7) / 8));
  // "update_rev": odd steps of the half sweep walk the triangle backwards (k_update rev)
  const int urev = (h->opt_update_rev < 0 ? nblk >= 45 : h->opt_update_rev != 0) ? 1 : 0;
  // the panel kernel over all Mp rows, 32 rows per block (k_panel<1>); the trailing update on 8 wavefronts per tile (k_update<*, 2>)
#define PANEL(STREAM, ...) hipLaunchKernelGGL((k_panel<1>), dim3(4 * nblk), dim3(256), 0, STREAM, __VA_ARGS__)
#define UPDK(SYMV, GRID, STREAM, ...) hipLaunchKernelGGL((k_update<SYMV, 2>), GRID, dim3(512), 0, STREAM, __VA_ARGS__, urev)
#define UPDX(GRID, STREAM, CO, CN, R, D, PART, COL, POUT, DCOPY)                                                             \
  do {                                                                                                                       \
    if (symsweep)                                                                                                            \
      UPDK(true, GRID, STREAM, T, ld, nblk, kb, (const double*)(CO), (const double*)(CN),                                    \
           (const double*)(R), (const double*)(D), PART, COL, POUT, DCOPY);                                                  \
    else                                                                                                                     \
      UPDK(false, GRID, STREAM, T, ld, nblk, kb, (const double*)(CO), (const double*)(CN),                                   \
           (const double*)(R), (const double*)(D), PART, COL, POUT, DCOPY);                                                  \
  } while (0)
#define UPD(GRID, STREAM, CO, CN, R, D, PART, COL, POUT) UPDX(GRID, STREAM, CO, CN, R, D, PART, COL, POUT, (double*)nullptr)
  const bool lookahead = h->opt_lookahead < 0 ? nblk >= 3 : h->opt_lookahead != 0;
  // measured (profiles/r02_inverse_timeline.txt): with up to ~2400 update tiles per step (N=5000 full sweep: 1600, N=8000 half
  // sweep: 2016) the serial chain is the step period and giving its head a CU of its own pays (-14 % / -10 %); with 3969 tiles
  // (N=8000 full sweep) the update is, and holding it back costs 3 %
  const bool gate = h->opt_gate < 0 ? ltiles <= 2400 : h->opt_gate != 0;
  if (!pivoted && nblk > 1 && lookahead) {
    // Look-ahead sweep.  Step kb's update is split: block column kb+1 first (nblk tiles), then -- on the second stream --
    // the whole panel chain of step kb+1 (diagonal inverse, panel copy, C_new, R^T; a serial ~160 us on few CUs) runs
    // while the first stream finishes the other nblk^2 - nblk tiles of step kb.  Two panel sets alternate.
    MIKC(h->Cold2.ensure(panel));
    MIKC(h->Cnew2.ensure(panel));
    MIKC(h->Rt2.ensure(panel));
    MIKC(h->Dinv2.ensure(sizeof(double) * 128 * 128));
    MIKC(h->DinvT2.ensure(sizeof(double) * 128 * 128));
    while (h->la_events.size() < 2 * (size_t)nblk + 2) {
      hipEvent_t e;
      HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h->la_events.push_back(e);
    }
    double* cold[2] = {h->Cold.as<double>(), h->Cold2.as<double>()};
    double* cnew[2] = {h->Cnew.as<double>(), h->Cnew2.as<double>()};
    double* rt[2] = {h->Rt.as<double>(), h->Rt2.as<double>()};
    double* dinv[2] = {h->Dinv.as<double>(), h->Dinv2.as<double>()};
    double* dinvT[2] = {h->DinvT.as<double>(), h->DinvT2.as<double>()};
    // the serial chain of a step: diagonal inverse -> [panel copy, unless the column update already left it in Cold] -> panel
    // kernel (C_new and R^T in one launch)
    auto panel_chain = [&](hipStream_t st, int kb, int set, bool have_cold) {
      const int k0 = kb * 128;
      launch_diag_inv(h, st, (const double*)T, ld, k0, nspd, dinv[set], dinvT[set], gate && st == h->stream2);
      if (!have_cold) {
        if (symsweep) hipLaunchKernelGGL(k_copy_panel_sym, dim3(Mp / 64), dim3(256), 0, st, (const double*)T, ld, k0, Mp, cold[set]);
        else hipLaunchKernelGGL(k_copy_panel, dim3(pgrid), dim3(256), 0, st, (const double*)T, ld, k0, Mp, cold[set]);
      }
      PANEL(st, (const double*)cold[set], 128L, (const double*)dinvT[set], -1.0, cnew[set], rt[set], k0, (int*)nullptr, -1);
    };
    panel_chain(h->stream, 0, 0, false);
    {
      // Early-diagonal schedule.  What the next diagonal inverse needs of step kb is ONE tile, D(kb+1) - C_b R_b^T, and that takes
      // only the 128 panel rows of block kb + 1.  The second stream therefore runs, per step,
      //     [wait: update kb-1 done]  k_gemm128<0> (R_b = C_b Dinv) -> k_gemm128<1> (the tile) -> diagonal inverse kb+1
      // from the column panel and the diagonal-tile copy (two alternate) that update kb-1 left behind (it never reads T), while
      // the first stream runs  [wait: diagonal inverse kb done]  k_panel (all rows) -> the WHOLE update of step kb  -- one
      // launch, no split.  The serial chain (diagonal inverse + two 6-us products spread over 64 blocks) no longer contains the
      // full panel kernel, the block-column update or a second cross-stream wait, and the diagonal inverse overlaps the update.
      // Same accumulation order per entry as k_panel / k_update: the inverse is bit-identical.
      MIKC(h->Dnext.ensure(sizeof(double) * 128 * 128));
      MIKC(h->Dcopy.ensure(sizeof(double) * 2 * 128 * 128));
      MIKC(h->Rb.ensure(sizeof(double) * 128 * 128));
      double* dnext = h->Dnext.as<double>();
      double* dcopy[2] = {h->Dcopy.as<double>(), h->Dcopy.as<double>() + 128 * 128};  // [kb & 1] is read by step kb's chain
      double* rb = h->Rb.as<double>();
      HIPC(hipMemcpy2DAsync(dcopy[0], sizeof(double) * 128, T + 128L * ld + 128, sizeof(double) * ld, sizeof(double) * 128, 128,
                            hipMemcpyDeviceToDevice, h->stream));  // tile (1, 1) as assembled: the second stream never reads T
      HIPC(hipEventRecord(h->la_events[0], h->stream));  // "update -1": the first panel set and diagonal inverse are there
      HIPC(hipStreamWaitEvent(h->stream2, h->la_events[0], 0));
      // The streams are ordered by events only (a satisfied hipStreamWaitEvent costs ~12 us of barrier-packet latency per step and
      // stream; the flag-ordered variants that avoided it -- a kernel waiting for a kernel of another stream -- break under tools
      // that serialise dispatches and left the library in round 6).  What IS folded into k_panel is k_gate's poll: a hint with a
      // bounded wait, harmless when serialised.
      int* fl = h->flag.as<int>();
      const bool pstream = h->opt_panel_stream < 0 ? nblk >= 24 : h->opt_panel_stream != 0;
      if (pstream) {
        // Panel-stream schedule (round 3).  The update stream of the schedule below carries k_panel + the whole update, one after
        // the other, and from ~4000 stations on it is the step period.  Here the update of a step is cut into the tiles the NEXT
        // step's head reads -- block column / row kb + 1 and the diagonal tile (kb + 2, kb + 2): "column part", k_update part 3 --
        // and the rest (part 4), and three streams run
        //   s1:  [panel kb ready]                          rest of update kb
        //   s3:  [diagonal inverse kb]  k_panel kb  ->  [rest kb-1 done]  column part of update kb
        //   s2:  [column part kb-1 done]  two 128^3 products -> diagonal inverse kb+1           (as below)
        // so that s1 is trailing updates back to back and the panel kernel (26 us at a tenth of the MFMA rate) and the small
        // column launch overlap them.  Only events order the streams.  The rest of update kb-1 still reads the diagonal inverse
        // kb-1 while kb+1 is being formed: three Dinv sets.  Same tiles, same kernels, same accumulation order: same bits.
        MIKC(h->Dinv3.ensure(sizeof(double) * 128 * 128));
        MIKC(h->DinvT3.ensure(sizeof(double) * 128 * 128));
        while (h->ps_events.size() < 4 * (size_t)nblk + 4) {
          hipEvent_t e;
          HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          h->ps_events.push_back(e);
        }
        double* dv[3] = {h->Dinv.as<double>(), h->Dinv2.as<double>(), h->Dinv3.as<double>()};
        double* dvT[3] = {h->DinvT.as<double>(), h->DinvT2.as<double>(), h->DinvT3.as<double>()};
        auto evD = [&](int kb) { return h->ps_events[4 * kb]; };      // diagonal inverse kb done (s2)
        auto evP = [&](int kb) { return h->ps_events[4 * kb + 1]; };  // panel kb done (s3)
        auto evC = [&](int kb) { return h->ps_events[4 * kb + 2]; };  // column part of update kb done (s3)
        auto evR = [&](int kb) { return h->ps_events[4 * kb + 3]; };  // rest of update kb done (s1)
        hipStream_t s1 = h->stream, s2 = h->stream2, s3 = h->stream3;
        HIPC(hipStreamWaitEvent(s3, h->la_events[0], 0));  // panel set 0, diagonal inverse 0 (dv[0]) and dcopy[0] are there
        for (int kb = 0; kb < nblk; ++kb) {
          const int set = kb & 1, k0 = kb * 128, k1 = k0 + 128, d3 = kb % 3, d3n = (kb + 1) % 3;
          if (kb + 1 < nblk) {  // s2: diagonal inverse kb + 1
            if (kb > 0) HIPC(hipStreamWaitEvent(s2, evC(kb - 1), 0));
            hipLaunchKernelGGL(k_gemm128<0>, dim3(64), dim3(256), 0, s2, (const double*)(cold[set] + (long)k1 * 128), (const double*)dvT[d3],
                               -1.0, (const double*)nullptr, 0L, rb);
            hipLaunchKernelGGL(k_gemm128<1>, dim3(64), dim3(256), 0, s2, (const double*)(cold[set] + (long)k1 * 128), (const double*)rb, 0.0,
                               (const double*)dcopy[set], 128L, dnext);
            const double* dview = (const double*)((uintptr_t)dnext - sizeof(double) * ((size_t)k1 * 128 + (size_t)k1));
            launch_diag_inv(h, s2, dview, 128L, k1, nspd, dv[d3n], dvT[d3n], gate);
            HIPC(hipEventRecord(evD(kb + 1), s2));
          }
          if (kb > 0) {  // s3: panel kb (its column panel was left by the column part of update kb - 1, on this stream)
            HIPC(hipStreamWaitEvent(s3, evD(kb), 0));
            PANEL(s3, (const double*)cold[set], 128L, (const double*)dvT[d3], -1.0, cnew[set], rt[set], k0, (int*)nullptr, -1);
            HIPC(hipEventRecord(evP(kb), s3));
          }
          if (kb + 1 < nblk) {  // s3: column part of update kb
            if (kb > 0) HIPC(hipStreamWaitEvent(s3, evR(kb - 1), 0));
            if (symsweep)
              UPDK(true, dim3(nblk + 1), s3, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set], (const double*)rt[set],
                   (const double*)dv[d3], 3, kb + 1, cold[set ^ 1], dcopy[set ^ 1]);
            else
              UPDK(false, dim3(nblk + 1), s3, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set], (const double*)rt[set],
                   (const double*)dv[d3], 3, kb + 1, cold[set ^ 1], dcopy[set ^ 1]);
            HIPC(hipEventRecord(evC(kb), s3));
          }
          if (kb > 0) HIPC(hipStreamWaitEvent(s1, evP(kb), 0));  // s1: the rest (last step: everything)
          const int part = kb + 1 < nblk ? 4 : 0, colarg = kb + 1 < nblk ? kb + 1 : -2;
          if (symsweep)
            UPDK(true, dim3(ug), s1, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set], (const double*)rt[set],
                 (const double*)dv[d3], part, colarg, (double*)nullptr, (double*)nullptr);
          else
            UPDK(false, dim3(ug), s1, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set], (const double*)rt[set],
                 (const double*)dv[d3], part, colarg, (double*)nullptr, (double*)nullptr);
          if (kb + 1 < nblk) HIPC(hipEventRecord(evR(kb), s1));
        }
      } else
      for (int kb = 0; kb < nblk; ++kb) {
        const int set = kb & 1, k0 = kb * 128, k1 = k0 + 128;
        if (kb + 1 < nblk) {
          hipStream_t s2 = h->stream2;
          if (kb > 0) HIPC(hipStreamWaitEvent(s2, h->la_events[2 * kb], 0));  // update kb-1 has left cold[set], dcopy[set]
          {  // two 128^3 products, one accumulator stream per wavefront, over 64 blocks
            hipLaunchKernelGGL(k_gemm128<0>, dim3(64), dim3(256), 0, s2, (const double*)(cold[set] + (long)k1 * 128), (const double*)dinvT[set],
                               -1.0, (const double*)nullptr, 0L, rb);
            hipLaunchKernelGGL(k_gemm128<1>, dim3(64), dim3(256), 0, s2, (const double*)(cold[set] + (long)k1 * 128), (const double*)rb, 0.0,
                               (const double*)dcopy[set], 128L, dnext);
          }
          // the diagonal-inverse kernels address T[(k0 + r) * ld + k0 + c]: hand them the 128 x 128 copy under that indexing
          const double* dview = (const double*)((uintptr_t)dnext - sizeof(double) * ((size_t)k1 * 128 + (size_t)k1));
          launch_diag_inv(h, s2, dview, 128L, k1, nspd, dinv[set ^ 1], dinvT[set ^ 1], gate);
          HIPC(hipEventRecord(h->la_events[2 * kb + 1], s2));
        }
        const bool gate_here = gate && kb + 1 < nblk;
        if (kb > 0) {
          HIPC(hipStreamWaitEvent(h->stream, h->la_events[2 * kb - 1], 0));  // diagonal inverse kb
          // (the per-wavefront form of k_gemm128 for ALL panel rows was tried here: 30 us against 26 us -- its strided fragment
          // loads do not coalesce -- and its 640 blocks delay the chain's 64)
          // k_panel, leaving, polls for diagonal inverse kb+1 to have started (the gate)
          PANEL(h->stream, (const double*)cold[set], 128L, (const double*)dinvT[set], -1.0, cnew[set], rt[set], k0, fl, gate_here ? kb + 1 : -1);
        }
        if (kb + 1 < nblk) {
          if (gate_here && kb == 0) hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, h->stream, (const int*)fl, kb + 1, 20000);
          if (symsweep)
            UPDK(true, dim3(ug), h->stream, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set],
                 (const double*)rt[set], (const double*)dinv[set], 0, kb + 1, cold[set ^ 1], dcopy[set ^ 1]);
          else
            UPDK(false, dim3(ug), h->stream, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set],
                 (const double*)rt[set], (const double*)dinv[set], 0, kb + 1, cold[set ^ 1], dcopy[set ^ 1]);
          HIPC(hipEventRecord(h->la_events[2 * kb + 2], h->stream));
        } else {
          UPD(dim3(ug), h->stream, cold[set], cnew[set], rt[set], dinv[set], 0, -2, (double*)nullptr);
        }
      }
    }
  } else
  for (int kb = 0; kb < nblk; ++kb) {
    const int k0 = kb * 128;
    if (pivoted) {
      hipLaunchKernelGGL(k_copy_panel, dim3(pgrid), dim3(256), 0, h->stream, T, ld, k0, Mp, h->P0.as<double>());
      hipLaunchKernelGGL(k_piv_first, dim3(ncand), dim3(64), 0, h->stream, h->P0.as<double>(), k0, h->M, Mp,
                         h->cand0.as<PivCand>());
      for (int c = 0; c < 128; ++c) {
        const double* Pin = (c & 1) ? h->P1.as<double>() : h->P0.as<double>();
        double* Pout = (c & 1) ? h->P0.as<double>() : h->P1.as<double>();
        const PivCand* cin = (c & 1) ? h->cand1.as<PivCand>() : h->cand0.as<PivCand>();
        PivCand* cout = (c & 1) ? h->cand0.as<PivCand>() : h->cand1.as<PivCand>();
        hipLaunchKernelGGL(k_piv_step, dim3(ncand), dim3(256), 0, h->stream, Pin, Pout, k0, c, h->M, Mp, cin, cout, ncand,
                           h->pivall.as<int>() + k0, h->flag.as<int>());
      }
      hipLaunchKernelGGL(k_swap_rows, dim3((Mp + 255) / 256), dim3(256), 0, h->stream, T, ld, k0,
                         (const int*)(h->pivall.as<int>() + k0), Mp);
    }
    launch_diag_inv(h, h->stream, (const double*)T, ld, k0, nspd, h->Dinv.as<double>(), h->DinvT.as<double>());
    if (symsweep) hipLaunchKernelGGL(k_copy_panel_sym, dim3(Mp / 64), dim3(256), 0, h->stream, (const double*)T, ld, k0, Mp,
                                     h->Cold.as<double>());
    else hipLaunchKernelGGL(k_copy_panel, dim3(pgrid), dim3(256), 0, h->stream, (const double*)T, ld, k0, Mp,
                            h->Cold.as<double>());
    // unpivoted sweep: the panel kernel writes R^T = -sigma C_new as well (one launch less per step)
    PANEL(h->stream, (const double*)h->Cold.as<double>(), 128L, (const double*)h->DinvT.as<double>(), -1.0, h->Cnew.as<double>(),
          pivoted ? (double*)nullptr : h->Rt.as<double>(), k0, (int*)nullptr, -1);
    if (pivoted) {
      hipLaunchKernelGGL(k_transpose_rows, dim3(Mp / 64, 2), dim3(256), 0, h->stream, (const double*)T, ld, k0, Mp,
                         h->TKt.as<double>());
      PANEL(h->stream, (const double*)h->TKt.as<double>(), 128L, (const double*)h->Dinv.as<double>(), 1.0, h->Rt.as<double>(), (double*)nullptr, 0,
            (int*)nullptr, -1);
    }
    UPD(dim3(ug), h->stream, h->Cold.as<double>(), h->Cnew.as<double>(), h->Rt.as<double>(), h->Dinv.as<double>(), 0, 0, (double*)nullptr);
  }
#undef PANEL
#undef UPD
#undef UPDX
#undef UPDK
  if (symsweep) hipLaunchKernelGGL(k_mirror_upper, dim3(Mp / 64, Mp / 64), dim3(256), 0, h->stream, T, ld, Mp / 64);
  if (pivoted)
    hipLaunchKernelGGL(k_swap_cols, dim3((Mp + 255) / 256), dim3(256), 0, h->stream, T, ld,
                       (const int*)h->pivall.as<int>(), Mp, Mp);
  HIPC(hipGetLastError());
  int flag = 0;
  HIPC(hipMemcpyAsync(&flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  *flag_out = flag;
  return MIK_OK;
}

// Is the matrix X in T provably the Moore-Penrose inverse of the kriging matrix?  Probe vectors against the matrix itself
// (assembled again, unshifted, into a scratch buffer): A X A v = A v and X A X v = X v to 1e-8 -- the second condition is what
// tells the pseudo-inverse from the other generalised inverses pinv(A) + c P with A P = 0 -- and an estimated condition number
// far below SciPy's cut-off 1 / (M eps), i.e. no singular value the pseudo-inverse would have dropped.
static int verify_pinv(mik_handle* h, bool* done) {
  *done = false;
  const int M = h->M;
  const long ld = h->Mp;
  DevBuf vec;
  MIKC(h->Averify.ensure(sizeof(double) * (size_t)h->Mp * h->Mp));
  DevBuf& A2 = h->Averify;
  MIKC(launch_assemble(h, 0.0, A2.as<double>()));
  constexpr int NPROBE = 3;
  MIKC(vec.ensure(sizeof(double) * 4 * (size_t)h->Mp));
  double *dv = vec.as<double>(), *dy = dv + h->Mp, *dw = dy + h->Mp, *dr = dw + h->Mp;
  std::vector<double> hv(M), hy(M), hw(M), hr(M), hx(M);
  unsigned long long seed = 0x9E3779B97F4A7C15ull;
  double worst_res = 0.0, worst_res2 = 0.0, est_a = 0.0, est_x = 0.0;
  const unsigned mg = (unsigned)((M + 3) / 4);
  auto norm = [&](const std::vector<double>& a) {
    double s2 = 0.0;
    for (double x : a) s2 += x * x;
    return std::sqrt(s2);
  };
  for (int pr = 0; pr < NPROBE; ++pr) {
    for (int i = 0; i < M; ++i) {
      seed = seed * 6364136223846793005ull + 1442695040888963407ull;
      hv[i] = (double)((seed >> 11) & 0xFFFFFFFFull) / 4294967296.0 - 0.5;
    }
    HIPC(hipMemcpyAsync(dv, hv.data(), sizeof(double) * M, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(k_matvec, dim3(mg), dim3(256), 0, h->stream, (const double*)A2.as<double>(), ld, M, (const double*)dv, dy);  // y = A v
    hipLaunchKernelGGL(k_matvec, dim3(mg), dim3(256), 0, h->stream, (const double*)h->T.as<double>(), ld, M, (const double*)dy, dw);  // w = X y
    hipLaunchKernelGGL(k_matvec, dim3(mg), dim3(256), 0, h->stream, (const double*)A2.as<double>(), ld, M, (const double*)dw, dr);  // r = A w
    HIPC(hipMemcpyAsync(hy.data(), dy, sizeof(double) * M, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipMemcpyAsync(hr.data(), dr, sizeof(double) * M, hipMemcpyDeviceToHost, h->stream));
    hipLaunchKernelGGL(k_matvec, dim3(mg), dim3(256), 0, h->stream, (const double*)h->T.as<double>(), ld, M, (const double*)dv, dw);  // u = X v
    HIPC(hipMemcpyAsync(hw.data(), dw, sizeof(double) * M, hipMemcpyDeviceToHost, h->stream));
    // the second Penrose condition, X A X v = X v: it is what tells the Moore-Penrose inverse from the other generalised
    // inverses pinv(A) + c P (P = the duplicated stations' projector, A P = 0), which all pass A X A v = A v
    hipLaunchKernelGGL(k_matvec, dim3(mg), dim3(256), 0, h->stream, (const double*)A2.as<double>(), ld, M, (const double*)dw, dy);   // A u
    hipLaunchKernelGGL(k_matvec, dim3(mg), dim3(256), 0, h->stream, (const double*)h->T.as<double>(), ld, M, (const double*)dy, dr);  // X A u
    HIPC(hipMemcpyAsync(hx.data(), dr, sizeof(double) * M, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    {
      double e2 = 0.0;
      for (int i = 0; i < M; ++i) e2 += (hx[i] - hw[i]) * (hx[i] - hw[i]);
      const double nu = norm(hw);
      if (!(nu > 0.0) || !std::isfinite(nu)) return MIK_OK;
      worst_res2 = std::max(worst_res2, std::sqrt(e2) / nu);
    }
    const double nv = norm(hv), ny = norm(hy);
    double d2 = 0.0;
    for (int i = 0; i < M; ++i) d2 += (hr[i] - hy[i]) * (hr[i] - hy[i]);
    if (!(ny > 0.0) || !std::isfinite(ny)) return MIK_OK;
    worst_res = std::max(worst_res, std::sqrt(d2) / ny);
    est_a = std::max(est_a, ny / nv);
    est_x = std::max(est_x, norm(hw) / nv);
  }
  HIPC(hipGetLastError());
  const double eps = 2.220446049250313e-16;
  if (!(worst_res <= 1e-8) || !(worst_res2 <= 1e-8) || !(est_a * est_x <= 1e-3 / ((double)M * eps))) return MIK_OK;  // not provably the pseudo-inverse
  *done = true;
  return MIK_OK;
}


// Pseudo-inverse without an SVD for the case it exists for: DUPLICATED STATIONS (core.py:33, "redundant points are averaged").
// With a zero nugget two stations at the same place give two identical rows, i.e. the null vector e_i - e_j; for a symmetric A
// whose null space has the orthonormal basis N,  A + N N^T  is regular and  pinv(A) = (A + N N^T)^-1 - N N^T.  A group of m
// coincident stations contributes the projector I_m - 11^T / m on its index set.  So: find the groups on the host (exact
// coordinate equality -- what makes the reference's distances exactly zero), add the projectors, invert with the ordinary
// shifted sweep (the station block C + N N^T is positive definite again), subtract them.  Nothing is assumed: the result is
// VERIFIED with probe vectors -- A X A v = A v to 1e-8 and an estimated condition number far below SciPy's cut-off
// 1 / (M eps), i.e. no singular value the pseudo-inverse would have dropped -- and on any doubt (other rank deficiencies,
// near-singular matrices, a flagged pivot) *done stays false and the caller runs the Jacobi pseudo-inverse.
static int run_deflated_inverse(mik_handle* h, bool* done) {
  *done = false;
  if (h->model == MIK_MODEL_CUSTOM || !h->opt_pinv_fast) return MIK_OK;
  const int N = h->N, M = h->M;
  const long ld = h->Mp;
  const double nugget = (h->v.model == 0) ? h->v.p1 : h->v.p2;
  std::vector<int> ij;
  std::vector<double> val;
  if (nugget == 0.0) {
    std::vector<int> order(N);
    for (int i = 0; i < N; ++i) order[i] = i;
    const bool three = h->ndim == 3;
    auto less = [&](int a, int b) {
      if (h->hxs[a] != h->hxs[b]) return h->hxs[a] < h->hxs[b];
      if (h->hys[a] != h->hys[b]) return h->hys[a] < h->hys[b];
      if (three && h->hzs[a] != h->hzs[b]) return h->hzs[a] < h->hzs[b];
      return a < b;
    };
    auto same = [&](int a, int b) { return h->hxs[a] == h->hxs[b] && h->hys[a] == h->hys[b] && (!three || h->hzs[a] == h->hzs[b]); };
    std::sort(order.begin(), order.end(), less);
    for (int s0 = 0; s0 < N;) {
      int s1 = s0 + 1;
      while (s1 < N && same(order[s0], order[s1])) ++s1;
      const int m = s1 - s0;
      if (m > 1) {
        if ((long)val.size() + (long)m * m > 4000000L) return MIK_OK;  // absurdly many duplicates: leave it to the general path
        for (int a = s0; a < s1; ++a)
          for (int b = s0; b < s1; ++b) {
            ij.push_back(order[a]);
            ij.push_back(order[b]);
            val.push_back((a == b ? 1.0 : 0.0) - 1.0 / m);
          }
      }
      s0 = s1;
    }
  }
  const int ne = (int)val.size();
  DevBuf dij, dval;
  if (ne) {
    MIKC(dij.ensure(sizeof(int) * ij.size()));
    MIKC(dval.ensure(sizeof(double) * val.size()));
    HIPC(hipMemcpyAsync(dij.p, ij.data(), sizeof(int) * ij.size(), hipMemcpyHostToDevice, h->stream));
    HIPC(hipMemcpyAsync(dval.p, val.data(), sizeof(double) * val.size(), hipMemcpyHostToDevice, h->stream));
  }
  const double shift = h->shift_guess;
  MIKC(launch_assemble(h, shift));
  if (ne) hipLaunchKernelGGL(k_coo_add, dim3((ne + 255) / 256), dim3(256), 0, h->stream, h->T.as<double>(), ld, (const int*)dij.as<int>(),
                             (const double*)dval.as<double>(), ne, 1.0);
  int flag = 0;
  MIKC(run_block_inverse(h, false, N, &flag));
  if (flag) return MIK_OK;
  hipLaunchKernelGGL(k_add_diag, dim3(1), dim3(1), 0, h->stream, h->T.as<double>(), ld, M - 1, shift);
  if (ne) hipLaunchKernelGGL(k_coo_add, dim3((ne + 255) / 256), dim3(256), 0, h->stream, h->T.as<double>(), ld, (const int*)dij.as<int>(),
                             (const double*)dval.as<double>(), ne, -1.0);
  return verify_pinv(h, done);
}

// Probe of the inverse X in T against the matrix itself (assembled again, unshifted, into a scratch buffer):
//   res_z   = max |A c - [Z; 0]| / max(1, max|Z|)   with c = X[:, :N] Z: every z_g = c.b_g is w_g.(A c) with the kriging weights
//             w_g of the point (sum 1, |w|_1 of order 1..10), so the error of z is bounded by |w_g|_1 res_z max|Z|;
//   res_inv = max_j max |X A e_j - e_j|  for three station columns j (first, middle, last): A e_j is the right-hand side of a
//             point ON station j, X A e_j its weight vector -- what sigma^2 is formed from.
// Cost: one assembly, one product with A, one pass over X (0.25 ms at N = 5000).  cvec must be current.
static int verify_inverse(mik_handle* h, double* res_z, double* res_inv) {
  const int M = h->M, N = h->N, Mp = h->Mp;
  const long ld = Mp;
  MIKC(h->Averify.ensure(sizeof(double) * (size_t)Mp * Mp));
  MIKC(h->vbuf.ensure(sizeof(double) * 4 * (size_t)Mp));
  MIKC(launch_assemble(h, 0.0, h->Averify.as<double>(), h->factor_sorted, h->factor_eq));
  const std::vector<double>& hv = h->factor_sorted ? h->hvals_s : h->hvals;
  const double* A2 = h->Averify.as<double>();
  double* y = h->vbuf.as<double>();
  const unsigned mg = (unsigned)((M + 3) / 4);
  const int cols[3] = {0, N / 2, N - 1};
  hipLaunchKernelGGL(k_matvec, dim3(mg), dim3(256), 0, h->stream, A2, ld, M, (const double*)h->cvec.as<double>(), y);
  hipLaunchKernelGGL(k_matvec3, dim3(mg), dim3(256), 0, h->stream, (const double*)h->T.as<double>(), ld, M, A2 + (long)cols[0] * ld,
                     A2 + (long)cols[1] * ld, A2 + (long)cols[2] * ld, y + Mp, y + 2 * Mp, y + 3 * Mp);
  HIPC(hipGetLastError());
  std::vector<double> host(4 * (size_t)Mp);
  HIPC(hipMemcpyAsync(host.data(), y, sizeof(double) * host.size(), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  double zmax = 1.0, rz = 0.0, ri = 0.0;
  for (int i = 0; i < N; ++i) zmax = std::max(zmax, std::fabs(hv[i]));
  for (int i = 0; i < M; ++i) {
    const double d = std::fabs(host[i] - (i < N ? hv[i] : 0.0));
    rz = std::max(rz, std::isfinite(d) ? d : 1e300);
    for (int k = 0; k < 3; ++k) {
      const double e = std::fabs(host[(size_t)(k + 1) * Mp + i] - (i == cols[k] ? 1.0 : 0.0));
      ri = std::max(ri, std::isfinite(e) ? e : 1e300);
    }
  }
  *res_z = rz / zmax;
  *res_inv = ri;
  return MIK_OK;
}

// Pseudo-inverse of a symmetric matrix with a SMALL null space of unknown origin (round 3; e.g. collinear stations under a
// regional-linear drift: two drift columns become dependent) without a decomposition of the whole matrix:
//   1. sigma = 1e-10 |A| (far below any eigenvalue a kriging matrix of cond <= 1e8 has, far above the rounding of the zero ones):
//      (A - sigma I)^-1 by the pivoted block inverse turns the eigenvalues lambda into 1 / (lambda - sigma), so the null space stands
//      out by a factor |lambda_min| / sigma; three rounds of subspace iteration with b = 24 random vectors;
//   2. Rayleigh-Ritz of A on that subspace (a b x b symmetric eigenproblem, host Jacobi): Ritz pairs with |theta| <= 1e-11 |A| and a
//      small residual are null vectors N (b of them = the null space may be larger than the subspace: give up);
//   3. pinv(A) = (A + |A| N N^T)^-1 - N N^T / |A| (the identity of the duplicated-stations path), pivoted block inverse;
//   4. the result is checked where it is most sensitive -- A X u = u for the OTHER Ritz vectors u, the directions of A's smallest
//      non-zero eigenvalues, to 1e-7 -- and then by verify_pinv (both Penrose conditions on random probes, condition estimate).
//      On any doubt *done stays false and the caller runs the one-sided Jacobi pseudo-inverse (9.3 s at M = 4000 against ~0.2 s).
static int run_nullspace_inverse(mik_handle* h, bool* done) {
  *done = false;
  if (h->model == MIK_MODEL_CUSTOM || !h->opt_pinv_fast) return MIK_OK;
  const int M = h->M, Mp = h->Mp;
  const long ld = Mp;
  constexpr int B = 24;
  const double eps = 2.220446049250313e-16;
  const unsigned mg = (unsigned)((M + 3) / 4);
  MIKC(h->Averify.ensure(sizeof(double) * (size_t)Mp * Mp));
  MIKC(launch_assemble(h, 0.0, h->Averify.as<double>()));
  const double* A2 = h->Averify.as<double>();
  DevBuf dq, dw;
  MIKC(dq.ensure(sizeof(double) * (size_t)B * Mp));
  MIKC(dw.ensure(sizeof(double) * (size_t)B * Mp));
  double* Q = dq.as<double>();
  double* W = dw.as<double>();
  std::vector<double> hq((size_t)B * M), hw((size_t)B * M);
  unsigned long long seed = 0x243F6A8885A308D3ull;
  auto rnd = [&]() {
    seed = seed * 6364136223846793005ull + 1442695040888963407ull;
    return (double)((seed >> 11) & 0xFFFFFFFFull) / 4294967296.0 - 0.5;
  };
  auto upload = [&](const std::vector<double>& v, double* dst) -> int {
    HIPC(hipMemcpy2DAsync(dst, sizeof(double) * Mp, v.data(), sizeof(double) * M, sizeof(double) * M, B, hipMemcpyHostToDevice, h->stream));
    return MIK_OK;
  };
  auto download = [&](std::vector<double>& v, const double* src) -> int {
    HIPC(hipMemcpy2DAsync(v.data(), sizeof(double) * M, src, sizeof(double) * Mp, sizeof(double) * M, B, hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    return MIK_OK;
  };
  auto apply = [&](const double* Mat, const double* src, double* dst) {  // dst_k = Mat src_k, k < B (rows of length Mp)
    for (int k = 0; k < B; ++k)
      hipLaunchKernelGGL(k_matvec, dim3(mg), dim3(256), 0, h->stream, Mat, ld, M, src + (size_t)k * Mp, dst + (size_t)k * Mp);
  };
  auto mgs = [&](std::vector<double>& v) {  // modified Gram-Schmidt (twice) on the B rows of v; false if a row vanishes
    for (int pass = 0; pass < 2; ++pass)
      for (int a = 0; a < B; ++a) {
        double* va = v.data() + (size_t)a * M;
        for (int b = 0; b < a; ++b) {
          const double* vb = v.data() + (size_t)b * M;
          double d = 0.0;
          for (int i = 0; i < M; ++i) d += va[i] * vb[i];
          for (int i = 0; i < M; ++i) va[i] -= d * vb[i];
        }
        double n2 = 0.0;
        for (int i = 0; i < M; ++i) n2 += va[i] * va[i];
        if (!(n2 > 1e-300) || !std::isfinite(n2)) return false;
        const double inv = 1.0 / std::sqrt(n2);
        for (int i = 0; i < M; ++i) va[i] *= inv;
      }
    return true;
  };
  // |A| by a few power iterations
  double anorm = 0.0;
  {
    for (int i = 0; i < M; ++i) hq[i] = rnd();
    for (int it = 0; it < 6; ++it) {
      double n2 = 0.0;
      for (int i = 0; i < M; ++i) n2 += hq[i] * hq[i];
      const double inv = 1.0 / std::sqrt(n2);
      for (int i = 0; i < M; ++i) hq[i] *= inv;
      HIPC(hipMemcpyAsync(Q, hq.data(), sizeof(double) * M, hipMemcpyHostToDevice, h->stream));
      hipLaunchKernelGGL(k_matvec, dim3(mg), dim3(256), 0, h->stream, A2, ld, M, (const double*)Q, W);
      HIPC(hipMemcpyAsync(hq.data(), W, sizeof(double) * M, hipMemcpyDeviceToHost, h->stream));
      HIPC(hipStreamSynchronize(h->stream));
      n2 = 0.0;
      for (int i = 0; i < M; ++i) n2 += hq[i] * hq[i];
      anorm = std::sqrt(n2);
    }
  }
  if (!(anorm > 0.0) || !std::isfinite(anorm)) return MIK_OK;
  // "zero" eigenvalue: SciPy's pinv drops singular values below M eps |A| (1e-13 .. 1e-12 |A|).  The null vectors come out of a
  // shift-and-invert iteration whose accuracy is eps |A| / lambda_min, so the classification here is |theta| <= 1e-11 |A|
  // with a residual |A y| <= 1e-9 |A|; an eigenvalue between the two cut-offs would make SciPy's own result rounding noise
  // (1 / lambda >= 1e11), and the checks below send anything that ill-conditioned to the Jacobi path anyway.
  const double tol_null = std::max(1e3 * (double)M * eps, 1e-11) * anorm, tol_res = 1e-9 * anorm;
  // 1. (A - sigma I)^-1
  const double sigma = 1e-10 * anorm;
  MIKC(launch_assemble(h, 0.0));
  hipLaunchKernelGGL(k_shift_diag, dim3((M + 255) / 256), dim3(256), 0, h->stream, h->T.as<double>(), ld, M, -sigma);
  int flag = 0;
  MIKC(run_block_inverse(h, true, 0, &flag));
  if (flag) return MIK_OK;
  for (size_t i = 0; i < hq.size(); ++i) hq[i] = rnd();
  if (!mgs(hq)) return MIK_OK;
  for (int round = 0; round < 3; ++round) {
    MIKC(upload(hq, Q));
    apply(h->T.as<double>(), Q, W);
    MIKC(download(hq, W));
    if (!mgs(hq)) return MIK_OK;
  }
  // 2. Rayleigh-Ritz of A on span(Q)
  MIKC(upload(hq, Q));
  apply(A2, Q, W);
  HIPC(hipGetLastError());
  MIKC(download(hw, W));  // rows: A q_k
  double H[B][B], S[B][B];
  for (int a = 0; a < B; ++a)
    for (int b = 0; b < B; ++b) {
      double d = 0.0;
      for (int i = 0; i < M; ++i) d += hq[(size_t)a * M + i] * hw[(size_t)b * M + i];
      H[a][b] = d;
      S[a][b] = a == b ? 1.0 : 0.0;
    }
  for (int a = 0; a < B; ++a)
    for (int b = 0; b < a; ++b) H[a][b] = H[b][a] = 0.5 * (H[a][b] + H[b][a]);
  for (int sweep = 0; sweep < 60; ++sweep) {  // cyclic Jacobi on the B x B matrix
    double off = 0.0;
    for (int a = 0; a < B; ++a)
      for (int b = a + 1; b < B; ++b) off += H[a][b] * H[a][b];
    if (off <= 1e-60) break;
    for (int p = 0; p < B; ++p)
      for (int q = p + 1; q < B; ++q) {
        if (H[p][q] == 0.0) continue;
        const double th = (H[q][q] - H[p][p]) / (2.0 * H[p][q]);
        const double t = (th >= 0.0 ? 1.0 : -1.0) / (std::fabs(th) + std::sqrt(th * th + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < B; ++k) {
          const double hkp = H[k][p], hkq = H[k][q];
          H[k][p] = c * hkp - sn * hkq;
          H[k][q] = sn * hkp + c * hkq;
        }
        for (int k = 0; k < B; ++k) {
          const double hpk = H[p][k], hqk = H[q][k];
          H[p][k] = c * hpk - sn * hqk;
          H[q][k] = sn * hpk + c * hqk;
        }
        for (int k = 0; k < B; ++k) {
          const double skp = S[k][p], skq = S[k][q];
          S[k][p] = c * skp - sn * skq;
          S[k][q] = sn * skp + c * skq;
        }
      }
  }
  std::vector<double> hn;  // null vectors, rows of length M
  std::vector<double> hu;  // the other Ritz vectors (directions of the smallest non-zero eigenvalues of A), for the check of step 4
  int r = 0;
  const bool dbg = getenv("MIK_DEBUG_PINV") != nullptr;
  if (dbg) fprintf(stderr, "[pinv] M %d |A| %.3e sigma %.3e tol_null %.3e\n", M, anorm, sigma, tol_null);
  for (int e = 0; e < B; ++e) {
    const double theta = H[e][e];
    if (dbg) fprintf(stderr, "[pinv] ritz %d theta %.3e\n", e, theta);
    std::vector<double> y(M, 0.0), ay(M, 0.0);
    for (int k = 0; k < B; ++k) {
      const double sk = S[k][e];
      const double* qk = hq.data() + (size_t)k * M;
      const double* aq = hw.data() + (size_t)k * M;
      for (int i = 0; i < M; ++i) y[i] += sk * qk[i], ay[i] += sk * aq[i];
    }
    if (!(std::fabs(theta) <= tol_null)) {
      hu.insert(hu.end(), y.begin(), y.end());
      continue;
    }
    double res2 = 0.0;
    for (int i = 0; i < M; ++i) res2 += ay[i] * ay[i];
    if (dbg) fprintf(stderr, "[pinv]   residual %.3e\n", std::sqrt(res2));
    if (!(std::sqrt(res2) <= tol_res)) return MIK_OK;  // a tiny Ritz value that is not an eigenpair of A to that accuracy: no proof
    hn.insert(hn.end(), y.begin(), y.end());
    ++r;
  }
  if (r >= B) return MIK_OK;  // the null space may be larger than the subspace
  // 3. (A + N N^T)^-1 - N N^T
  DevBuf dn;
  if (r > 0) {
    // re-orthonormalise the null vectors among themselves
    for (int pass = 0; pass < 2; ++pass)
      for (int a = 0; a < r; ++a) {
        double* va = hn.data() + (size_t)a * M;
        for (int b = 0; b < a; ++b) {
          const double* vb = hn.data() + (size_t)b * M;
          double d = 0.0;
          for (int i = 0; i < M; ++i) d += va[i] * vb[i];
          for (int i = 0; i < M; ++i) va[i] -= d * vb[i];
        }
        double n2 = 0.0;
        for (int i = 0; i < M; ++i) n2 += va[i] * va[i];
        if (!(n2 > 0.25)) return MIK_OK;
        const double inv = 1.0 / std::sqrt(n2);
        for (int i = 0; i < M; ++i) va[i] *= inv;
      }
    MIKC(dn.ensure(sizeof(double) * (size_t)r * M));
    HIPC(hipMemcpyAsync(dn.p, hn.data(), sizeof(double) * (size_t)r * M, hipMemcpyHostToDevice, h->stream));
  }
  MIKC(launch_assemble(h, 0.0));
  const dim3 lg((M + 63) / 64, (M + 63) / 64);
  // the projector is scaled to the matrix (|A| N N^T): the deflated matrix keeps the conditioning of A's range
  const double scale = anorm;
  if (r > 0) hipLaunchKernelGGL(k_lowrank_add, lg, dim3(256), 0, h->stream, h->T.as<double>(), ld, M, (const double*)dn.as<double>(), (long)M, r, scale);
  MIKC(run_block_inverse(h, true, 0, &flag));
  if (flag) return MIK_OK;
  if (r > 0) hipLaunchKernelGGL(k_lowrank_add, lg, dim3(256), 0, h->stream, h->T.as<double>(), ld, M, (const double*)dn.as<double>(), (long)M, r, -1.0 / scale);
  HIPC(hipGetLastError());
  h->tm.null_dim = r;
  {  // A X u = u on the non-null Ritz vectors
    const int nu = B - r;
    hq.assign((size_t)B * M, 0.0);
    std::copy(hu.begin(), hu.end(), hq.begin());
    MIKC(upload(hq, Q));
    apply(h->T.as<double>(), Q, W);   // X u
    apply(A2, W, Q);                  // A X u
    HIPC(hipGetLastError());
    MIKC(download(hw, Q));
    double worst = 0.0;
    for (int k = 0; k < nu; ++k) {
      double d2 = 0.0, n2 = 0.0;
      for (int i = 0; i < M; ++i) {
        const double u = hu[(size_t)k * M + i], d = hw[(size_t)k * M + i] - u;
        d2 += d * d;
        n2 += u * u;
      }
      worst = std::max(worst, std::sqrt(d2 / std::max(n2, 1e-300)));
    }
    if (dbg) fprintf(stderr, "[pinv] null_dim %d, worst |A X u - u| / |u| over %d Ritz vectors: %.3e\n", r, nu, worst);
    if (!(worst <= 1e-7)) return MIK_OK;
  }
  MIKC(verify_pinv(h, done));
  return MIK_OK;
}

static int launch_cvec(mik_handle* h) {
  hipLaunchKernelGGL(k_cvec, dim3((h->Mp + 3) / 4), dim3(256), 0, h->stream, (const double*)h->T.as<double>(),
                     (long)h->Mp, h->M, h->N, (const double*)(h->factor_sorted ? h->vals_s.as<double>() : h->vals.as<double>()),
                     h->cvec.as<double>(), h->Mp);
  HIPC(hipGetLastError());
  return MIK_OK;
}

static int finish_factor(mik_handle* h) {
  // the pseudo-inverse paths (4: Jacobi, 5 / 6: deflated sweeps, verified by the Penrose conditions) end here with a matrix that
  // is symmetric up to rounding: average the triangles as after a full sweep (the caller's own inverse, path 3, is left alone)
  if (h->opt_symmetrize && h->tm.factor_path >= 4 && h->tm.factor_path <= 6)
    hipLaunchKernelGGL(k_symmetrize, dim3(h->Mp / 64, h->Mp / 64), dim3(256), 0, h->stream, h->T.as<double>(), (long)h->Mp, h->Mp / 64);
  MIKC(launch_cvec(h));
  HIPC(hipStreamSynchronize(h->stream));
  h->have_factor = true;
  h->t_state = 2;
  h->have_results = false;
  return MIK_OK;
}

int one_factor(mik_handle* h) {
  if (!h || !h->have_problem) return fail(MIK_ESTATE, "mik_factor: no problem set");
  HIPC(hipSetDevice(h->device));
  h->t_state = 0;
  h->have_factor = false;
  h->xpack_valid = false;
  h->factor_sorted = want_sorted(h);
  h->factor_eq = h->drift_eq && h->opt_drift_eq;
  MIKC(ensure_factor_buffers(h));
  MIKC(get_events(h, 4));
  h->tm.assemble_ms = h->tm.invert_ms = 0.0;
  if (h->host_inv) {
    HIPC(hipMemsetAsync(h->T.p, 0, h->T.bytes, h->stream));
    HIPC(hipMemcpy2DAsync(h->T.p, sizeof(double) * h->Mp, h->host_ainv.data(), sizeof(double) * h->M,
                          sizeof(double) * h->M, h->M, hipMemcpyHostToDevice, h->stream));
    h->tm.factor_path = 3;
    return finish_factor(h);
  }
  if (h->pinv) {
    {
      HIPC(hipEventRecord(h->evpool[0], h->stream));
      bool done = false;
      MIKC(run_deflated_inverse(h, &done));
      if (done) {
        HIPC(hipEventRecord(h->evpool[2], h->stream));
        HIPC(hipStreamSynchronize(h->stream));
        float ms0 = 0.f;
        HIPC(hipEventElapsedTime(&ms0, h->evpool[0], h->evpool[2]));
        h->tm.invert_ms = ms0;
        h->tm.factor_path = 5;
        return finish_factor(h);
      }
      // any other small null space: found numerically, deflated, verified (factor_path 6)
      HIPC(hipEventRecord(h->evpool[0], h->stream));
      MIKC(run_nullspace_inverse(h, &done));
      if (done) {
        HIPC(hipEventRecord(h->evpool[2], h->stream));
        HIPC(hipStreamSynchronize(h->stream));
        float ms0 = 0.f;
        HIPC(hipEventElapsedTime(&ms0, h->evpool[0], h->evpool[2]));
        h->tm.invert_ms = ms0;
        h->tm.factor_path = 6;
        return finish_factor(h);
      }
    }
    HIPC(hipEventRecord(h->evpool[0], h->stream));
    MIKC(launch_assemble(h, 0.0));
    HIPC(hipEventRecord(h->evpool[1], h->stream));
    MIKC(run_pseudo_inverse(h));
    HIPC(hipEventRecord(h->evpool[2], h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, h->evpool[0], h->evpool[1]));
    h->tm.assemble_ms = ms;
    HIPC(hipEventElapsedTime(&ms, h->evpool[1], h->evpool[2]));
    h->tm.invert_ms = ms;
    h->tm.factor_path = 4;
    return finish_factor(h);
  }
  // auto: every model first tries the unpivoted sweep on the shifted matrix s.11^T - Gamma (s = sill for the
  // bounded models, gamma(bounding-box diagonal) for linear/power); a non-positive station pivot (matrix not
  // positive definite, e.g. hole-effect in 2-D) sends the attempt to the pivoted path below.
  bool try_sweep = h->opt_factor == 1 || h->opt_factor == 0;
  if (h->model == MIK_MODEL_CUSTOM) try_sweep = false;  // no sill to shift by: pivoted elimination
  h->no_half_sweep = false;
  h->tm.factor_attempts = 0;
  h->tm.verify_ms = h->tm.verify_res_z = h->tm.verify_res_inv = 0.0;
  MIKC(get_events(h, 6));
  for (int attempt = 0; attempt < 3; ++attempt) {
    const bool pivoted = !try_sweep;
    const double shift = pivoted ? 0.0 : h->shift_guess;
    ++h->tm.factor_attempts;
    HIPC(hipEventRecord(h->evpool[0], h->stream));
    MIKC(launch_assemble(h, shift, nullptr, h->factor_sorted, h->factor_eq));
    HIPC(hipEventRecord(h->evpool[1], h->stream));
    int flag = 0;
    MIKC(run_block_inverse(h, pivoted, pivoted ? 0 : h->N, &flag));
    if (!pivoted) hipLaunchKernelGGL(k_add_diag, dim3(1), dim3(1), 0, h->stream, h->T.as<double>(), (long)h->Mp, h->M - 1, shift);
    // the half sweep leaves an exactly symmetric matrix (mirrored); every other elimination one that is symmetric up to
    // rounding: average the triangles (k_symmetrize) -- the symmetric contraction reads one of them
    if (!h->last_half_sweep && h->opt_symmetrize)
      hipLaunchKernelGGL(k_symmetrize, dim3(h->Mp / 64, h->Mp / 64), dim3(256), 0, h->stream, h->T.as<double>(), (long)h->Mp, h->Mp / 64);
    HIPC(hipEventRecord(h->evpool[2], h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    float ms = 0.f;
    HIPC(hipEventElapsedTime(&ms, h->evpool[0], h->evpool[1]));
    h->tm.assemble_ms += ms;
    HIPC(hipEventElapsedTime(&ms, h->evpool[1], h->evpool[2]));
    h->tm.invert_ms += ms;
    h->tm.factor_path = pivoted ? 2 : 1;
    h->tm.half_sweep = h->last_half_sweep ? 1 : 0;
    if (flag != 0) {
      if (!pivoted && h->opt_factor == 0) {  // shifted matrix not positive definite: redo with pivoting
        try_sweep = false;
        continue;
      }
      return fail(MIK_ESINGULAR, pivoted ? "singular matrix" : "singular matrix (unpivoted sweep hit a bad pivot; use factor=auto or pivoted)");
    }
    if (!h->opt_verify || h->model == MIK_MODEL_CUSTOM) return finish_factor(h);
    // the probe (verify_inverse): a half sweep the library chose by itself that fails it is redone as a full sweep, a full
    // sweep of factor = auto that fails it by partial pivoting; what the caller forced is only reported
    HIPC(hipEventRecord(h->evpool[4], h->stream));
    MIKC(launch_cvec(h));
    double rz = 0.0, ri = 0.0;
    MIKC(verify_inverse(h, &rz, &ri));
    HIPC(hipEventRecord(h->evpool[5], h->stream));
    HIPC(hipStreamSynchronize(h->stream));
    HIPC(hipEventElapsedTime(&ms, h->evpool[4], h->evpool[5]));
    h->tm.verify_ms += ms;
    h->tm.verify_res_z = rz;
    h->tm.verify_res_inv = ri;
    const bool good = rz <= h->verify_tol_z && ri <= h->verify_tol_inv;
    if (good || pivoted) return finish_factor(h);
    if (h->last_half_sweep && h->opt_symsweep < 0) {
      h->no_half_sweep = true;
      continue;
    }
    if (h->opt_factor == 0) {
      try_sweep = false;
      continue;
    }
    return finish_factor(h);
  }
  return fail(MIK_ESINGULAR, "singular matrix");
}
