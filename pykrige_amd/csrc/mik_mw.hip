// mik_mw.hip -- moving-window kriging: neighbour search, solver dispatch, prediction
// One translation unit of libmikrige.so (pykrige_amd/build.py compiles them in parallel).
#include "mik_k_mw.h"
#include "mik_host.h"

// Sort the stations into a uniform grid of cells for the moving-window neighbour search (counting sort on the host, O(N)).
// The cell edge aims at `target` stations per cell; geographic problems are binned by their unit-sphere vectors.
static int build_mw_grid(mik_handle* h, int target) {
  if (h->grid.target == target) return MIK_OK;
  const int N = h->N, D = (h->geo || h->ndim == 3) ? 3 : 2;
  std::vector<double> c[3];
  if (h->geo) {
    for (int d = 0; d < 3; ++d) c[d].resize(N);
    for (int i = 0; i < N; ++i) {  // k_geo_unit's formula
      const double lo = h->hxs[i] * MIK_PI / 180.0, la = h->hys[i] * MIK_PI / 180.0;
      c[0][i] = cos(lo) * cos(la);
      c[1][i] = sin(lo) * cos(la);
      c[2][i] = sin(la);
    }
  } else {
    c[0] = h->hxs;
    c[1] = h->hys;
    if (D == 3) c[2] = h->hzs;
  }
  double lo[3] = {0, 0, 0}, ext[3] = {0, 0, 0};
  double vol = 1.0;
  int live = 0;
  for (int d = 0; d < D; ++d) {
    const auto mm = std::minmax_element(c[d].begin(), c[d].end());
    lo[d] = *mm.first;
    ext[d] = *mm.second - *mm.first;
    if (ext[d] > 0.0 && std::isfinite(ext[d])) {
      vol *= ext[d];
      ++live;
    }
  }
  int n[3] = {1, 1, 1};
  double cell = 1.0;
  if (live > 0 && N > 4 * target) {
    cell = pow(vol * (double)target / (double)N, 1.0 / live);
    for (;;) {  // keep the grid below ~4M cells
      double cells = 1.0;
      for (int d = 0; d < D; ++d) cells *= (ext[d] > 0.0 && std::isfinite(ext[d])) ? std::max(1.0, ceil(ext[d] / cell)) : 1.0;
      if (cells <= 4.0e6) break;
      cell *= 1.5;
    }
    for (int d = 0; d < D; ++d)
      if (ext[d] > 0.0 && std::isfinite(ext[d])) n[d] = (int)std::max(1.0, ceil(ext[d] / cell));
  }
  const long ncell = (long)n[0] * n[1] * n[2];
  std::vector<int> cellof(N), start(ncell + 1, 0), orig(N);
  for (int i = 0; i < N; ++i) {
    long id[3] = {0, 0, 0};
    for (int d = 0; d < D; ++d)
      if (n[d] > 1) id[d] = std::min<long>(n[d] - 1, std::max<long>(0, (long)floor((c[d][i] - lo[d]) / cell)));
    const long ci = (id[2] * n[1] + id[1]) * n[0] + id[0];
    cellof[i] = (int)ci;
    ++start[ci + 1];
  }
  for (long k = 0; k < ncell; ++k) start[k + 1] += start[k];
  std::vector<int> fill(start.begin(), start.end() - 1);
  std::vector<double> g[3];
  for (int d = 0; d < D; ++d) g[d].resize(N);
  for (int i = 0; i < N; ++i) {  // stable: stations of a cell stay in index order
    const int pos = fill[cellof[i]]++;
    orig[pos] = i;
    for (int d = 0; d < D; ++d) g[d][pos] = c[d][i];
  }
  auto& G = h->grid;
  MIKC(G.gx.ensure(sizeof(double) * N));
  MIKC(G.gy.ensure(sizeof(double) * N));
  MIKC(G.gz.ensure(sizeof(double) * N));
  MIKC(G.orig.ensure(sizeof(int) * N));
  MIKC(G.cstart.ensure(sizeof(int) * (size_t)(ncell + 1)));
  HIPC(hipMemcpyAsync(G.gx.p, g[0].data(), sizeof(double) * N, hipMemcpyHostToDevice, h->stream));
  HIPC(hipMemcpyAsync(G.gy.p, g[1].data(), sizeof(double) * N, hipMemcpyHostToDevice, h->stream));
  if (D == 3) HIPC(hipMemcpyAsync(G.gz.p, g[2].data(), sizeof(double) * N, hipMemcpyHostToDevice, h->stream));
  HIPC(hipMemcpyAsync(G.orig.p, orig.data(), sizeof(int) * N, hipMemcpyHostToDevice, h->stream));
  HIPC(hipMemcpyAsync(G.cstart.p, start.data(), sizeof(int) * (size_t)(ncell + 1), hipMemcpyHostToDevice, h->stream));
  HIPC(hipStreamSynchronize(h->stream));  // the host vectors go out of scope
  G.nx = n[0], G.ny = n[1], G.nz = n[2];
  G.x0 = lo[0], G.y0 = lo[1], G.z0 = lo[2];
  G.cell = cell;
  G.target = target;
  G.live = live;
  G.per_cell = (double)N / ((double)n[0] * n[1] * n[2]);
  return MIK_OK;
}

// (dispatch_mw_solve: mik_mw_solve.hip)
// the classes of k_mw_chol live in mik_mw_chol.hip (four translation units): class = 100 G + RI
static int launch_mw_chol_class(mik_handle* h, const MwArgs& a, long pc, int cls) {
  for (int part = 0; part < MIK_MWC_PARTS; ++part) {
    const int rc = mw_chol_part(part, cls, h->stream, h->opt_mw_static != 0, a, pc);
    if (rc != MIK_MWC_NOCLASS) return rc;
  }
  return fail(MIK_EINVAL, "mw_class: no such LDL^T class");
}
template <int G, int RI>
static int launch_mw_chol(mik_handle* h, const MwArgs& a, long pc) { return launch_mw_chol_class(h, a, pc, 100 * G + RI); }

// thread-grid / register-tile classes of k_mw_chol: {G, RI} covers K <= G * RI
#define MIK_MW_CHOL_KMAX 256
static int dispatch_mw_chol(mik_handle* h, const MwArgs& a, long pc) {
  const int K = a.K;
  if (h->opt_mw_class) {  // "mw_class" = 100 G + RI: a class forced for A/B runs (scripts/mw_classes.py)
    // (round 4: the classes that lost every A/B of rounds 2-3 -- {8,14}, {8,16}, {16,4..6}, {16,15}, {16,16}, {32,5..7} -- are no longer
    // built: each was a 100 000-instruction kernel; profiles/r03_mw_classes_*.txt keep their measurements)
    return launch_mw_chol_class(h, a, pc, h->opt_mw_class);
  }
  // measured per window size (scripts/mw_classes.py, profiles/r03_mw_classes_after_kernel_changes.txt): one wavefront per point
  // as long as the register tile stays at RI <= 13 (RI = 13 only with the lean update below; beyond that the kernel needs more
  // than 256 registers and the occupancy halves: 2 x slower), then 256 threads per point up to RI = 12, 1024 threads for the last two
  if (K <= 16) return launch_mw_chol<4, 4>(h, a, pc);    // 16 threads per point, 4 points per wavefront
  // round 4 (profiles/r04_mw_classes_g4.txt): 16 threads per point keep winning while the tile fits -- k = 24: {4,6} 0.37 ms per
  // 2e5 points against {8,4} 0.76; k = 32: {4,8} 0.70 / 0.93; k = 40: {4,10} 1.39 / {8,6} 1.50; k = 50: {4,13} 2.15 / {8,8} 2.54
  if (K <= 24) return launch_mw_chol<4, 6>(h, a, pc);
  if (K <= 32) return launch_mw_chol<4, 8>(h, a, pc);
  if (K <= 40) return launch_mw_chol<4, 10>(h, a, pc);
  if (K <= 48) return launch_mw_chol<8, 6>(h, a, pc);    // one wavefront per point from here to K = 104: no workgroup barrier
  if (K <= 52) return launch_mw_chol<4, 13>(h, a, pc);
  if (K <= 64) return launch_mw_chol<8, 8>(h, a, pc);
  if (K <= 80) return launch_mw_chol<8, 10>(h, a, pc);
  if (K <= 88) return launch_mw_chol<8, 11>(h, a, pc);
  if (K <= 96) return launch_mw_chol<8, 12>(h, a, pc);
  // RI = 13 in one wavefront (round 3, second session): held to 2 wavefronts per SIMD by its launch bound, row factors read as
  // they are used (MIK_MWC_LEAN): 8 spilled registers instead of 24 AGPRs and half the occupancy -- k = 100: 10.9 ms per 2e5
  // points against 13.5 for {16,7}.  {8,14} ties with {16,7} at k = 112 (14.5 / 14.2 ms): not used.
  if (K <= 104) return launch_mw_chol<8, 13>(h, a, pc);
  if (K <= 112) return launch_mw_chol<16, 7>(h, a, pc);  // 256 threads per point
  if (K <= 128) return launch_mw_chol<16, 8>(h, a, pc);
  if (K <= 144) return launch_mw_chol<16, 9>(h, a, pc);
  if (K <= 160) return launch_mw_chol<16, 10>(h, a, pc);
  if (K <= 176) return launch_mw_chol<16, 11>(h, a, pc);
  if (K <= 192) return launch_mw_chol<16, 12>(h, a, pc);
  // second session of round 3: RI = 13 / 14 on 256 threads, held to 2 wavefronts per SIMD (launch bound + lean update): k = 200
  // 113 -> 65 ms per 2e5 points, k = 224 123 -> 88 ms -- they replace the 1024-thread class {32,7}
  if (K <= 208) return launch_mw_chol<16, 13>(h, a, pc);
  if (K <= 224) return launch_mw_chol<16, 14>(h, a, pc);
  return launch_mw_chol<32, 8>(h, a, pc);                // K <= 256: 1024 threads per point
}

int one_predict_mw(mik_handle* h, int n_closest) {
  if (!h || !h->have_problem) return fail(MIK_ESTATE, "mik_predict_moving_window: set the problem first");
  if (!h->have_points) return fail(MIK_ESTATE, "mik_predict_moving_window: set points first");
  if (h->p != 0) return fail(MIK_EINVAL, "moving-window kriging exists for ordinary kriging only (ok.py:929, ok3d.py:901)");
  if (n_closest < 2) return fail(MIK_EINVAL, "n_closest_points has to be at least two!");
  if (n_closest > h->N) return fail(MIK_EINVAL, "n_closest_points exceeds the number of stations");
  HIPC(hipSetDevice(h->device));
  MIKC(get_events(h, 2));
  const long npt = h->npt;
  const int K = n_closest;
  long solve_chunks = 0;
  h->tm.rhs_ms = h->tm.contract_ms = h->tm.predict_ms = 0.0;
  h->tm.contract_launches = 0;
  h->tm.contract_flops_executed = 0.0;
  if (npt == 0) {
    h->have_results = true;
    return MIK_OK;
  }
  HIPC(hipStreamWaitEvent(h->stream, h->ev_d2h, 0));
  HIPC(hipEventRecord(h->evpool[0], h->stream));
  // The reference cuts each point's system out of a_all = self._get_kriging_matrix(n); here its entries are computed
  // from the selected stations' coordinates, so no N x N matrix exists on this path (and a factor held by the handle
  // stays valid).
  // K <= MIK_MW_KMAX: candidate lists in registers, systems in LDS, all points in one pass.  Larger K: working sets in
  // HBM, points in chunks that bound those work arrays to ~2 GB.
  const int nb = K + 1;
  const bool custom = h->model == MIK_MODEL_CUSTOM;
  // small windows are solved without a pivot search on the SPD-shifted local system unless the model cannot promise a
  // positive definite station block (hole-effect), has no device functor for the shift (custom), or a previous attempt
  // of this call hit a bad pivot
  const bool mw_piv = custom || h->model == MIK_MODEL_HOLE_EFFECT || h->mw_force_piv || h->opt_mw_pivot;
  // three solvers: LDL^T of the shifted system in registers (no pivot search; windows up to 256), Gauss-Jordan in registers
  // with implicit partial pivoting (when the model cannot promise a positive definite
  // station block; windows up to 127), LU with partial pivoting in HBM scratch (any window)
  const bool chol = !mw_piv && K <= MIK_MW_CHOL_KMAX && h->opt_mw_class != 1;
  // beyond the register classes: blocked Cholesky of the shifted system (one block per point, panels of 64 in LDS, the matrix
  // in an L2-resident scratch slot); "mw_class" 1 forces it for smaller windows too (A/B runs)
  const bool cholb = !mw_piv && !chol && K >= 8;
  const bool big = !chol && !cholb && K > MIK_MW_KMAX;
  long chunk = npt;
  if (K > MIK_MW_KMAX) {  // neighbour lists of 12 K bytes per point: bound them to ~2 GB
    chunk = ((long)(2e9 / (24.0 * K)) / 256) * 256;
    if (chunk < 256) chunk = 256;
    if (chunk > npt) chunk = npt;
  }
  if (custom) {  // the K x K pair distances of every point visit the host: bound that table to ~1 GB
    long cc = ((long)(1e9 / (8.0 * K * (K + 1.0))) / 256) * 256;
    if (cc < 256) cc = 256;
    if (chunk > cc) chunk = cc;
    if (chunk > npt) chunk = npt;
  }
  MIKC(h->mw_idx.ensure(sizeof(int) * (size_t)chunk * K));
  MIKC(h->mw_dist.ensure(sizeof(double) * (size_t)chunk * K));
  MIKC(h->flag.ensure(sizeof(int)));
  HIPC(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  DevBuf su, pu, wd, wi, sysbuf, gtab, gvec, todo;
  if (custom) {
    MIKC(gtab.ensure(sizeof(double) * (size_t)chunk * K * K));
    MIKC(gvec.ensure(sizeof(double) * (size_t)chunk * K));
  }
  const double *sx = h->xs.as<double>(), *sy = h->ys.as<double>(), *sz = h->zs.as<double>();
  const double *qx = h->px.as<double>(), *qy = h->py.as<double>(), *qz = h->pz.as<double>();
  if (h->geo) {
    // neighbours by chord length on the unit sphere (same ordering as great-circle), distances recomputed below
    MIKC(pu.ensure(sizeof(double) * 3 * (size_t)npt));
    double* p3 = pu.as<double>();
    hipLaunchKernelGGL(k_geo_unit, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, h->stream, qx, qy, (int)npt, p3,
                       p3 + npt, p3 + 2 * (size_t)npt);
    qx = p3, qy = p3 + npt, qz = p3 + 2 * (size_t)npt;
  }
  const bool three = h->geo || h->ndim == 3;
  int sgrid = 0;
  int cap = 512;  // candidate buffer of the wave-per-point neighbour search: a power of two >= K + 256
  while (cap < K + 256) cap <<= 1;
  const bool wave_knn = cap <= h->opt_mw_lds_cap;  // default 8192 = 96 KB of LDS; beyond that the lists live in HBM
  // Small windows over a point list in no spatial order (round 4, second session): the lane-per-point search below needs 64
  // consecutive points to share a few cells of the station grid.  The points are then put in Hilbert-curve order on the device
  // (k_ps_*: the sorter of the range-aware contraction), searched and solved in that order -- coordinates gathered once, z and
  // sigma^2 scattered back at the end -- so a shuffled list costs what the rows of a grid cost.
  bool mw_sorted = false;
  double *zout = h->z.as<double>(), *ssout = h->ss.as<double>();
  if (wave_knn) {
    MIKC(build_mw_grid(h, std::max(8, std::min(K, 256))));
    const bool cells = (long)h->grid.nx * h->grid.ny * h->grid.nz > 1;
    const bool coherent = h->pts_step >= 0.0 && 64.0 * h->pts_step <= 10.0 * h->grid.cell;
    if (h->opt_mw_knn_lane && h->opt_sort_points != 0 && K <= 16 && cells && !h->geo && !custom && !coherent && h->pts_extent > 0.0 &&
        npt >= 4096) {
      const double spacing = h->pts_extent / std::pow((double)npt, 1.0 / h->ndim);  // of a sorted list: a wavefront's 64 points are a patch
      if (12.0 * spacing <= 10.0 * h->grid.cell) {                                   // ~8 spacings across
        // (segments of 131 072 points like the contraction's launches: the bounding box and the scan of a segment are ONE workgroup
        // each -- a single 2^20-point segment spent 0.53 + 2 x 0.39 ms in them, eight segments side by side 0.2 ms in all)
        const long schunk = std::min<long>(((npt + 127) / 128) * 128, 131072L);
        if (!(h->ps_valid && h->ps_chunk == schunk)) MIKC(sort_points(h, schunk, (npt + schunk - 1) / schunk));
        const size_t nbp = sizeof(double) * (size_t)npt;
        MIKC(h->ps_x.ensure(nbp));
        MIKC(h->ps_y.ensure(nbp));
        if (h->ndim == 3) MIKC(h->ps_z.ensure(nbp));
        MIKC(h->ps_zs.ensure(nbp));
        MIKC(h->ps_sss.ensure(nbp));
        hipLaunchKernelGGL(k_ps_gather, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, h->stream, (const unsigned*)h->ps_idx[0].as<unsigned>(),
                           npt, qx, qy, h->ndim == 3 ? qz : (const double*)nullptr, h->ps_x.as<double>(), h->ps_y.as<double>(),
                           h->ndim == 3 ? h->ps_z.as<double>() : (double*)nullptr);
        qx = h->ps_x.as<double>(), qy = h->ps_y.as<double>();
        if (h->ndim == 3) qz = h->ps_z.as<double>();
        zout = h->ps_zs.as<double>(), ssout = h->ps_sss.as<double>();
        mw_sorted = true;
      }
    }
  } else {
    MIKC(wd.ensure(sizeof(double) * (size_t)chunk * K));
    MIKC(wi.ensure(sizeof(int) * (size_t)chunk * K));
    if (h->geo) {  // station unit vectors for the plain scan
      MIKC(su.ensure(sizeof(double) * 3 * (size_t)h->N));
      double* s3 = su.as<double>();
      hipLaunchKernelGGL(k_geo_unit, dim3((h->N + 255) / 256), dim3(256), 0, h->stream, sx, sy, h->N, s3, s3 + h->N,
                         s3 + 2 * (size_t)h->N);
      sx = s3, sy = s3 + h->N, sz = s3 + 2 * (size_t)h->N;
    }
  }
  int ldc = 0;
  long cslot = 0;
  if (cholb) {
    ldc = ((K + MIK_MWP - 1) / MIK_MWP) * MIK_MWP;
    cslot = (long)(ldc + MIK_MWP) * ldc + 3L * K;
    cslot += cslot & 1;
    long g = (long)(6e9 / (8.0 * (double)cslot));  // per-block scratch systems, <= ~6 GB in total
    if (g > 2L * h->n_cu) g = 2L * h->n_cu;
    if (g > chunk) g = chunk;
    if (g < 1) g = 1;
    sgrid = (int)g;
    MIKC(sysbuf.ensure(sizeof(double) * (size_t)cslot * (size_t)sgrid));
  }
  if (big) {
    const double per = 8.0 * nb * (nb + 1.0);
    long g = (long)(4e9 / per);  // per-block scratch systems, <= ~4 GB in total
    if (g > 4L * h->n_cu) g = 4L * h->n_cu;
    if (g > chunk) g = chunk;
    if (g < 1) g = 1;
    sgrid = (int)g;
    MIKC(sysbuf.ensure((size_t)per * (size_t)sgrid));
  }
  MIKC(get_events(h, 2 + 2 * (size_t)((npt + chunk - 1) / chunk)));
  for (long p0 = 0; p0 < npt; p0 += chunk) {
    const long pc = (npt - p0 < chunk) ? npt - p0 : chunk;
    const unsigned kgrid = (unsigned)((pc + 255) / 256);
    int* idx = h->mw_idx.as<int>();
    double* dist = h->mw_dist.as<double>();
    if (!wave_knn) {
      if (three)
        hipLaunchKernelGGL(k_mw_knn_big<3>, dim3(kgrid), dim3(256), 0, h->stream, qx + p0, qy + p0, qz + p0, (int)pc, sx, sy, sz,
                           h->N, K, wd.as<double>(), wi.as<int>(), idx, dist);
      else
        hipLaunchKernelGGL(k_mw_knn_big<2>, dim3(kgrid), dim3(256), 0, h->stream, qx + p0, qy + p0, (const double*)nullptr,
                           (int)pc, sx, sy, (const double*)nullptr, h->N, K, wd.as<double>(), wi.as<int>(), idx, dist);
    } else {
      const long wg = 32L * h->n_cu;
      const unsigned wgrid = (unsigned)(pc < wg ? pc : wg);
      const size_t klds = (size_t)cap * (sizeof(double) + sizeof(int));
      KnnArgs ka{};
      ka.px = qx + p0;
      ka.py = qy + p0;
      ka.pz = three ? qz + p0 : nullptr;
      ka.npt = (int)pc;
      ka.gx = h->grid.gx.as<double>();
      ka.gy = h->grid.gy.as<double>();
      ka.gz = h->grid.gz.as<double>();
      ka.orig = h->grid.orig.as<int>();
      ka.cstart = h->grid.cstart.as<int>();
      ka.N = h->N, ka.K = K, ka.CAP = cap;
      ka.nx = h->grid.nx, ka.ny = h->grid.ny, ka.nz = h->grid.nz;
      ka.x0 = h->grid.x0, ka.y0 = h->grid.y0, ka.z0 = h->grid.z0;
      ka.inv_cell = 1.0 / h->grid.cell;
      ka.cell2 = h->grid.cell * h->grid.cell;
      ka.tau0 = 0.0;
      if (h->opt_mw_knn_bound && !h->geo && h->grid.live >= 2 && (long)h->grid.nx * h->grid.ny * h->grid.nz > 1) {
        // radius of the disc / ball expected to hold K + 4 sqrt(K) + 2 of the ~per_cell stations a cell holds; it must stay
        // inside the 3 x 3 (x 3) cells around the point's cell
        const double m = K + 4.0 * std::sqrt((double)K) + 2.0, T = std::max(1.0, h->grid.per_cell);
        const double r2 = h->grid.live == 3 ? std::pow(m / (4.18879020478639 * T), 2.0 / 3.0) : m / (3.14159265358979 * T);
        if (r2 <= 1.0) ka.tau0 = r2 * ka.cell2;
      }
      ka.idx_out = idx;
      ka.dist_out = dist;
      // (measured, profiles/r04_mw_knn_ab.txt: rows of a grid, k = 10: search + rhs 2.65 -> 0.62 ms per 1e6 points, bit-identical; a
      // 32-entry list per lane only ties with the wave-per-point search, and a shuffled point list sends every lane to the list --
      // one same-address atomic per wavefront, +0.3 ms -- hence K <= 16 and the coherence test: 64 consecutive points must span
      // few cells, judged from the median step between consecutive points that mik_set_points / mik_set_grid recorded)
      if (h->opt_mw_knn_lane && K <= 16 && (long)h->grid.nx * h->grid.ny * h->grid.nz > 1 &&
          (mw_sorted || (h->pts_step >= 0.0 && 64.0 * h->pts_step * (h->geo ? MIK_PI / 180.0 : 1.0) <= 10.0 * h->grid.cell))) {
        // small windows: one lane per point over the box of cells its wavefront's 64 consecutive points share (k_mw_knn_lane); the
        // wave-per-point kernel below then only walks the list of points that pass left unfinished
        MIKC(todo.ensure(sizeof(int) * ((size_t)pc + 1)));
        ka.todo_count = todo.as<int>();
        ka.todo = todo.as<int>() + 1;
        HIPC(hipMemsetAsync(ka.todo_count, 0, sizeof(int), h->stream));
        const unsigned lgrid = (unsigned)std::min<long>((pc + 63) / 64, 64L * h->n_cu);
        if (three) hipLaunchKernelGGL((k_mw_knn_lane<3, 16>), dim3(lgrid), dim3(64), 0, h->stream, ka);
        else hipLaunchKernelGGL((k_mw_knn_lane<2, 16>), dim3(lgrid), dim3(64), 0, h->stream, ka);
      }
      if (three) {
        HIPC(hipFuncSetAttribute((const void*)k_mw_knn<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)klds));
        hipLaunchKernelGGL(k_mw_knn<3>, dim3(wgrid), dim3(64), klds, h->stream, ka);
      } else {
        HIPC(hipFuncSetAttribute((const void*)k_mw_knn<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)klds));
        hipLaunchKernelGGL(k_mw_knn<2>, dim3(wgrid), dim3(64), klds, h->stream, ka);
      }
    }
    if (h->geo)
      hipLaunchKernelGGL(k_mw_geo_dist, dim3((unsigned)((pc * K + 255) / 256)), dim3(256), 0, h->stream,
                         (const double*)h->px.as<double>() + p0, (const double*)h->py.as<double>() + p0, pc, K,
                         (const double*)h->xs.as<double>(), (const double*)h->ys.as<double>(), (const int*)idx, dist);
    MwArgs a{};
    a.sx = h->xs.as<double>();
    a.sy = h->ys.as<double>();
    a.sz = h->zs.as<double>();
    a.mode = h->geo ? 1 : h->ndim;
    a.K = K;
    a.npt = (int)pc;
    a.idx = idx;
    a.dist = dist;
    a.Z = h->vals.as<double>();
    a.v = h->v;
    a.exact = h->exact;
    a.eps = h->eps;
    a.z = zout + p0;
    a.ss = ssout + p0;
    a.flag = h->flag.as<int>();
    {  // right-hand sides in place over the distances
      const long ne = pc * K;
      const unsigned rg = (unsigned)((ne + 255) / 256);
      if (custom) {
        // d -> gamma(d) on the host for the point-station distances and for the K x K station pairs of every point
        HIPC(hipMemcpyAsync(gvec.p, dist, sizeof(double) * ne, hipMemcpyDeviceToDevice, h->stream));
        MIKC(custom_roundtrip(h, gvec.as<double>(), pc, K, K));
        hipLaunchKernelGGL(k_mw_rhs_table, dim3(rg), dim3(256), 0, h->stream, dist, (const double*)gvec.as<double>(), ne, h->exact,
                           h->eps);
        hipLaunchKernelGGL(k_mw_pairdist, dim3((unsigned)((ne * K + 255) / 256)), dim3(256), 0, h->stream, (const int*)idx, pc, K,
                           a.sx, a.sy, a.sz, a.mode, gtab.as<double>());
        MIKC(custom_roundtrip(h, gtab.as<double>(), pc * K, K, K));
        a.gtab = gtab.as<double>();
      } else
      switch (h->model) {
        case 0: hipLaunchKernelGGL(k_mw_rhs<0>, dim3(rg), dim3(256), 0, h->stream, dist, ne, h->v, h->exact, h->eps); break;
        case 1: hipLaunchKernelGGL(k_mw_rhs<1>, dim3(rg), dim3(256), 0, h->stream, dist, ne, h->v, h->exact, h->eps); break;
        case 2: hipLaunchKernelGGL(k_mw_rhs<2>, dim3(rg), dim3(256), 0, h->stream, dist, ne, h->v, h->exact, h->eps); break;
        case 3: hipLaunchKernelGGL(k_mw_rhs<3>, dim3(rg), dim3(256), 0, h->stream, dist, ne, h->v, h->exact, h->eps); break;
        case 4: hipLaunchKernelGGL(k_mw_rhs<4>, dim3(rg), dim3(256), 0, h->stream, dist, ne, h->v, h->exact, h->eps); break;
        default: hipLaunchKernelGGL(k_mw_rhs<5>, dim3(rg), dim3(256), 0, h->stream, dist, ne, h->v, h->exact, h->eps); break;
      }
    }
    HIPC(hipEventRecord(h->evpool[2 + 2 * solve_chunks], h->stream));
    if (big) {
      const size_t lds = sizeof(double) * 2 * (size_t)nb + sizeof(int) * (size_t)nb;
      if (lds > 150 * 1024) return fail(MIK_EINVAL, "n_closest_points too large for the device path (> ~7600)");
      const int grid = (int)(pc < sgrid ? pc : sgrid);
      HIPC(hipFuncSetAttribute((const void*)k_mw_solve_big, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_mw_solve_big, dim3(grid), dim3(256), lds, h->stream, a, sysbuf.as<double>());
    } else if (cholb) {
      const size_t lds = sizeof(double) * 2 * MIK_MWP * MIK_MWP_LD;
      const int grid = (int)std::min<long>(sgrid, pc);
      HIPC(hipFuncSetAttribute((const void*)k_mw_chol_blocked, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k_mw_chol_blocked, dim3(grid), dim3(256), lds, h->stream, a, sysbuf.as<double>(), cslot, ldc);
    } else if (chol) {
      MIKC(dispatch_mw_chol(h, a, pc));
    } else {
      MIKC(dispatch_mw_solve(h, a, pc));  // (always the pivoting form: valid for every window, the only one built)
    }
    HIPC(hipEventRecord(h->evpool[3 + 2 * solve_chunks], h->stream));
    ++solve_chunks;
    HIPC(hipGetLastError());
  }
  if (mw_sorted)  // back to the caller's order
    hipLaunchKernelGGL(k_ps_unsort, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, h->stream, (const unsigned*)h->ps_idx[0].as<unsigned>(), npt,
                       (const double*)zout, (const double*)ssout, h->z.as<double>(), h->ss.as<double>());
  h->tm.points_sorted = mw_sorted ? 1 : 0;
  int flag = 0;
  HIPC(hipMemcpyAsync(&flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipEventRecord(h->evpool[1], h->stream));
  HIPC(hipStreamSynchronize(h->stream));  // also: the scoped work buffers are released only after the stream drained
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, h->evpool[0], h->evpool[1]));
  h->tm.predict_ms = ms;
  for (long c = 0; c < solve_chunks; ++c) {  // the per-point solves (the dominant kernel of this path) on their own
    HIPC(hipEventElapsedTime(&ms, h->evpool[2 + 2 * c], h->evpool[3 + 2 * c]));
    h->tm.contract_ms += ms;
  }
  h->tm.contract_launches = solve_chunks;
  h->tm.mw_kernel = chol ? 1 : cholb ? 4 : (big ? 3 : 2);
  h->tm.rhs_ms = h->tm.predict_ms - h->tm.contract_ms;  // neighbour search + right-hand sides
  if ((flag & 2) && !mw_piv) {  // a local system was not positive definite after the shift: redo with partial pivoting
    h->mw_force_piv = true;
    const int rc = one_predict_mw(h, n_closest);
    h->mw_force_piv = false;
    return rc;
  }
  if (flag) return fail(MIK_ESINGULAR, "Singular matrix");  // cok.pyx:176-177
  MIKC(h->pin_out.ensure(sizeof(double) * 2 * (size_t)npt));
  HIPC(hipMemcpyAsync(h->pin_out.as<double>(), h->z.p, sizeof(double) * npt, hipMemcpyDeviceToHost, h->stream_d2h));
  HIPC(hipMemcpyAsync(h->pin_out.as<double>() + npt, h->ss.p, sizeof(double) * npt, hipMemcpyDeviceToHost, h->stream_d2h));
  HIPC(hipEventRecord(h->ev_d2h, h->stream_d2h));
  h->have_results = true;
  return MIK_OK;
}
