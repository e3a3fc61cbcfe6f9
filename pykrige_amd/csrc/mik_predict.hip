// mik_predict.hip -- K3: right-hand sides + dense / range-aware contraction of the resident points
// One translation unit of libmikrige.so (pykrige_amd/build.py compiles them in parallel).
#include "mik_k_predict.h"
#include "mik_host.h"

// Hilbert-curve order of the resident points inside every launch of `chunk` points (k_ps_*, mik_kernels.h): ps_idx[0][s] = index of
// the point at sorted position s.  On the handle's stream; two radix passes of 10-bit digits, all segments side by side.
// unit vectors of the resident (geographic) points, gu[0 .. npt) x, [npt .. 2 npt) y, [2 npt .. 3 npt) z; on the handle's stream
static int geo_point_vectors(mik_handle* h) {
  const long npt = h->npt;
  MIKC(h->gu.ensure(sizeof(double) * 3 * (size_t)npt));
  double* u = h->gu.as<double>();
  hipLaunchKernelGGL(k_geo_unit_p, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, h->stream, (const double*)h->px.as<double>(),
                     (const double*)h->py.as<double>(), npt, u, u + npt, u + 2 * npt);
  return MIK_OK;
}

int sort_points(mik_handle* h, long chunk, long nchunks) {
  const long npt = h->npt;
  const int kd = h->ndim;  // (geographic points: the curve runs through (lon, lat), like the stations' -- mikrige.hip, station_order)
  const int bits = ps_bits(kd), bps = (int)((chunk + MIK_PS_TILE - 1) / MIK_PS_TILE);
  for (int q = 0; q < 2; ++q) {
    MIKC(h->ps_key[q].ensure(sizeof(unsigned) * (size_t)npt));
    MIKC(h->ps_idx[q].ensure(sizeof(unsigned) * (size_t)npt));
  }
  MIKC(h->ps_table.ensure(sizeof(unsigned) * (size_t)nchunks * (1u << MIK_PS_DB) * (size_t)bps));
  MIKC(h->ps_box.ensure(sizeof(double) * 4 * (size_t)nchunks));
  const double *px = h->px.as<double>(), *py = h->py.as<double>(), *pz = h->ndim == 3 ? h->pz.as<double>() : nullptr;
  hipStream_t st = h->stream;
  hipLaunchKernelGGL(k_ps_bbox, dim3((unsigned)nchunks), dim3(1024), 0, st, px, py, pz, npt, chunk, bits, h->ps_box.as<double>());
  hipLaunchKernelGGL(k_ps_keys, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, st, px, py, pz, npt, chunk, kd, bits,
                     (const double*)h->ps_box.as<double>(), h->ps_key[0].as<unsigned>(), h->ps_idx[0].as<unsigned>());
  for (int pass = 0; pass < 2; ++pass) {
    const unsigned* kin = h->ps_key[pass].as<unsigned>();
    const unsigned* iin = h->ps_idx[pass].as<unsigned>();
    hipLaunchKernelGGL(k_ps_hist, dim3((unsigned)(nchunks * bps)), dim3(256), 0, st, kin, npt, chunk, bps, MIK_PS_DB * pass,
                       h->ps_table.as<unsigned>());
    hipLaunchKernelGGL(k_ps_scan, dim3((unsigned)nchunks), dim3(1 << MIK_PS_DB), 0, st, h->ps_table.as<unsigned>(), bps);
    hipLaunchKernelGGL(k_ps_scatter, dim3((unsigned)(nchunks * bps)), dim3(256), 0, st, kin, iin, npt, chunk, bps, MIK_PS_DB * pass,
                       (const unsigned*)h->ps_table.as<unsigned>(), h->ps_key[pass ^ 1].as<unsigned>(), h->ps_idx[pass ^ 1].as<unsigned>());
  }
  HIPC(hipGetLastError());
  h->ps_valid = true;
  h->ps_chunk = chunk;
  return MIK_OK;
}

int one_predict(mik_handle* h) {
  if (!h || !h->have_factor) return fail(MIK_ESTATE, "mik_predict: factor first");
  if (!h->have_points) return fail(MIK_ESTATE, "mik_predict: set points first");
  HIPC(hipSetDevice(h->device));
  const long npt = h->npt;
  const int Mp = h->Mp, nIblk = Mp / 128;
  h->tm.rhs_ms = h->tm.contract_ms = h->tm.predict_ms = 0.0;
  h->tm.contract_launches = 0;
  h->tm.contract_flops_executed = 0.0;
  h->tm.symmetric = h->opt_sym;
  h->tm.engine = 0;  // (the v_fma_f64 contraction left the library in round 6: tools/kernel_bench)
  h->tm.mw_kernel = 0;
  if (npt == 0) {
    h->have_results = true;
    return MIK_OK;
  }
  long chunk = std::min<long>(h->opt_chunk, ((npt + 127) / 128) * 128);
  if (h->model == MIK_MODEL_CUSTOM) chunk = std::min<long>(chunk, 16384);  // each chunk's distances visit the host
  // range-aware contraction (k_contract_sp): the factor is in Hilbert-curve station order and the variogram has compact support
  const bool sparse = h->factor_sorted && h->opt_sparse != 2 && h->opt_sparse != 0;
  if (sparse) chunk = std::min<long>(chunk, 131072);  // k_sp_tiles: at most 1024 point blocks per launch
  const int nK16 = Mp / 16;
  // tiles of gathered 16-row groups (k_contract_spg) wherever 32-bit LDS-DMA offsets reach every row of the inverse
  const bool gathered = sparse && h->opt_sparse_rows != 128 && (double)Mp * (double)Mp * 8.0 < 4294967296.0;
  // gathered row groups (round 5): flags and lists per 8 stations (candidates stay per 16), a K step = a pair of list-adjacent 8-station tiles
  // (k_contract_spg H8); the aligned-block fallback keeps 16-station lists.  nKt = tiles per point block in the units of this launch's lists.
  // (Round 4's 16-station lists under gathered groups and the epilogue from global memory -- "sparse_ktile" 16, "sparse_epilogue" 0 -- lost
  // their A/B in round 5 and left the library in round 6.)
  const bool h8 = gathered;
  const int nKt = h8 ? Mp / 8 : nK16;
  h->tm.sparse_ktile = !sparse ? 0 : h8 ? 8 : 16;
  h->tm.sparse = sparse ? 1 : 0;
  h->tm.sparse_rows = sparse ? (gathered ? 16 : 128) : 0;
  h->tm.stations_sorted = h->factor_sorted ? 1 : 0;
  h->tm.sparse_tiles = h->tm.sparse_tiles_dense = h->tm.sparse_ktiles = h->tm.sparse_ktiles_dense = h->tm.sparse_lists_ms = 0.0;
  h->tm.sparse_diag_products = 0.0;
  // the points of every launch in Hilbert-curve order among themselves (compact point blocks: option "sort_points")
  // (auto: not for small jobs -- seven more launches, 0.07 ms, against a contraction of microseconds; one tile per point block anyway
  // while the matrix has fewer than 512 rows)
  const bool sortpts = sparse && (h->opt_sort_points == 1 || (h->opt_sort_points < 0 && npt >= 4096 && Mp >= 512));
  h->tm.points_sorted = sortpts ? 1 : 0;
  h->tm.sort_points_ms = 0.0;
  // "rhs_overlap" (off by default, see the option): two RHS panels, k_rhs of chunk c + 1 on a second stream while chunk c is
  // contracted.
  const bool overlap = h->opt_rhs_overlap && h->model != MIK_MODEL_CUSTOM && !sparse;
  const bool lanes2_wanted = sparse && h->opt_sparse_lanes == 2;
  // keep the RHS panels under ~1/4 of device memory
  size_t freeb = 0, totalb = 0;
  HIPC(hipMemGetInfo(&freeb, &totalb));
  const size_t have = h->Bt.bytes + h->Bt2.bytes;
  while (chunk > 128 && (size_t)chunk * Mp * sizeof(double) * ((overlap || lanes2_wanted) ? 2 : 1) > std::max(freeb + have, have) / 2) chunk = ((chunk / 2 + 127) / 128) * 128;
  // equal chunks: ceil(npt / chunk) launches of the same size (a short last launch drains as long as a full one)
  long nchunks = (npt + chunk - 1) / chunk;
  chunk = (((npt + nchunks - 1) / nchunks + 127) / 128) * 128;
  nchunks = (npt + chunk - 1) / chunk;
  MIKC(h->Bt.ensure(sizeof(double) * (size_t)chunk * Mp));
  if (overlap && nchunks > 1) MIKC(h->Bt2.ensure(sizeof(double) * (size_t)chunk * Mp));
  const bool two = overlap && nchunks > 1;
  MIKC(h->part.ensure(sizeof(double) * (size_t)chunk * nIblk));
  MIKC(h->pin_out.ensure(sizeof(double) * 2 * (size_t)npt));  // (a previous result may have left with mik_take_results)
  MIKC(get_events(h, 2 + 6 * (size_t)nchunks));
  std::vector<unsigned long long> sp_host;
  const bool lanes2 = lanes2_wanted && nchunks > 1;
  struct SpLane {
    DevBuf *cand, *flags, *klist, *kcount, *nrows, *rows, *rstart, *tiles, *xoff, *part, *queue, *Bt, *recs;
    hipStream_t st;
  };
  SpLane lane[2] = {{&h->sp_cand, &h->sp_flags, &h->sp_klist, &h->sp_kcount, &h->sp_nrows, &h->sp_rows, &h->sp_rstart, &h->sp_tiles, &h->sp_xoff,
                     &h->part, &h->queue, &h->Bt, &h->sp_recs, h->stream},
                    {&h->sp2_cand, &h->sp2_flags, &h->sp2_klist, &h->sp2_kcount, &h->sp2_nrows, &h->sp2_rows, &h->sp2_rstart, &h->sp2_tiles,
                     &h->sp2_xoff, &h->part2, &h->queue2, &h->Bt2, &h->sp2_recs, h->stream2}};
  if (sparse) {
    const size_t nTb = (size_t)chunk / 128;
    for (int L = 0; L < (lanes2 ? 2 : 1); ++L) {
      MIKC(lane[L].cand->ensure(nTb * nK16));
      MIKC(lane[L].flags->ensure(nTb * nKt));
      MIKC(lane[L].klist->ensure(sizeof(unsigned short) * nTb * nKt));
      MIKC(lane[L].kcount->ensure(sizeof(int) * nTb));
      MIKC(lane[L].nrows->ensure(sizeof(int) * nTb));
      if (gathered) {
        MIKC(lane[L].recs->ensure((h8 ? 48 : 32) * nTb * nIblk));  // ceil(nk / 8) <= nK16 / 8 = nIblk tiles per point block
      } else {
        MIKC(lane[L].rows->ensure(sizeof(unsigned short) * nTb * nIblk));
        MIKC(lane[L].rstart->ensure(sizeof(unsigned short) * nTb * nIblk));
        MIKC(lane[L].tiles->ensure(sizeof(unsigned) * nTb * nIblk));
      }
      MIKC(lane[L].xoff->ensure(sizeof(int) * 9));
      MIKC(lane[L].queue->ensure(8 * sizeof(unsigned long long)));
      if (L == 1) {
        MIKC(h->Bt2.ensure(sizeof(double) * (size_t)chunk * Mp));
        MIKC(h->part2.ensure(sizeof(double) * (size_t)chunk * nIblk));
      }
    }
    MIKC(h->sp_stats.ensure(sizeof(unsigned long long) * 4 * (size_t)nchunks));
    sp_host.assign(4 * (size_t)nchunks, 0ULL);
  }
  while (h->pr_events.size() < 2 * (size_t)nchunks) {
    hipEvent_t e;
    HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->pr_events.push_back(e);
  }
  const int kend = ((h->M + MIK_BK - 1) / MIK_BK) * MIK_BK;
  hipStream_t sc = h->stream;                  // contraction, reduction
  hipStream_t sr = two ? h->stream2 : h->stream;  // right-hand sides
  HIPC(hipStreamWaitEvent(h->stream, h->ev_d2h, 0));  // an earlier predict's result copies still read z / ss
  if (sparse && h->geo) MIKC(geo_point_vectors(h));  // what the candidate boxes of a geographic problem are built from
  HIPC(hipEventRecord(h->evpool[0], h->stream));
  bool sorted_now = false;
  if (sortpts && !(h->ps_valid && h->ps_chunk == chunk)) {
    MIKC(sort_points(h, chunk, nchunks));
    HIPC(hipEventRecord(h->ev_sort, h->stream));
    sorted_now = true;
  }
  const unsigned* perm_all = sortpts ? h->ps_idx[0].as<unsigned>() : nullptr;
  if (two || lanes2) HIPC(hipStreamWaitEvent(h->stream2, h->evpool[0], 0));
  if (lanes2 && sorted_now) HIPC(hipStreamWaitEvent(h->stream2, h->ev_sort, 0));
  auto launch_rhs = [&](long c) -> int {
    const long t0 = c * chunk;
    const int nvalid = (int)std::min<long>(chunk, npt - t0);
    const int palloc = ((nvalid + 127) / 128) * 128;
    RhsArgs a{};
    a.Bt = (two && (c & 1)) ? h->Bt2.as<double>() : h->Bt.as<double>();
    a.ld = Mp;
    a.palloc = palloc;
    a.nvalid = nvalid;
    a.px = h->px.as<double>() + t0;
    a.py = h->py.as<double>() + t0;
    a.pz = h->ndim == 3 ? h->pz.as<double>() + t0 : nullptr;
    a.N = h->N;
    a.p = h->p;
    a.M = h->M;
    a.Mp = Mp;
    a.ndim = h->ndim;
    a.xs = h->factor_sorted ? h->xs_s.as<double>() : h->xs.as<double>();
    a.ys = h->factor_sorted ? h->ys_s.as<double>() : h->ys.as<double>();
    a.zs = h->factor_sorted ? h->zs_s.as<double>() : h->zs.as<double>();
    a.dsc = h->factor_eq ? h->dsc.as<double>() : nullptr;
    a.v = h->v;
    a.exact = h->exact;
    a.eps = h->eps;
    a.rl = h->rl;
    a.nwells = h->nwells;
    a.nextra = h->nextra;
    a.wells = h->wells.as<double>();
    a.extra = h->nextra ? h->extra_rows.as<double>() + t0 : nullptr;
    a.extra_stride = npt;
    a.cvec = h->cvec.as<double>();
    a.zout = h->z.as<double>() + t0;
    if (two && c >= 2) HIPC(hipStreamWaitEvent(sr, h->pr_events[2 * (c - 2) + 1], 0));  // the contraction that read this panel is done
    if (sparse) {
      // candidates (bounding boxes), cleared flags, then delta for the candidate blocks only
      const SpLane& ln = lane[lanes2 ? (c & 1) : 0];
      hipStream_t ss = ln.st;
      a.Bt = ln.Bt->as<double>();
      a.cand = ln.cand->as<unsigned char>();
      a.flags = ln.flags->as<unsigned char>();
      a.nIblk = nIblk;
      a.nK16 = nK16;
      a.nKf = nKt;
      a.sill = h->v.p0 + h->v.p2;
      if (perm_all) {  // sorted order: the chunk's points are reached through perm, from the list's base pointers
        a.perm = perm_all + t0;
        a.px = h->px.as<double>();
        a.py = h->py.as<double>();
        a.pz = h->ndim == 3 ? h->pz.as<double>() : nullptr;
        a.extra = h->nextra ? h->extra_rows.as<double>() : nullptr;
        a.zout = h->z.as<double>();
      }
      HIPC(hipEventRecord(h->evpool[2 + 4 * nchunks + 2 * c], ss));
      // candidates by bounding boxes: Euclidean coordinates against the range; geographic: unit vectors against the CHORD of the range
      // (2 sin(arc / 2), the range being degrees of arc; beyond 180 degrees everything is in range)
      const double *cx = a.px, *cy = a.py, *cz = a.pz;
      double radius = std::max(h->v.p1, h->eps);
      if (h->geo) {
        const long off = perm_all ? 0 : t0;
        cx = h->gu.as<double>() + off, cy = cx + npt, cz = cy + npt;
        radius = radius >= 180.0 ? 4.0 : 2.0 * std::sin(radius * 3.14159265358979323846 / 360.0) * (1.0 + 1e-12) + 1e-15;
      }
      hipLaunchKernelGGL(k_sp_cand, dim3(palloc / 128), dim3(128), 0, ss, cx, cy, cz, nvalid, (const double*)h->sbox.as<double>(), nK16,
                         h->N / 16, (h->M + 15) / 16, radius, ln.cand->as<unsigned char>(), a.perm, gathered ? 0 : 1, ln.flags->as<unsigned char>(), nKt);
      HIPC(hipEventRecord(h->evpool[2 + 4 * c], ss));
      if (h8) {
        if (h->geo) hipLaunchKernelGGL((k_rhs<3, 1, true, true>), dim3(palloc / MIK_TP), dim3(256), 0, ss, a);
        else if (h->ndim == 3) hipLaunchKernelGGL((k_rhs<3, 3, true, true>), dim3(palloc / MIK_TP), dim3(256), 0, ss, a);
        else hipLaunchKernelGGL((k_rhs<3, 2, true, true>), dim3(palloc / MIK_TP), dim3(256), 0, ss, a);
      } else if (h->geo) hipLaunchKernelGGL((k_rhs<3, 1, true>), dim3(palloc / MIK_TP), dim3(256), 0, ss, a);
      else if (h->ndim == 3) hipLaunchKernelGGL((k_rhs<3, 3, true>), dim3(palloc / MIK_TP), dim3(256), 0, ss, a);
      else hipLaunchKernelGGL((k_rhs<3, 2, true>), dim3(palloc / MIK_TP), dim3(256), 0, ss, a);
      HIPC(hipEventRecord(h->evpool[3 + 4 * c], ss));
      return MIK_OK;
    }
    HIPC(hipEventRecord(h->evpool[2 + 4 * c], sr));
    if (h->model == MIK_MODEL_CUSTOM) {
      DISPATCH_NDIM_FIXED(7, h->geo ? 1 : h->ndim, k_rhs, dim3(palloc / MIK_TP), dim3(256), sr, a);
      MIKC(custom_roundtrip(h, a.Bt, nvalid, h->N, Mp));
      DISPATCH_NDIM_FIXED(6, h->geo ? 1 : h->ndim, k_rhs, dim3(palloc / MIK_TP), dim3(256), sr, a);
    } else {
      DISPATCH_MODEL_NDIM(h->model, h->geo ? 1 : h->ndim, k_rhs, dim3(palloc / MIK_TP), dim3(256), sr, a);
    }
    HIPC(hipEventRecord(h->evpool[3 + 4 * c], sr));
    if (two) HIPC(hipEventRecord(h->pr_events[2 * c], sr));
    return MIK_OK;
  };
  if (two) MIKC(launch_rhs(0));
  for (long c = 0; c < nchunks; ++c) {
    const long t0 = c * chunk;
    const int nvalid = (int)std::min<long>(chunk, npt - t0);
    const int palloc = ((nvalid + 127) / 128) * 128;
    if (two) {
      if (c + 1 < nchunks) MIKC(launch_rhs(c + 1));  // queued behind chunk c's right-hand sides on the second stream
      HIPC(hipStreamWaitEvent(sc, h->pr_events[2 * c], 0));
    } else {
      MIKC(launch_rhs(c));
    }
    hipEvent_t e1 = h->evpool[4 + 4 * c], e2 = h->evpool[5 + 4 * c];
    if (sparse) {
      const int nTb = palloc / 128;
      const SpLane& ln = lane[lanes2 ? (c & 1) : 0];
      hipStream_t sc = ln.st;  // (shadows the dense path's stream: this launch lives on its lane's)
      if (gathered) {
        hipLaunchKernelGGL(k_sp_lists_g, dim3(nTb), dim3(64), 0, sc, (const unsigned char*)ln.flags->as<unsigned char>(), nKt,
                           ln.klist->as<unsigned short>(), ln.kcount->as<int>(), ln.nrows->as<int>(), h8 ? 1 : 0);
        hipLaunchKernelGGL(k_sp_tiles_g<true>, dim3((unsigned)((nTb + h->opt_sparse_group - 1) / h->opt_sparse_group)), dim3(256), 0, sc, (const int*)ln.nrows->as<int>(),
                           (const int*)ln.kcount->as<int>(), (const unsigned short*)ln.klist->as<unsigned short>(), nKt, nTb, ln.recs->as<uint4>(),
                           ln.xoff->as<int>(), h->sp_stats.as<unsigned long long>() + 4 * c, h->opt_sparse_group, ln.queue->as<unsigned long long>());
      } else {
        hipLaunchKernelGGL(k_sp_lists, dim3(nTb), dim3(64), 0, sc, (const unsigned char*)ln.flags->as<unsigned char>(), nK16, nIblk,
                           ln.klist->as<unsigned short>(), ln.kcount->as<int>(), ln.rows->as<unsigned short>(),
                           ln.rstart->as<unsigned short>(), ln.nrows->as<int>());
        hipLaunchKernelGGL(k_sp_tiles, dim3(1), dim3(1024), 0, sc, (const int*)ln.nrows->as<int>(), (const int*)ln.kcount->as<int>(),
                           (const unsigned short*)ln.rstart->as<unsigned short>(), nIblk, nTb, ln.tiles->as<unsigned>(),
                           ln.xoff->as<int>(), h->sp_stats.as<unsigned long long>() + 4 * c);
      }
      HIPC(hipEventRecord(h->evpool[3 + 4 * nchunks + 2 * c], sc));
      if (!gathered) HIPC(hipMemsetAsync(ln.queue->p, 0, 8 * sizeof(unsigned long long), sc));  // (gathered: k_sp_tiles_g zeroes the queues)
      SpArgs sa{};
      sa.Ainv = h->T.as<double>();
      sa.lda = Mp;
      sa.Bt = ln.Bt->as<double>();
      sa.ldb = Mp;
      sa.part = ln.part->as<double>();
      sa.palloc = palloc;
      sa.kend = kend;
      sa.nIblk = nIblk;
      sa.nK16 = nK16;
      sa.klist = ln.klist->as<unsigned short>();
      sa.kcount = ln.kcount->as<int>();
      sa.rows = ln.rows->as<unsigned short>();
      sa.rstart = ln.rstart->as<unsigned short>();
      sa.tiles = ln.tiles->as<unsigned>();
      sa.xoff = ln.xoff->as<int>();
      sa.queue = ln.queue->as<unsigned long long>();
      // two lanes: the CONTRACTIONS run one after the other (this one behind the other lane's previous one); what overlaps a contraction is the
      // other lane's candidate / right-hand-side / list kernels in its tail.  Two persistent launches side by side share every CU and mix two
      // tile queues in every XCD's L2: measured 3.5 % slower per pair (profiles/r06_predict_timeline_c5.txt; rounds 4-5 got this order by accident:
      // the queue memset in front of the contraction waited for a free CU).
      if (lanes2 && c > 0) HIPC(hipStreamWaitEvent(sc, h->evpool[5 + 4 * (c - 1)], 0));
      HIPC(hipEventRecord(e1, sc));
      if (gathered) {
        SpgArgs ga{};
        ga.Ainv = sa.Ainv;
        ga.lda = Mp;
        ga.Bt = sa.Bt;
        ga.ldb = Mp;
        ga.part = sa.part;
        ga.palloc = palloc;
        ga.nK16 = nKt;
        ga.klist = sa.klist;
        ga.recs = ln.recs->as<uint4>();
        ga.xoff = sa.xoff;
        ga.queue = sa.queue;
        static const bool spg_prof = getenv("MIK_SPG_PROF") && atoi(getenv("MIK_SPG_PROF")) != 0;
        if (spg_prof) {  // diagnostic (MIK_SPG_PROF=1): cycle sums per phase of the tile loop, one launch, printed to stderr
          const unsigned nb = (unsigned)std::min<long>(2L * h->n_cu, (long)nTb * nIblk);
          static DevBuf pb;
          MIKC(pb.ensure(sizeof(unsigned long long) * nb * 96));
          HIPC(hipMemsetAsync(pb.p, 0, pb.bytes, sc));
          ga.prof = pb.as<unsigned long long>();
          hipLaunchKernelGGL((k_contract_spg<2, true, true, true>), dim3(nb), dim3(512), 0, sc, ga);
          std::vector<unsigned long long> hp((size_t)nb * 96);
          HIPC(hipMemcpyAsync(hp.data(), pb.p, sizeof(unsigned long long) * hp.size(), hipMemcpyDeviceToHost, sc));
          HIPC(hipStreamSynchronize(sc));
          static const char* names[8] = {"top drain+barrier", "off-diagonal K steps", "triangle K steps", "acquire", "adopt", "epilogue loads+sums", "reduce+store", "-"};
          {
            double c[16] = {0};
            for (unsigned b = 0; b < nb; ++b)
              for (int i = 0; i < 16; ++i) c[i] += (double)hp[(size_t)nb * 80 + (size_t)b * 16 + i];
            fprintf(stderr, "spg triangle steps (cycles per visit):");
            for (int w = 7; w >= 0; --w) fprintf(stderr, "  w=%d %.0f", w, c[w] / std::max(1.0, c[8 + w]));
            fprintf(stderr, "\n");
          }
          for (int wv : {0, 6}) {
            double sum[10] = {0};
            for (unsigned b = 0; b < nb; ++b)
              for (int i = 0; i < 10; ++i) sum[i] += (double)hp[((size_t)b * 8 + wv) * 10 + i];
            double tot = 0;
            for (int i = 0; i < 7; ++i) tot += sum[i];
            fprintf(stderr, "spg phases, wavefront %d: %.0f tiles of the launch, %.1f per block, %.1f off-diagonal steps per tile, %.0f cycles per tile\n", wv, sum[8], sum[8] / nb, sum[9] / std::max(1.0, sum[8]), tot / std::max(1.0, sum[8]));
            for (int i = 0; i < 7; ++i) fprintf(stderr, "   %-22s %8.0f cycles per tile  %5.1f %%\n", names[i], sum[i] / std::max(1.0, sum[8]), 100.0 * sum[i] / tot);
            fprintf(stderr, "   per off-diagonal step %.0f cycles\n", sum[1] / std::max(1.0, sum[9]));
          }
        } else
        // (2 n_cu persistent blocks: leaving 32 .. 128 of the slots to the other lane's preparation kernels was tried -- they then run beside the
        // contraction at a fraction of the chip -- and measured a tie at 32 and 1 - 3 % slower beyond: profiles/r06_predict_timeline_c5_after.txt)
        hipLaunchKernelGGL((k_contract_spg<2, true, true>), dim3((unsigned)std::min<long>(2L * h->n_cu, (long)nTb * nIblk)), dim3(512), 0, sc, ga);
      } else {
        hipLaunchKernelGGL((k_contract_sp<2>), dim3((unsigned)std::min<long>(2L * h->n_cu, (long)nTb * nIblk)), dim3(512), 0, sc, sa);
      }
      HIPC(hipEventRecord(e2, sc));
      hipLaunchKernelGGL(k_ss_reduce_sp, dim3((nvalid + 255) / 256), dim3(256), 0, sc, (const double*)ln.part->as<double>(), palloc,
                         (const int*)ln.nrows->as<int>(), nvalid, 2.0 * (h->v.p0 + h->v.p2),
                         perm_all ? h->ss.as<double>() : h->ss.as<double>() + t0, perm_all ? perm_all + t0 : (const unsigned*)nullptr);
      if (lanes2 && (c & 1)) HIPC(hipEventRecord(h->pr_events[0], sc));  // lane 1's latest launch (joined below)
      HIPC(hipEventRecord(h->ev_chunk, sc));
      HIPC(hipStreamWaitEvent(h->stream_d2h, h->ev_chunk, 0));
      HIPC(hipMemcpyAsync(h->pin_out.as<double>() + t0, h->z.as<double>() + t0, sizeof(double) * nvalid, hipMemcpyDeviceToHost,
                          h->stream_d2h));
      HIPC(hipMemcpyAsync(h->pin_out.as<double>() + npt + t0, h->ss.as<double>() + t0, sizeof(double) * nvalid,
                          hipMemcpyDeviceToHost, h->stream_d2h));
      h->tm.sparse_tiles_dense += (double)nTb * nIblk;
      h->tm.sparse_ktiles_dense += (double)nTb * (kend / 16.0) * (nIblk - 1) / 2.0;  // off-diagonal K tiles of the dense symmetric form (about)
      continue;
    }
    HIPC(hipEventRecord(e1, sc));
    const long tiles = (long)nIblk * (palloc / 128);
    const unsigned grid = (unsigned)(8 * ((tiles + 7) / 8));
    {
      const double* Ai = h->T.as<double>();
      const double* Bi = (two && (c & 1)) ? h->Bt2.as<double>() : h->Bt.as<double>();
      double* pp = h->part.as<double>();
      const long ldm = Mp;
      const unsigned sgrid = (unsigned)super_grid(nIblk, palloc / 128);
      // persistent launch: 2 blocks per CU pop tiles from per-XCD sequences (8 counters, zeroed per launch); 8 wavefronts per tile
      // (wave tile 32 x 64).  Three forms: the symmetric half product with triangular diagonal blocks (default), with whole
      // diagonal blocks ("tri" 0) and the reference's full product w = A_inv b ("symmetric" 0) -- the cross-checks of the parity tests.
      // (The 4-wave tiles, the v_fma_f64 engine, pair units, popped-ahead tiles: every A/B of rounds 2-5 lost; tools/kernel_bench.)
      MIKC(h->queue.ensure(8 * sizeof(unsigned long long)));
      HIPC(hipMemsetAsync(h->queue.p, 0, 8 * sizeof(unsigned long long), sc));
      unsigned long long* qp = h->queue.as<unsigned long long>();
      const unsigned pgrid = (unsigned)std::min<long>(2L * h->n_cu, (long)sgrid);
      if (h->opt_sym && h->opt_tri) hipLaunchKernelGGL((k_contract<true, 2, true, false, true>), dim3(pgrid), dim3(512), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend, qp);
      else if (h->opt_sym) hipLaunchKernelGGL((k_contract<true, 2>), dim3(pgrid), dim3(512), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend, qp);
      else hipLaunchKernelGGL((k_contract<false, 2>), dim3(pgrid), dim3(512), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend, qp);
    }
    HIPC(hipEventRecord(e2, sc));
    if (two) HIPC(hipEventRecord(h->pr_events[2 * c + 1], sc));
    hipLaunchKernelGGL(k_ss_reduce, dim3((nvalid + 255) / 256), dim3(256), 0, sc, (const double*)h->part.as<double>(),
                       palloc, nIblk, nvalid, h->ss.as<double>() + t0);
    // this chunk's z and sigma^2 leave for the page-locked landing zone while the next chunk is computed
    HIPC(hipEventRecord(h->ev_chunk, sc));
    HIPC(hipStreamWaitEvent(h->stream_d2h, h->ev_chunk, 0));
    HIPC(hipMemcpyAsync(h->pin_out.as<double>() + t0, h->z.as<double>() + t0, sizeof(double) * nvalid, hipMemcpyDeviceToHost,
                        h->stream_d2h));
    HIPC(hipMemcpyAsync(h->pin_out.as<double>() + npt + t0, h->ss.as<double>() + t0, sizeof(double) * nvalid,
                        hipMemcpyDeviceToHost, h->stream_d2h));
    // executed flops of this launch: per tile 2*128*128*(k extent)
    // (triangular diagonal blocks: nt (nt + 1) / 2 products of 16 rows x 16 k instead of 8 nt, nt = K tiles of the block)
    const bool tri = h->opt_sym && h->opt_tri;
    double kext = 0.0;
    for (int ib = 0; ib < nIblk; ++ib) {
      const int ext = h->opt_sym ? std::max(0, kend - ib * 128) : kend;
      if (tri) {
        const int nt = std::min(ext, 128) / 16;
        kext += (ext - 16 * nt) + 16.0 * (nt * (nt + 1) / 2) / 8.0;
      } else kext += ext;
    }
    h->tm.contract_flops_executed += 2.0 * 128.0 * 128.0 * kext * (palloc / 128);
  }
  HIPC(hipGetLastError());
  if (lanes2) HIPC(hipStreamWaitEvent(h->stream, h->pr_events[0], 0));  // the handle's stream ends behind both lanes
  HIPC(hipEventRecord(h->evpool[1], h->stream));
  HIPC(hipEventRecord(h->ev_d2h, h->stream_d2h));
  HIPC(hipStreamSynchronize(h->stream));
  float ms = 0.f;
  HIPC(hipEventElapsedTime(&ms, h->evpool[0], h->evpool[1]));
  h->tm.predict_ms = ms;
  if (sorted_now) {
    HIPC(hipEventElapsedTime(&ms, h->evpool[0], h->ev_sort));
    h->tm.sort_points_ms = ms;
  }
  for (long c = 0; c < nchunks; ++c) {
    HIPC(hipEventElapsedTime(&ms, h->evpool[2 + 4 * c], h->evpool[3 + 4 * c]));
    h->tm.rhs_ms += ms;
    HIPC(hipEventElapsedTime(&ms, h->evpool[4 + 4 * c], h->evpool[5 + 4 * c]));
    h->tm.contract_ms += ms;
  }
  if (sparse) {
    HIPC(hipMemcpy(sp_host.data(), h->sp_stats.p, sizeof(unsigned long long) * sp_host.size(), hipMemcpyDeviceToHost));
    const int ntl = (kend - (nIblk - 1) * 128) / 16;  // K tiles of the (short) last block
    for (long c = 0; c < nchunks; ++c) {
      HIPC(hipEventElapsedTime(&ms, h->evpool[2 + 4 * nchunks + 2 * c], h->evpool[2 + 4 * c]));
      h->tm.sparse_lists_ms += ms;
      HIPC(hipEventElapsedTime(&ms, h->evpool[3 + 4 * c], h->evpool[3 + 4 * nchunks + 2 * c]));
      h->tm.sparse_lists_ms += ms;
      const long nTb = (std::min<long>(chunk, npt - c * chunk) + 127) / 128;
      const double tiles = (double)sp_host[4 * c], offk = (double)sp_host[4 * c + 1];
      h->tm.sparse_tiles += tiles;
      h->tm.sparse_ktiles += offk;
      // executed flops: off-diagonal K tiles are 128 x 16 x 128 products; a diagonal block is nt (nt + 1) / 2 products of 16 rows x 16 k
      // x 128 points (nt = 8, or the short last block's -- every point block has that row block: the last row is the 1 of ok.py:673;
      // gathered groups: k_sp_tiles_g counted the products of the triangular parts, short last tiles included)
      const double diagp = gathered ? (double)sp_host[4 * c + 2] : 36.0 * std::max(0.0, tiles - (double)nTb) + (ntl * (ntl + 1) / 2) * (double)nTb;
      h->tm.sparse_diag_products += diagp;
      h->tm.contract_flops_executed += 2.0 * 128.0 * 16.0 * 128.0 * offk + 2.0 * 16.0 * 16.0 * 128.0 * diagp;
    }
  }
  h->tm.contract_launches = nchunks;
  h->tm.rhs_overlapped = two ? 1 : 0;
  h->have_results = true;
  return MIK_OK;
}
