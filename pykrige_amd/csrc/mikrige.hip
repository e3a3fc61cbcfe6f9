// mikrige.hip -- C ABI (include/mikrige.h), handles, device groups and the factor exchange, set_problem / set_points / set_grid,
// results, statistics.  The other translation units: mik_inverse.hip, mik_predict.hip, mik_mw.hip, mik_mw_chol.hip.
#include "mik_k_core.h"
#include "mik_host.h"

extern "C" {

const char* mik_last_error(void) { return g_err.c_str(); }

int mik_abi_version(void) { return MIK_ABI_VERSION; }

int mik_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

static void destroy_one(mik_handle* h);

static double env_seconds(const char* name, double dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  const double v = atof(e);
  return v > 0.0 ? v : dflt;
}

static int create_one_body(mik_handle* h, int device) {
  h->device = device;
  HIPC(hipSetDevice(device));
  HIPC(hipStreamCreate(&h->stream));
  {
    int lo = 0, hi = 0;  // the look-ahead branch is the critical path: give it the dispatcher's highest priority
    HIPC(hipDeviceGetStreamPriorityRange(&lo, &hi));
    HIPC(hipStreamCreateWithPriority(&h->stream2, hipStreamNonBlocking, hi));
    HIPC(hipStreamCreateWithPriority(&h->stream3, hipStreamNonBlocking, hi));
  }
  HIPC(hipStreamCreateWithFlags(&h->stream_d2h, hipStreamNonBlocking));
  HIPC(hipEventCreateWithFlags(&h->ev_d2h, hipEventDisableTiming));
  HIPC(hipEventCreateWithFlags(&h->ev_chunk, hipEventDisableTiming));
  HIPC(hipEventCreate(&h->ev_sort));
  {
    int ncu = 0;
    if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && ncu > 0) h->n_cu = ncu;
  }
  const char* env = getenv("MIK_FACTOR");
  if (env) h->opt_factor = !strcmp(env, "sweep") ? 1 : (!strcmp(env, "lu") || !strcmp(env, "pivoted")) ? 2 : 0;
  env = getenv("MIK_SYMMETRIC");
  if (env) h->opt_sym = atoi(env) ? 1 : 0;
  env = getenv("MIK_CHUNK");
  if (env && atol(env) >= 128) h->opt_chunk = (atol(env) / 128) * 128;
  env = getenv("MIK_SYMSWEEP");
  if (env) h->opt_symsweep = atoi(env) < 0 ? -1 : atoi(env) ? 1 : 0;
  env = getenv("MIK_SPARSE");
  if (env && atoi(env) >= -1 && atoi(env) <= 2) h->opt_sparse = atoi(env);
  env = getenv("MIK_SORT_POINTS");
  if (env && atoi(env) >= -1 && atoi(env) <= 1) h->opt_sort_points = atoi(env);
  env = getenv("MIK_SPARSE_GROUP");
  if (env && atoi(env) >= 1 && atoi(env) <= 16) h->opt_sparse_group = atoi(env);
  env = getenv("MIK_SPARSE_ROWS");
  if (env && (atoi(env) == -1 || atoi(env) == 16 || atoi(env) == 128)) h->opt_sparse_rows = atoi(env);
  env = getenv("MIK_UPDATE_REV");
  if (env) h->opt_update_rev = atoi(env) < 0 ? -1 : atoi(env) ? 1 : 0;
  env = getenv("MIK_PANEL_STREAM");
  if (env) h->opt_panel_stream = atoi(env) < 0 ? -1 : atoi(env) ? 1 : 0;
  env = getenv("MIK_TRI");
  if (env) h->opt_tri = atoi(env) ? 1 : 0;
  env = getenv("MIK_EXCHANGE");
  if (env) h->opt_exchange = !strcmp(env, "rccl") ? 1 : !strcmp(env, "peer") ? 2 : !strcmp(env, "redundant") ? 3 : 0;
  env = getenv("MIK_EXCHANGE_TRI");
  if (env) h->opt_exchange_tri = atoi(env) ? 1 : 0;
  env = getenv("MIK_ALIAS_DEVICES");
  if (env) h->alias_ok = atoi(env) != 0;
  env = getenv("MIK_RHS_OVERLAP");
  if (env) h->opt_rhs_overlap = atoi(env) ? 1 : 0;
  env = getenv("MIK_ASYNC_EXCHANGE");
  if (env && atoi(env) >= 0 && atoi(env) <= 2) h->opt_async_exchange = atoi(env);
  h->rccl_init_limit = env_seconds("MIK_RCCL_INIT_TIMEOUT", 120.0);
  h->rccl_bcast_limit = env_seconds("MIK_RCCL_BCAST_TIMEOUT", 30.0);
  h->peer_limit = env_seconds("MIK_PEER_TIMEOUT", 30.0);
  return MIK_OK;
}

static int create_one(int device, mik_handle** out) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(MIK_EHIP, "mik_create: no HIP device visible (this library has no CPU path)");
  if (device < 0 || device >= n) return fail(MIK_EINVAL, "mik_create: device index out of range");
  mik_handle* h = new mik_handle();
  const int rc = create_one_body(h, device);
  if (rc != MIK_OK) {  // a HIP call failed half-way: give back what was created
    const std::string keep = g_err;
    destroy_one(h);
    g_err = keep;
    return rc;
  }
  *out = h;
  return MIK_OK;
}

static void destroy_one(mik_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->stream_d2h) (void)hipStreamSynchronize(h->stream_d2h);
  if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
  if (h->xstream) {
    (void)hipStreamSynchronize(h->xstream);
    (void)hipStreamDestroy(h->xstream);
  }
  h->xsum.release();
  DevBuf* bufs[] = {&h->xs, &h->ys, &h->zs, &h->vals, &h->wells, &h->extra_cols, &h->T, &h->cvec, &h->Cold, &h->Cnew,
                    &h->Rt, &h->TKt, &h->Dinv, &h->DinvT, &h->P0, &h->P1, &h->cand0, &h->cand1, &h->pivall, &h->flag,
                    &h->Cold2, &h->Cnew2, &h->Rt2, &h->Dinv2, &h->DinvT2, &h->Dinv3, &h->DinvT3, &h->Dnext, &h->Dcopy, &h->Rb, &h->grid.gx, &h->grid.gy, &h->grid.gz, &h->grid.orig,
                    &h->grid.cstart,
                    &h->px, &h->py, &h->pz, &h->grid_axes, &h->grid_idx, &h->Averify, &h->vbuf, &h->extra_rows, &h->z, &h->ss, &h->Bt, &h->Bt2, &h->part, &h->mw_idx, &h->mw_dist, &h->stat_S, &h->stat_x, &h->stat_out, &h->queue,
                    &h->xs_s, &h->ys_s, &h->zs_s, &h->vals_s, &h->extra_cols_s, &h->sbox, &h->sp_cand, &h->sp_flags, &h->sp_klist, &h->sp_kcount,
                    &h->sp_nrows, &h->sp_rows, &h->sp_rstart, &h->sp_tiles, &h->sp_xoff, &h->sp_stats, &h->sp2_cand, &h->sp2_flags, &h->sp2_klist, &h->sp2_kcount,
                    &h->sp2_nrows, &h->sp2_rows, &h->sp2_rstart, &h->sp2_tiles, &h->sp2_xoff, &h->part2, &h->queue2, &h->dsc, &h->sp_recs, &h->sp2_recs, &h->ps_key[0], &h->ps_key[1], &h->ps_idx[0], &h->ps_idx[1], &h->ps_table, &h->ps_box, &h->ps_x, &h->ps_y, &h->ps_z, &h->ps_zs, &h->ps_sss, &h->xpack};
  for (DevBuf* b : bufs) b->release();
  h->pin_in.release();
  h->pin_out.release();
  for (hipEvent_t e : h->evpool) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->la_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ps_events) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->pr_events) (void)hipEventDestroy(e);
  if (h->ev_chunk) (void)hipEventDestroy(h->ev_chunk);
  if (h->ev_sort) (void)hipEventDestroy(h->ev_sort);
  for (hipEvent_t e : h->xevents) (void)hipEventDestroy(e);
  if (h->ev_d2h) (void)hipEventDestroy(h->ev_d2h);
  if (h->stream_d2h) (void)hipStreamDestroy(h->stream_d2h);
  if (h->stream2) (void)hipStreamDestroy(h->stream2);
  if (h->stream3) (void)hipStreamDestroy(h->stream3);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

// Process-wide default for the number of devices a new handle spans (mik_set_devices, or MIK_NGPU in the environment):
// 1 = the handle's own device only; 0 = every visible device.
static int g_default_devices = -1;  // -1: not set programmatically, read MIK_NGPU
static int default_devices() {
  if (g_default_devices >= 0) return g_default_devices;
  const char* env = getenv("MIK_NGPU");
  if (!env || !*env) return 1;
  if (!strcmp(env, "all")) return 0;
  return std::max(0, atoi(env));
}

static int join_exchange(mik_handle* h);

static void release_group(mik_handle* h) {
  (void)join_exchange(h);  // bounded; an exchange that cannot finish has given up its buffers and streams already
  for (size_t i = 0; i < h->xstreams.size(); ++i) {
    const int dev = i == 0 ? h->device : h->kids[i - 1]->device;
    (void)hipSetDevice(dev);
    for (hipStream_t st : h->xstreams[i])
      if (st) (void)hipStreamDestroy(st);
  }
  h->xstreams.clear();
  for (mik_handle* k : h->kids) destroy_one(k);
  h->kids.clear();
  (void)hipSetDevice(h->device);
}

static int set_group(mik_handle* h, int n) {
  if (h->is_kid) return fail(MIK_EINVAL, "mik_handle_set_devices: not on a member of a device group");
  int visible = 0;
  if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return fail(MIK_EHIP, "no HIP device visible");
  if (n == 0) n = visible;
  if (n < 1 || n > 64) return fail(MIK_EINVAL, "number of devices out of range");
  if (n > visible && !h->alias_ok)
    return fail(MIK_EINVAL, "more devices requested than are visible (option alias_devices / MIK_ALIAS_DEVICES=1 lets several "
                            "logical devices share one GPU -- for tests on a 1-GPU box)");
  if ((int)h->kids.size() + 1 == n) return MIK_OK;
  release_group(h);
  for (int i = 1; i < n; ++i) {
    mik_handle* k = nullptr;
    const int rc = create_one((h->device + i) % visible, &k);
    if (rc != MIK_OK) {
      const std::string keep = g_err;
      release_group(h);
      g_err = keep;
      return rc;
    }
    k->is_kid = true;
    k->opt_factor = h->opt_factor, k->opt_sym = h->opt_sym;
    k->opt_chunk = h->opt_chunk, k->opt_symsweep = h->opt_symsweep, k->opt_panel_stream = h->opt_panel_stream, k->opt_update_rev = h->opt_update_rev, k->opt_lookahead = h->opt_lookahead, k->opt_gate = h->opt_gate, k->opt_pinv_fast = h->opt_pinv_fast, k->opt_rhs_overlap = h->opt_rhs_overlap, k->opt_verify = h->opt_verify, k->verify_tol_z = h->verify_tol_z, k->verify_tol_inv = h->verify_tol_inv;
    k->opt_mw_class = h->opt_mw_class, k->opt_mw_knn_bound = h->opt_mw_knn_bound, k->opt_mw_pivot = h->opt_mw_pivot, k->opt_mw_lds_cap = h->opt_mw_lds_cap, k->opt_tri = h->opt_tri, k->opt_symmetrize = h->opt_symmetrize;
    k->opt_sparse = h->opt_sparse, k->opt_sparse_lanes = h->opt_sparse_lanes, k->opt_sparse_rows = h->opt_sparse_rows, k->opt_sparse_group = h->opt_sparse_group, k->opt_sort_points = h->opt_sort_points, k->opt_drift_eq = h->opt_drift_eq;
    k->opt_pinv_block = h->opt_pinv_block, k->opt_mw_static = h->opt_mw_static, k->opt_mw_knn_lane = h->opt_mw_knn_lane;
    k->custom_fn = h->custom_fn, k->custom_user = h->custom_user;
    h->kids.push_back(k);
  }
  // the group's data changed hands: whatever the leader held is stale for the new members
  h->have_problem = h->have_factor = h->have_points = h->have_results = false;
  h->t_state = 0;
  HIPC(hipSetDevice(h->device));
  return MIK_OK;
}

int mik_create(int device, mik_handle** out) {
  if (!out) return fail(MIK_EINVAL, "mik_create: out is NULL");
  mik_handle* h = nullptr;
  MIKC(create_one(device, &h));
  const int want = default_devices();
  if (want != 1) {
    int visible = 1;
    (void)hipGetDeviceCount(&visible);
    const int rc = set_group(h, want == 0 ? visible : want);
    if (rc != MIK_OK) {
      const std::string keep = g_err;
      destroy_one(h);
      g_err = keep;
      return rc;
    }
  }
  *out = h;
  return MIK_OK;
}

void mik_destroy(mik_handle* h) {
  if (!h) return;
  release_group(h);
  destroy_one(h);
}

int mik_set_devices(int n) {
  if (n < 0 || n > 64) return fail(MIK_EINVAL, "mik_set_devices: n must be 0 (all visible devices) or 1..64");
  g_default_devices = n;
  return MIK_OK;
}

int mik_handle_set_devices(mik_handle* h, int n) {
  if (!h) return fail(MIK_EINVAL, "mik_handle_set_devices: NULL handle");
  return set_group(h, n);
}

int mik_handle_devices(mik_handle* h) { return h ? (int)h->kids.size() + 1 : 0; }

int mik_set_custom_variogram(mik_handle* h, mik_variogram_fn fn, void* user) {
  if (!h) return fail(MIK_EINVAL, "mik_set_custom_variogram: NULL handle");
  h->custom_fn = fn;
  h->custom_user = user;
  for (mik_handle* k : h->kids) k->custom_fn = fn, k->custom_user = user;
  return MIK_OK;
}

int mik_set_option(mik_handle* h, const char* key, double value) {
  if (!h || !key) return fail(MIK_EINVAL, "mik_set_option: NULL argument");
  MIKC(join_exchange(h));
  for (mik_handle* k : h->kids) MIKC(mik_set_option(k, key, value));
  if (!strcmp(key, "async_exchange")) {
    if (value != 0.0 && value != 1.0 && value != 2.0) return fail(MIK_EINVAL, "async_exchange must be 0, 1 or 2");
    h->opt_async_exchange = (int)value;
  } else if (!strcmp(key, "rccl_init_timeout") || !strcmp(key, "rccl_bcast_timeout") || !strcmp(key, "peer_timeout")) {
    if (!(value > 0.0)) return fail(MIK_EINVAL, "a timeout must be a positive number of seconds");
    (key[0] == 'p' ? h->peer_limit : key[5] == 'i' ? h->rccl_init_limit : h->rccl_bcast_limit) = value;
  } else if (!strcmp(key, "exchange")) {
    if (value < 0 || value > 3) return fail(MIK_EINVAL, "exchange must be 0 (auto), 1 (rccl), 2 (peer copies) or 3 (redundant factorisation)");
    h->opt_exchange = (int)value;
  } else if (!strcmp(key, "exchange_tri")) {
    h->opt_exchange_tri = value != 0.0;
  } else if (!strcmp(key, "alias_devices")) {
    h->alias_ok = value != 0.0;
  } else if (!strcmp(key, "factor")) {
    if (value < 0 || value > 2) return fail(MIK_EINVAL, "factor must be 0 (auto), 1 (sweep) or 2 (pivoted)");
    h->opt_factor = (int)value;
  } else if (!strcmp(key, "symmetric")) {
    h->opt_sym = value != 0.0;
  } else if (!strcmp(key, "sparse")) {
    if (value != -1.0 && value != 0.0 && value != 1.0 && value != 2.0) return fail(MIK_EINVAL, "sparse must be -1 (auto), 0, 1 or 2");
    h->opt_sparse = (int)value;
  } else if (!strcmp(key, "sort_points")) {
    if (value != -1.0 && value != 0.0 && value != 1.0) return fail(MIK_EINVAL, "sort_points must be -1 (auto), 0 or 1");
    h->opt_sort_points = (int)value;
  } else if (!strcmp(key, "sparse_group")) {
    if (!(value >= 1.0 && value <= 16.0)) return fail(MIK_EINVAL, "sparse_group must be 1 .. 16");
    h->opt_sparse_group = (int)value;
  } else if (!strcmp(key, "sparse_rows")) {
    if (value != -1.0 && value != 16.0 && value != 128.0) return fail(MIK_EINVAL, "sparse_rows must be -1 (auto), 16 or 128");
    h->opt_sparse_rows = (int)value;
  } else if (!strcmp(key, "sparse_lanes")) {
    if (value != 1.0 && value != 2.0) return fail(MIK_EINVAL, "sparse_lanes must be 1 or 2");
    h->opt_sparse_lanes = (int)value;
  } else if (!strcmp(key, "drift_eq")) {
    h->opt_drift_eq = value != 0.0;
  } else if (!strcmp(key, "tri")) {
    h->opt_tri = value != 0.0;
  } else if (!strcmp(key, "symmetrize")) {
    h->opt_symmetrize = value != 0.0;
  } else if (!strcmp(key, "chunk")) {
    if (value < 128) return fail(MIK_EINVAL, "chunk must be >= 128");
    h->opt_chunk = ((long)value / 128) * 128;
  } else if (!strcmp(key, "symsweep")) {
    h->opt_symsweep = value < 0.0 ? -1 : (value != 0.0);
  } else if (!strcmp(key, "rhs_overlap")) {
    h->opt_rhs_overlap = value != 0.0;
  } else if (!strcmp(key, "verify")) {
    h->opt_verify = value != 0.0;
  } else if (!strcmp(key, "verify_tol_z") || !strcmp(key, "verify_tol_inv")) {
    if (!(value > 0.0)) return fail(MIK_EINVAL, "a tolerance must be positive");
    (key[11] == 'z' ? h->verify_tol_z : h->verify_tol_inv) = value;
  } else if (!strcmp(key, "pinv_fast")) {
    h->opt_pinv_fast = value != 0.0;
  } else if (!strcmp(key, "pinv_block")) {
    h->opt_pinv_block = value < 0.0 ? -1 : (value != 0.0);
  } else if (!strcmp(key, "gate")) {
    h->opt_gate = value < 0.0 ? -1 : (value != 0.0);
  } else if (!strcmp(key, "update_rev")) {
    h->opt_update_rev = value < 0.0 ? -1 : value != 0.0 ? 1 : 0;
  } else if (!strcmp(key, "panel_stream")) {
    h->opt_panel_stream = value < 0.0 ? -1 : (value != 0.0);
  } else if (!strcmp(key, "lookahead")) {
    h->opt_lookahead = value < 0.0 ? -1 : (value != 0.0);
  } else if (!strcmp(key, "mw_knn_bound")) {
    h->opt_mw_knn_bound = value != 0.0;
  } else if (!strcmp(key, "mw_knn_lane")) {
    h->opt_mw_knn_lane = value != 0.0;
  } else if (!strcmp(key, "mw_static")) {
    h->opt_mw_static = value != 0.0;
  } else if (!strcmp(key, "mw_class")) {
    h->opt_mw_class = (int)value;
  } else if (!strcmp(key, "mw_pivot")) {
    h->opt_mw_pivot = value != 0.0;
  } else if (!strcmp(key, "mw_lds_cap")) {
    if (value < 0 || value > 8192) return fail(MIK_EINVAL, "mw_lds_cap must be in 0..8192");
    h->opt_mw_lds_cap = (int)value;
  } else {
    return fail(MIK_EINVAL, std::string("unknown option ") + key);
  }
  return MIK_OK;
}

int64_t mik_matrix_order(mik_handle* h) { return h ? h->M : 0; }

// (hilbert_key: mik_kernels.h -- the device sorts the points of a launch with the same function)
// order[i] = index of the station at position i of the Hilbert-curve order (ties by index: deterministic on every rank / member)
static void hilbert_order(int ndim, long n, const double* xs, const double* ys, const double* zs, std::vector<int>& order) {
  const double* c[3] = {xs, ys, zs};
  double lo[3] = {0, 0, 0}, ext = 0.0;
  for (int d = 0; d < ndim; ++d) {
    double a = 1e300, b = -1e300;
    for (long i = 0; i < n; ++i) {
      a = std::min(a, c[d][i]);
      b = std::max(b, c[d][i]);
    }
    lo[d] = a;
    ext = std::max(ext, b - a);
  }
  const int bits = 16;
  const double scale = (ext > 0.0 && std::isfinite(ext)) ? (double)((1u << bits) - 1) / ext : 0.0;  // one scale: cells are cubes
  std::vector<std::pair<uint64_t, int>> keys((size_t)n);
  for (long i = 0; i < n; ++i) {
    uint32_t X[3] = {0, 0, 0};
    for (int d = 0; d < ndim; ++d) {
      const double q = (c[d][i] - lo[d]) * scale;
      X[d] = (uint32_t)std::min<double>((double)((1u << bits) - 1), std::max(0.0, std::isfinite(q) ? q : 0.0));
    }
    keys[(size_t)i] = {hilbert_key(X, ndim, bits), (int)i};
  }
  std::sort(keys.begin(), keys.end());
  order.resize((size_t)n);
  for (long i = 0; i < n; ++i) order[(size_t)i] = keys[(size_t)i].second;
}

// geographic stations (lon, lat in degrees) as unit vectors: the chord |u_i - u_j| = 2 sin(arc / 2) is monotone in the great-circle
// distance the reference's variogram takes (core.py:36-97), so boxes of the unit vectors bound it
static void geo_unit_vectors(long n, const double* lon, const double* lat, std::vector<double>& u) {
  u.resize(3 * (size_t)n);
  const double rad = 3.14159265358979323846 / 180.0;
  for (long i = 0; i < n; ++i) {
    const double lo = lon[i] * rad, la = lat[i] * rad;
    u[(size_t)i] = std::cos(la) * std::cos(lo);
    u[(size_t)n + i] = std::cos(la) * std::sin(lo);
    u[2 * (size_t)n + i] = std::sin(la);
  }
}
// The order itself is the 2-D curve through (lon, lat) also for geographic problems: a plane curve on a surface keeps 16 consecutive
// stations closer together ON THE SPHERE than a space curve through the unit vectors, which meets the surface in pieces (measured on
// 4096 stations uniform on the sphere, largest box edge of a tile in chord units: median 0.26 / 90 % 0.35 / max 0.51 against
// 0.28 / 0.45 / 1.23) -- lon / lat -> sphere is continuous, so a tile that is compact in lon / lat is compact on the sphere; only the
// BOXES must be boxes of the unit vectors (a lon / lat box is not a distance box at the date line and the poles).
static void station_order(const mik_problem* p, std::vector<int>& order) { hilbert_order(p->ndim, p->n, p->xs, p->ys, p->zs, order); }

int mik_station_order(const mik_problem* p, int32_t* order_out) {
  if (!p || !order_out) return fail(MIK_EINVAL, "mik_station_order: NULL argument");
  if ((p->ndim != 2 && p->ndim != 3) || p->n < 1 || !p->xs || !p->ys || (p->ndim == 3 && !p->zs))
    return fail(MIK_EINVAL, "mik_station_order: station arrays missing");
  std::vector<int> order;
  station_order(p, order);
  for (long i = 0; i < p->n; ++i) order_out[i] = order[(size_t)i];
  return MIK_OK;
}

static int upload_sorted_stations(mik_handle* h, const mik_problem* p) {
  const long N = h->N;
  std::vector<double> unit;  // geographic: the stations' unit vectors (for the boxes below)
  if (!(h->stations_same && (long)h->sort_perm.size() == N)) station_order(p, h->sort_perm);
  if (h->geo) geo_unit_vectors(N, p->xs, p->ys, unit);
  const size_t nb = sizeof(double) * (size_t)N;
  std::vector<double> tmp((size_t)N);
  auto up = [&](DevBuf& dst, const double* src) -> int {
    MIKC(dst.ensure(nb));
    for (long i = 0; i < N; ++i) tmp[(size_t)i] = src[h->sort_perm[(size_t)i]];
    HIPC(hipMemcpyAsync(dst.p, tmp.data(), nb, hipMemcpyHostToDevice, h->stream));
    HIPC(hipStreamSynchronize(h->stream));  // tmp is reused
    return MIK_OK;
  };
  MIKC(up(h->xs_s, p->xs));
  MIKC(up(h->ys_s, p->ys));
  if (h->ndim == 3) MIKC(up(h->zs_s, p->zs));
  MIKC(up(h->vals_s, p->values));
  h->hvals_s = tmp;
  if (h->nextra) {
    MIKC(h->extra_cols_s.ensure(nb * h->nextra));
    for (int c = 0; c < h->nextra; ++c) {
      for (long i = 0; i < N; ++i) tmp[(size_t)i] = p->extra_cols[(size_t)c * N + h->sort_perm[(size_t)i]];
      HIPC(hipMemcpyAsync(h->extra_cols_s.as<double>() + (size_t)c * N, tmp.data(), nb, hipMemcpyHostToDevice, h->stream));
      HIPC(hipStreamSynchronize(h->stream));
    }
  }
  // bounding boxes of the K tiles (16 consecutive stations): lo[3], hi[3] each (tiles without stations: an empty box, never near anything)
  constexpr int gran = 16;
  const int nIblk = h->Mp / gran;
  std::vector<double> box((size_t)nIblk * 6);
  const double* c[3] = {p->xs, p->ys, p->zs};
  if (h->geo) c[0] = unit.data(), c[1] = unit.data() + N, c[2] = unit.data() + 2 * N;
  const int bd = h->geo ? 3 : h->ndim;  // dimension of the boxes
  for (int b = 0; b < nIblk; ++b) {
    double* q = box.data() + (size_t)b * 6;
    for (int d = 0; d < 3; ++d) {
      q[d] = d < bd ? 1e300 : 0.0;
      q[3 + d] = d < bd ? -1e300 : 0.0;
    }
    for (long i = (long)b * gran; i < std::min<long>(N, (long)(b + 1) * gran); ++i)
      for (int d = 0; d < bd; ++d) {
        const double v = c[d][h->sort_perm[(size_t)i]];
        q[d] = std::min(q[d], v);
        q[3 + d] = std::max(q[3 + d], v);
      }
  }
  MIKC(h->sbox.ensure(sizeof(double) * box.size()));
  HIPC(hipMemcpyAsync(h->sbox.p, box.data(), sizeof(double) * box.size(), hipMemcpyHostToDevice, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return MIK_OK;
}


static int one_set_problem(mik_handle* h, const mik_problem* p) {
  if (!h || !p) return fail(MIK_EINVAL, "mik_set_problem: NULL argument");
  if (p->ndim != 2 && p->ndim != 3) return fail(MIK_EINVAL, "ndim must be 2 or 3");
  if (p->n < 1 || p->n > 2000000) return fail(MIK_EINVAL, "n out of range");
  if (p->model_id < 0 || p->model_id > MIK_MODEL_CUSTOM) return fail(MIK_EINVAL, "unknown variogram model id");
  if (!p->xs || !p->ys || !p->values || (p->ndim == 3 && !p->zs)) return fail(MIK_EINVAL, "station arrays missing");
  if (p->n_wells < 0 || p->n_extra < 0 || (p->n_wells > 0 && !p->wells) || (p->n_extra > 0 && !p->extra_cols))
    return fail(MIK_EINVAL, "drift description inconsistent");
  if (p->n_wells > 0 && p->ndim != 2) return fail(MIK_EINVAL, "point_log drift exists only in 2D (uk.py:884-896)");
  if (p->geographic && (p->ndim != 2 || p->regional_linear || p->n_wells || p->n_extra))
    return fail(MIK_EINVAL, "geographic coordinates exist for 2D ordinary kriging only (ok.py:634-640)");
  HIPC(hipSetDevice(h->device));
  h->ndim = p->ndim;
  h->model = p->model_id;
  h->N = (int)p->n;
  h->rl = p->regional_linear ? 1 : 0;
  h->geo = p->geographic ? 1 : 0;
  h->nwells = p->n_wells;
  h->nextra = p->n_extra;
  h->p = (h->rl ? h->ndim : 0) + h->nwells + h->nextra;
  h->M = h->N + h->p + 1;
  h->Mp = ((h->M + 127) / 128) * 128;
  h->exact = p->exact_values ? 1 : 0;
  h->eps = p->eps;
  Vario v{};
  v.model = p->model_id;
  v.p0 = p->params[0];
  v.p1 = p->params[1];
  v.p2 = p->params[2];
  v.c0 = 1.0;
  if (v.model == 2) {
    const double t = v.p1 * 4.0 / 7.0;
    v.c0 = t * t;
  } else if (v.model == 4 || v.model == 5) {
    v.c0 = v.p1 / 3.0;
  }
  v.c0inv = 1.0 / v.c0;
  v.sa = v.sb = 0.0;
  if (v.model == 3) {
    v.sa = 3.0 / (2.0 * v.p1);
    v.sb = 1.0 / (2.0 * (v.p1 * v.p1 * v.p1));
  }
  h->v = v;
  const size_t nb = sizeof(double) * (size_t)h->N;
  MIKC(h->xs.ensure(nb));
  MIKC(h->ys.ensure(nb));
  MIKC(h->vals.ensure(nb));
  {  // the same station coordinates as at the last mik_set_problem (the class sets the problem at every execute()): their Hilbert order
     // is a function of the coordinates alone and is kept
    const size_t cb = sizeof(double) * (size_t)h->N;
    h->stations_same = h->hxs.size() == (size_t)h->N && h->hys.size() == (size_t)h->N && !memcmp(h->hxs.data(), p->xs, cb) &&
                       !memcmp(h->hys.data(), p->ys, cb) &&
                       (p->ndim == 3 ? (h->hzs.size() == (size_t)h->N && !memcmp(h->hzs.data(), p->zs, cb)) : h->hzs.empty());
  }
  h->hvals.assign(p->values, p->values + h->N);
  h->hxs.assign(p->xs, p->xs + h->N);
  h->hys.assign(p->ys, p->ys + h->N);
  if (p->ndim == 3) h->hzs.assign(p->zs, p->zs + h->N);
  else h->hzs.clear();
  h->grid.target = -1;
  HIPC(hipMemcpyAsync(h->xs.p, p->xs, nb, hipMemcpyHostToDevice, h->stream));
  HIPC(hipMemcpyAsync(h->ys.p, p->ys, nb, hipMemcpyHostToDevice, h->stream));
  HIPC(hipMemcpyAsync(h->vals.p, p->values, nb, hipMemcpyHostToDevice, h->stream));
  if (h->ndim == 3) {
    MIKC(h->zs.ensure(nb));
    HIPC(hipMemcpyAsync(h->zs.p, p->zs, nb, hipMemcpyHostToDevice, h->stream));
  }
  if (h->nwells) {
    MIKC(h->wells.ensure(sizeof(double) * 3 * h->nwells));
    HIPC(hipMemcpyAsync(h->wells.p, p->wells, sizeof(double) * 3 * h->nwells, hipMemcpyHostToDevice, h->stream));
  }
  if (h->nextra) {
    MIKC(h->extra_cols.ensure(nb * h->nextra));
    HIPC(hipMemcpyAsync(h->extra_cols.p, p->extra_cols, nb * h->nextra, hipMemcpyHostToDevice, h->stream));
  }
  // shift for the unpivoted sweep: the sill for bounded models, gamma(bounding-box diagonal) otherwise
  {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    const double* c[3] = {p->xs, p->ys, p->zs};
    for (int d = 0; d < h->ndim; ++d)
      for (long i = 0; i < h->N; ++i) {
        lo[d] = std::min(lo[d], c[d][i]);
        hi[d] = std::max(hi[d], c[d][i]);
      }
    double diag2 = 0.0;
    for (int d = 0; d < h->ndim; ++d) diag2 += (hi[d] - lo[d]) * (hi[d] - lo[d]);
    if (h->geo) diag2 = std::min(diag2, 180.0 * 180.0);  // great-circle distances never exceed 180 degrees
    if (v.model >= 2) h->shift_guess = v.p0 + v.p2;
    else h->shift_guess = host_vario(v, std::sqrt(diag2));
    if (!(h->shift_guess > 0.0) || !std::isfinite(h->shift_guess)) h->shift_guess = 1.0;
  }
  h->host_inv = p->a_inv != nullptr;
  if (p->pseudo_inv < 0 || p->pseudo_inv > 2) return fail(MIK_EINVAL, "pseudo_inv must be 0, 1 ('pinv') or 2 ('pinvh')");
  h->pinv = p->pseudo_inv;
  if (h->host_inv) h->host_ainv.assign(p->a_inv, p->a_inv + (size_t)h->M * h->M);
  else h->host_ainv.clear();
  // compact-support model (spherical: gamma constant beyond the range): a second copy of the stations in Hilbert-curve order for
  // the range-aware contraction (k_contract_sp).  Not with a pseudo-inverse (A+ u != e_last) or a caller's inverse (its order is the
  // caller's).  Geographic coordinates (round 5): the curve runs through (lon, lat), the BOXES are boxes of the stations' unit vectors --
  // the chord is monotone in the great-circle distance, so a box test on chords is a superset test on arcs.
  h->sort_ok = h->model == MIK_MODEL_SPHERICAL && !h->pinv && !h->host_inv && h->Mp / 16 <= MIK_SP_MAXK16 &&
               std::isfinite(v.p1) && v.p1 > 0.0 && std::isfinite(v.p0 + v.p2);
  h->factor_sorted = false;
  if (h->sort_ok) MIKC(upload_sorted_stations(h, p));
  else h->sort_perm.clear();  // (stations_same compares with the LAST problem's coordinates: an order built for an earlier set must not survive it)
  // drift equilibration: centre = mean, scale = 1 / max |f - centre| of each drift term over the stations (wells: left alone --
  // their logarithms are O(1..10) already)
  h->drift_eq = h->p > 0 && !h->pinv && !h->host_inv;
  h->factor_eq = false;
  if (h->drift_eq) {
    h->hdsc.assign(2 * (size_t)h->p, 0.0);
    for (int j = 0; j < h->p; ++j) h->hdsc[2 * j + 1] = 1.0;
    auto fit = [&](int j, const double* col) {
      double mean = 0.0;
      for (long i = 0; i < h->N; ++i) mean += col[i];
      mean /= (double)h->N;
      double dev = 0.0;
      for (long i = 0; i < h->N; ++i) dev = std::max(dev, std::fabs(col[i] - mean));
      if (std::isfinite(mean) && std::isfinite(dev) && dev > 0.0) {
        h->hdsc[2 * j] = mean;
        h->hdsc[2 * j + 1] = 1.0 / dev;
      }
    };
    int j = 0;
    if (h->rl) {
      fit(j++, p->xs);
      fit(j++, p->ys);
      if (h->ndim == 3) fit(j++, p->zs);
    }
    j += h->nwells;
    for (int c = 0; c < h->nextra; ++c) fit(j++, p->extra_cols + (size_t)c * h->N);
    MIKC(h->dsc.ensure(sizeof(double) * h->hdsc.size()));
    HIPC(hipMemcpyAsync(h->dsc.p, h->hdsc.data(), sizeof(double) * h->hdsc.size(), hipMemcpyHostToDevice, h->stream));
  }
  HIPC(hipStreamSynchronize(h->stream));
  h->have_problem = true;
  h->have_factor = false;
  h->t_state = 0;
  h->have_results = false;
  return MIK_OK;
}

int mik_set_problem(mik_handle* h, const mik_problem* p) {
  MIKC(join_exchange(h));
  MIKC(one_set_problem(h, p));
  for (mik_handle* k : h->kids) MIKC(one_set_problem(k, p));  // a few hundred KB of station data per device
  return MIK_OK;
}


// custom variogram: bring `rows` rows of a device array (row length ld, `cols` meaningful columns) to the host, let the
// caller's function turn distances into gamma in place, send them back
}  // extern "C"
int custom_roundtrip(mik_handle* h, double* dev, long rows, long cols, long ld) {
  if (!h->custom_fn) return fail(MIK_ESTATE, "variogram model 'custom' needs mik_set_custom_variogram first");
  if (rows <= 0) return MIK_OK;
  std::vector<double> host((size_t)rows * ld);
  HIPC(hipMemcpyAsync(host.data(), dev, sizeof(double) * host.size(), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  h->custom_fn(h->custom_user, host.data(), rows, cols, ld);
  HIPC(hipMemcpyAsync(dev, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return MIK_OK;
}

int launch_assemble(mik_handle* h, double shift, double* dst, bool sorted, bool eq) {
  AsmArgs a{};
  a.T = dst ? dst : h->T.as<double>();
  a.ld = h->Mp;
  a.N = h->N;
  a.p = h->p;
  a.M = h->M;
  a.Mp = h->Mp;
  a.ndim = h->ndim;
  a.xs = sorted ? h->xs_s.as<double>() : h->xs.as<double>();
  a.ys = sorted ? h->ys_s.as<double>() : h->ys.as<double>();
  a.zs = sorted ? h->zs_s.as<double>() : h->zs.as<double>();
  a.v = h->v;
  a.shift = shift;
  a.rl = h->rl;
  a.nwells = h->nwells;
  a.nextra = h->nextra;
  a.wells = h->wells.as<double>();
  a.extra = sorted ? h->extra_cols_s.as<double>() : h->extra_cols.as<double>();
  a.dsc = eq ? h->dsc.as<double>() : nullptr;
  dim3 grid(h->Mp / 64, h->Mp / 64);
  if (h->model == MIK_MODEL_CUSTOM) {
    DISPATCH_NDIM_FIXED(7, h->geo ? 1 : h->ndim, k_assemble, grid, dim3(256), h->stream, a);  // distances
    MIKC(custom_roundtrip(h, a.T, h->N, h->N, a.ld));                                           // d -> gamma on the host
    DISPATCH_NDIM_FIXED(6, h->geo ? 1 : h->ndim, k_assemble, grid, dim3(256), h->stream, a);  // the matrix proper
  } else {
    DISPATCH_MODEL_NDIM(h->model, h->geo ? 1 : h->ndim, k_assemble, grid, dim3(256), h->stream, a);
  }
  HIPC(hipGetLastError());
  return MIK_OK;
}

extern "C" {

int mik_assemble_only(mik_handle* h) {
  if (!h || !h->have_problem) return fail(MIK_ESTATE, "mik_assemble_only: no problem set");
  MIKC(join_exchange(h));
  HIPC(hipSetDevice(h->device));
  MIKC(ensure_factor_buffers(h));
  MIKC(launch_assemble(h, 0.0));
  HIPC(hipStreamSynchronize(h->stream));
  h->have_factor = false;
  h->t_state = 1;
  return MIK_OK;
}



// ---- device groups (mik_set_devices): one host thread per member for the blocking per-device calls ----------------------
static mik_handle* member(mik_handle* h, int i) { return i == 0 ? h : h->kids[i - 1]; }

extern "C++" {
template <class F>
static int for_each_device(mik_handle* h, F fn) {
  const size_t n = h->kids.size() + 1;
  if (n == 1) return fn(0, h);
  std::vector<int> rc(n, MIK_OK);
  std::vector<std::string> err(n);
  std::vector<std::thread> th;
  th.reserve(n - 1);
  for (size_t i = 1; i < n; ++i)
    th.emplace_back([&, i] {
      g_err.clear();
      rc[i] = fn((int)i, h->kids[i - 1]);
      if (rc[i] != MIK_OK) err[i] = g_err;
    });
  rc[0] = fn(0, h);
  if (rc[0] != MIK_OK) err[0] = g_err;
  for (auto& t : th) t.join();
  (void)hipSetDevice(h->device);
  for (size_t i = 0; i < n; ++i)
    if (rc[i] != MIK_OK) return fail(rc[i], "device " + std::to_string(member(h, (int)i)->device) + " (group member " + std::to_string(i) + "): " + err[i]);
  return MIK_OK;
}
}  // extern "C++"

// ---- the factor exchange of a device group: bounded, checked, asynchronous ---------------------------------------------
//
// mik_factor on a group = the leader's K1 + K2 (blocking), then the EXCHANGE of the inverse and of c.  The exchange runs on
// a worker thread over dedicated streams and is joined by the next call that needs the members (mik_predict starts the
// leader's slab first: its prediction overlaps the transfer).  Every wait is BOUNDED:
//   MIK_RCCL_INIT_TIMEOUT  (s, default 120; option "rccl_init_timeout")   ncclCommInitAll of the group's communicators
//   MIK_RCCL_BCAST_TIMEOUT (s, default 30;  option "rccl_bcast_timeout")  the grouped ncclBroadcast until every stream drained
//   MIK_PEER_TIMEOUT       (s, default 30;  option "peer_timeout")        the peer scatter + all-gather
// When a limit expires the worker is abandoned (detached; it owns everything it touches through a shared job record, never
// the handle), the members' matrix buffers and exchange streams are LEAKED on purpose (a late transfer may still write
// them) and replaced, RCCL is marked unusable for the rest of the process, and the exchange continues on the next path of
// exchange = auto:  RCCL broadcast -> peer copies -> every member factors the matrix itself.  A forced path ("exchange"
// 1 / 2) returns the error instead.  After every transfer each member's copy is CHECKSUMMED on its device against the
// leader's (k_checksum: order-independent 2 x 64-bit sums of T and c); a mismatch counts as a failed exchange.
// does the factor exchange move the packed upper block triangle?  Only an inverse that is exactly symmetric by construction: computed on the
// device (not mik_problem.a_inv) with "symmetrize" on -- the half sweep mirrors its triangle whatever that option says, but whether it ran is known
// on the root only, so the rule every rank can evaluate is the option.
static bool exchange_triangle(const mik_handle* h) { return h->opt_exchange_tri && h->Mp > 128 && !h->host_inv && h->opt_symmetrize; }

struct XchgMember {
  int device = 0;
  double* T = nullptr;
  double* X = nullptr;                   // what travels: T itself, or the packed upper block triangle (XchgJob::tri)
  double* cvec = nullptr;
  hipStream_t xs = nullptr;              // the member's exchange stream
  unsigned long long* sum_dev = nullptr; // 4 words on the member's device: checksums of T and of c
  ncclComm_t comm = nullptr;
};
struct XchgJob {
  std::mutex m;
  std::condition_variable cv;
  bool done = false;
  int rc = MIK_OK;
  std::string err;
  std::atomic<int> phase{0};  // 0 = communicator set-up (RCCL only), 1 = transfer
  std::chrono::steady_clock::time_point t_start, t_phase1, t_done;
  int path = 0;               // 1 = RCCL broadcast, 2 = peer copies
  bool dry = false;           // mik_selftest_exchange: no HIP calls (drives the control flow against a stand-in RCCL on CPU)
  size_t Mp = 0;
  size_t xlen = 0;            // doubles of the matrix payload: Mp * Mp, or tri_len(Mp)
  bool tri = false;           // the payload is the packed upper block triangle; members unpack it into T after the checksum
  std::vector<XchgMember> mem;
  std::vector<std::vector<hipStream_t>> xstreams;  // peer path: xstreams[i][k] = stream on device i for the copy to device k
  std::vector<hipEvent_t> xevents;
  std::vector<unsigned long long> sums;            // 4 words per member, host side
  unsigned long long leader_sums[4] = {0, 0, 0, 0};  // computed by mik_factor before the exchange starts
  int rccl_ranks = 0;
};

static std::timed_mutex g_group_mutex;
static std::map<std::vector<int>, std::vector<ncclComm_t>> g_group_comms;
static std::atomic<bool> g_rccl_dead{false};  // a bounded wait on RCCL ran out: not used again in this process
static std::string g_rccl_dead_why;           // (written before the flag is raised)

// communicators of a single-process device group, cached per device list for the life of the process: creating them
// (ncclCommInitAll) costs seconds on an 8-GPU node, and every kriging object has its own handle.  g_group_mutex held.
static int group_comms(const std::vector<int>& devs, std::vector<ncclComm_t>** out) {
  {
    std::vector<int> sorted = devs;
    std::sort(sorted.begin(), sorted.end());
    const char* allow = getenv("MIK_RCCL_ALLOW_ALIAS");  // stand-in libraries of the tests accept duplicate devices
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end() && !(allow && atoi(allow)))
      return fail(MIK_ERCCL, "RCCL needs one distinct GPU per group member (this group aliases a device)");
  }
  auto it = g_group_comms.find(devs);
  if (it == g_group_comms.end()) {
    MIKC(rccl_load());
    if (!g_rccl.CommInitAll || !g_rccl.GroupStart || !g_rccl.GroupEnd) return fail(MIK_ERCCL, "librccl.so lacks ncclCommInitAll / ncclGroupStart / ncclGroupEnd");
    std::vector<ncclComm_t> comms(devs.size(), nullptr);
    NCCLC(g_rccl.CommInitAll(comms.data(), (int)devs.size(), devs.data()));
    it = g_group_comms.emplace(devs, std::move(comms)).first;
  }
  *out = &it->second;
  return MIK_OK;
}

// checksums of every member's T and c on its exchange stream, then the streams are drained and the sums compared
static int xchg_verify(XchgJob* j) {
  const size_t Mp = j->Mp, n = j->mem.size();
  j->sums.assign(4 * n, 0ull);
  for (int w = 0; w < 4; ++w) j->sums[w] = j->leader_sums[w];
  for (size_t i = 1; i < n; ++i) {
    const XchgMember& d = j->mem[i];
    HIPC(hipSetDevice(d.device));
    HIPC(hipMemsetAsync(d.sum_dev, 0, 4 * sizeof(unsigned long long), d.xs));
    hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, d.xs, (const unsigned long long*)d.X, j->xlen, d.sum_dev);
    hipLaunchKernelGGL(k_checksum, dim3(4), dim3(256), 0, d.xs, (const unsigned long long*)d.cvec, Mp, d.sum_dev + 2);
    if (j->tri) {  // (the sums are compared below, before anybody reads T) unpack, then the lower block triangle as the mirror image
      hipLaunchKernelGGL(k_tri_pack, dim3((unsigned)Mp), dim3(256), 0, d.xs, d.T, Mp, d.X, 1);
      MIKC(launch_mirror_upper(d.T, (long)Mp, d.xs));
    }
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(j->sums.data() + 4 * i, d.sum_dev, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, d.xs));
  }
  for (size_t i = 1; i < n; ++i) {
    HIPC(hipSetDevice(j->mem[i].device));
    HIPC(hipStreamSynchronize(j->mem[i].xs));
  }
  if (j->path == 1) {  // the root's part of the broadcast runs on the leader's exchange stream
    HIPC(hipSetDevice(j->mem[0].device));
    HIPC(hipStreamSynchronize(j->mem[0].xs));
  }
  for (size_t i = 1; i < n; ++i)
    for (int w = 0; w < 4; ++w)
      if (j->sums[4 * i + w] != j->sums[w]) {
        char b[200];
        snprintf(b, sizeof b, "exchange checksum mismatch on group member %zu (device %d): the copy of the inverse differs from the leader's",
                 i, j->mem[i].device);
        return fail(MIK_ERCCL, b);
      }
  return MIK_OK;
}

// the north_star's exchange: ONE ncclBroadcast of the inverted matrix (and one of c) from the leader to every member, all
// members' calls fused in a group, each on its own device's exchange stream
static int xchg_rccl(XchgJob* j) {
  // one RCCL exchange at a time; a worker stuck inside RCCL keeps the lock for ever, so waiting for it watches the flag
  std::unique_lock<std::timed_mutex> lock(g_group_mutex, std::defer_lock);
  for (;;) {
    if (g_rccl_dead.load()) return fail(MIK_ERCCL, "RCCL disabled for this process: " + g_rccl_dead_why);
    if (lock.try_lock_for(std::chrono::milliseconds(50))) break;
  }
  if (g_rccl_dead.load()) return fail(MIK_ERCCL, "RCCL disabled for this process: " + g_rccl_dead_why);
  const int n = (int)j->mem.size();
  std::vector<int> devs(n);
  for (int i = 0; i < n; ++i) devs[i] = j->mem[i].device;
  std::vector<ncclComm_t>* comms = nullptr;
  MIKC(group_comms(devs, &comms));
  j->rccl_ranks = (int)comms->size();
  {
    std::lock_guard<std::mutex> lk(j->m);
    j->t_phase1 = std::chrono::steady_clock::now();
    j->phase.store(1);
    j->cv.notify_all();  // the waiter switches from the set-up limit to the transfer limit
  }
  NCCLC(g_rccl.GroupStart());
  ncclResult_t first_bad = ncclSuccess;  // a failing call must not leave RCCL inside an open group
  for (int i = 0; i < n && first_bad == ncclSuccess; ++i) {
    const XchgMember& d = j->mem[i];
    if (!j->dry && hipSetDevice(d.device) != hipSuccess) {
      first_bad = ncclUnhandledCudaError;
      break;
    }
    first_bad = g_rccl.Broadcast(d.X, d.X, j->xlen, ncclDouble, 0, (*comms)[i], d.xs);
    if (first_bad == ncclSuccess) first_bad = g_rccl.Broadcast(d.cvec, d.cvec, j->Mp, ncclDouble, 0, (*comms)[i], d.xs);
  }
  const ncclResult_t end_rc = g_rccl.GroupEnd();
  NCCLC(first_bad);
  NCCLC(end_rc);
  if (j->dry) return MIK_OK;
  return xchg_verify(j);
}

static int copy_between(void* dst, int ddev, const void* src, int sdev, size_t bytes, hipStream_t st) {
  if (bytes == 0) return MIK_OK;
  if (ddev == sdev) HIPC(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st));
  else HIPC(hipMemcpyPeerAsync(dst, ddev, src, sdev, bytes, st));
  return MIK_OK;
}

// The exchange without RCCL, shaped for xGMI's full mesh (every GPU has its own link to every other one): the leader
// SCATTERS the matrix in n-1 pieces, one per member, and every member forwards its piece to the other members
// (ALL-GATHER) as soon as it has arrived.  Every link carries 1/(n-1) of the matrix in each of the two steps, against the
// whole matrix on each of the leader's links for a direct fan-out: 3.5x less time on 8 GPUs.
static int xchg_peer(XchgJob* j) {
  const int n = (int)j->mem.size();
  const size_t Mp = j->Mp, S = j->xlen;
  const size_t pieces = (size_t)(n - 1);
  const size_t per = ((S + pieces - 1) / pieces + 511) / 512 * 512;
  auto piece = [&](int k, size_t* off, size_t* len) {
    *off = std::min(S, (size_t)(k - 1) * per);
    *len = std::min(per, S - *off);
  };
  const XchgMember& d0 = j->mem[0];
  HIPC(hipSetDevice(d0.device));
  for (int k = 1; k < n; ++k) {  // scatter (and c, which is small, to everybody directly)
    const XchgMember& dk = j->mem[k];
    size_t off, len;
    piece(k, &off, &len);
    hipStream_t st = j->xstreams[0][k];
    MIKC(copy_between(dk.X + off, dk.device, d0.X + off, d0.device, sizeof(double) * len, st));
    MIKC(copy_between(dk.cvec, dk.device, d0.cvec, d0.device, sizeof(double) * Mp, st));
    HIPC(hipEventRecord(j->xevents[k], st));
  }
  for (int q = 1; q < n; ++q) {  // all-gather among the members
    const XchgMember& dq = j->mem[q];
    size_t off, len;
    piece(q, &off, &len);
    HIPC(hipSetDevice(dq.device));
    for (int k = 1; k < n; ++k) {
      if (k == q) continue;
      const XchgMember& dk = j->mem[k];
      hipStream_t st = j->xstreams[q][k];
      HIPC(hipStreamWaitEvent(st, j->xevents[q], 0));
      MIKC(copy_between(dk.X + off, dk.device, dq.X + off, dq.device, sizeof(double) * len, st));
    }
  }
  for (int i = 0; i < n; ++i) {
    HIPC(hipSetDevice(j->mem[i].device));
    for (int k = 0; k < n; ++k)
      if (j->xstreams[i][k]) HIPC(hipStreamSynchronize(j->xstreams[i][k]));
  }
  return xchg_verify(j);
}

static void xchg_worker(std::shared_ptr<XchgJob> j) {
  g_err.clear();
  const int rc = j->path == 1 ? xchg_rccl(j.get()) : xchg_peer(j.get());
  std::lock_guard<std::mutex> lk(j->m);
  j->rc = rc;
  j->err = g_err;
  j->t_done = std::chrono::steady_clock::now();
  j->done = true;
  j->cv.notify_all();
}

// wait for the worker within the limits; returns true if it finished (its status is then in j->rc / j->err)
static bool xchg_wait(XchgJob* j, double init_limit, double xfer_limit) {
  using clock = std::chrono::steady_clock;
  std::unique_lock<std::mutex> lk(j->m);
  while (!j->done) {
    const bool in_xfer = j->phase.load() >= 1;
    const auto deadline = in_xfer ? j->t_phase1 + std::chrono::duration_cast<clock::duration>(std::chrono::duration<double>(xfer_limit))
                                  : j->t_start + std::chrono::duration_cast<clock::duration>(std::chrono::duration<double>(init_limit));
    if (j->cv.wait_until(lk, deadline) == std::cv_status::timeout && !j->done) {
      const bool now_xfer = j->phase.load() >= 1;
      if (now_xfer == in_xfer) return false;  // same phase, limit expired
    }
  }
  return true;
}

static int ensure_exchange_streams(mik_handle* h, bool peer) {
  const int n = (int)h->kids.size() + 1;
  for (int i = 0; i < n; ++i) {
    mik_handle* d = member(h, i);
    HIPC(hipSetDevice(d->device));
    if (!d->xstream) HIPC(hipStreamCreateWithFlags(&d->xstream, hipStreamNonBlocking));
    MIKC(d->xsum.ensure(4 * sizeof(unsigned long long)));
  }
  if (peer && h->xstreams.size() != (size_t)n) {
    h->xstreams.assign(n, std::vector<hipStream_t>(n, nullptr));
    for (int i = 0; i < n; ++i) {
      HIPC(hipSetDevice(member(h, i)->device));
      for (int k = 0; k < n; ++k) {
        if (k == i) continue;
        HIPC(hipStreamCreateWithFlags(&h->xstreams[i][k], hipStreamNonBlocking));
        const int di = member(h, i)->device, dk = member(h, k)->device;
        int can = 0;
        if (di != dk && hipDeviceCanAccessPeer(&can, di, dk) == hipSuccess && can) {
          (void)hipDeviceEnablePeerAccess(dk, 0);  // "already enabled" is fine; without peer access the copies are staged
          (void)hipGetLastError();
        }
      }
    }
    HIPC(hipSetDevice(h->device));  // the arrival events are recorded on the LEADER's streams: they belong to its device
    while (h->xevents.size() < (size_t)n) {
      hipEvent_t e;
      HIPC(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      h->xevents.push_back(e);
    }
  }
  HIPC(hipSetDevice(h->device));
  return MIK_OK;
}

// launch the exchange on `path` (1 = RCCL, 2 = peer copies) on a worker thread; the job is joined by join_exchange
static int start_exchange(mik_handle* h, int path) {
  const int n = (int)h->kids.size() + 1;
  for (int i = 1; i < n; ++i) {
    HIPC(hipSetDevice(member(h, i)->device));
    MIKC(ensure_factor_buffers(member(h, i)));
  }
  MIKC(ensure_exchange_streams(h, path == 2));
  auto j = std::make_shared<XchgJob>();
  j->path = path;
  j->Mp = (size_t)h->Mp;
  // an inverse the device computed is EXACTLY symmetric (the half sweep mirrors its triangle; the full sweep, the pivoted elimination and the
  // pseudo-inverses end in k_symmetrize): its upper block triangle travels, the members mirror it.  Not a caller's inverse (used as handed
  // over, never symmetrized) nor "symmetrize" 0.
  j->tri = exchange_triangle(h);
  j->xlen = j->tri ? tri_len(j->Mp) : j->Mp * j->Mp;
  j->mem.resize(n);
  for (int i = 0; i < n; ++i) {
    mik_handle* d = member(h, i);
    if (j->tri) {
      HIPC(hipSetDevice(d->device));
      MIKC(d->xpack.ensure(sizeof(double) * j->xlen));
    }
    j->mem[i].device = d->device;
    j->mem[i].T = d->T.as<double>();
    j->mem[i].X = j->tri ? d->xpack.as<double>() : d->T.as<double>();
    j->mem[i].cvec = d->cvec.as<double>();
    j->mem[i].xs = d->xstream;
    j->mem[i].sum_dev = d->xsum.as<unsigned long long>();
  }
  if (path == 2) {
    j->xstreams = h->xstreams;
    j->xevents = h->xevents;
    j->phase.store(1);
  }
  {
    // the leader's checksums NOW, on its compute stream (0.1 ms): once its prediction runs, a small kernel on another stream
    // of that device would queue behind a persistent contraction launch (47 - 480 ms) and hold the whole exchange up
    unsigned long long* sd = h->xsum.as<unsigned long long>();
    HIPC(hipSetDevice(h->device));
    HIPC(hipMemsetAsync(sd, 0, 4 * sizeof(unsigned long long), h->stream));
    if (j->tri) hipLaunchKernelGGL(k_tri_pack, dim3((unsigned)j->Mp), dim3(256), 0, h->stream, h->T.as<double>(), j->Mp, j->mem[0].X, 0);
    hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, h->stream, (const unsigned long long*)j->mem[0].X, j->xlen, sd);
    hipLaunchKernelGGL(k_checksum, dim3(4), dim3(256), 0, h->stream, (const unsigned long long*)h->cvec.p, j->Mp, sd + 2);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(j->leader_sums, sd, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost, h->stream));
    HIPC(hipStreamSynchronize(h->stream));
  }
  j->t_start = j->t_phase1 = std::chrono::steady_clock::now();
  h->xjob = j;
  std::thread(xchg_worker, j).detach();
  return MIK_OK;
}

// a bounded wait ran out: nothing the abandoned worker may still touch is reused or freed
static void abandon_exchange(mik_handle* h, XchgJob* j) {
  for (size_t i = 0; i < h->kids.size() + 1; ++i) {
    mik_handle* d = member(h, (int)i);
    d->xstream = nullptr;  // leaked with whatever is stuck on it
    if (i > 0) {      // the leader's matrix is only read by a transfer
      d->T.leak();
      d->cvec.leak();
      d->xpack.leak();
    }
    d->xsum.leak();
  }
  if (j->path == 2) {
    h->xstreams.clear();  // leaked
    h->xevents.clear();
  } else {
    g_rccl_dead_why = "a bounded wait on " + std::string(j->phase.load() >= 1 ? "the grouped ncclBroadcast" : "ncclCommInitAll") + " ran out";
    g_rccl_dead.store(true);
    // the communicator cache may be locked by the stuck worker: try, do not wait (a dead RCCL is never looked up again)
    if (g_group_mutex.try_lock()) {
      std::vector<int> devs;
      for (const XchgMember& m : j->mem) devs.push_back(m.device);
      g_group_comms.erase(devs);  // the communicators themselves are leaked
      g_group_mutex.unlock();
    }
  }
}

static void mark_kids_factored(mik_handle* h) {
  for (mik_handle* k : h->kids) {
    k->have_factor = true;
    k->t_state = 2;
    k->have_results = false;
    k->factor_sorted = h->factor_sorted;  // (the member sorted its own copy of the stations the same way in mik_set_problem)
    k->factor_eq = h->factor_eq;
    k->tm.factor_path = h->tm.factor_path;
    k->tm.assemble_ms = k->tm.invert_ms = 0.0;
  }
}

// Finish the exchange mik_factor started (no-op when none is in flight): wait within the limits, fall through the paths
// of exchange = auto on failure, leave every member with a verified copy of the inverse or return the error.
static int join_exchange(mik_handle* h) {
  if (!h || !h->xjob) return MIK_OK;
  using clock = std::chrono::steady_clock;
  const auto t_join = clock::now();
  int path = h->xjob->path;
  for (;;) {
    std::shared_ptr<XchgJob> j = h->xjob;
    h->xjob.reset();
    const bool finished = xchg_wait(j.get(), h->rccl_init_limit, path == 1 ? h->rccl_bcast_limit : h->peer_limit);
    int rc;
    std::string why;
    if (finished) {
      rc = j->rc;
      why = j->err;
    } else {
      char b[160];
      snprintf(b, sizeof b, "%s did not finish within %.1f s", path == 1 ? (j->phase.load() >= 1 ? "the grouped ncclBroadcast" : "ncclCommInitAll") : "the peer scatter + all-gather",
               path == 2 ? h->peer_limit : (j->phase.load() >= 1 ? h->rccl_bcast_limit : h->rccl_init_limit));
      why = b;
      rc = path == 1 ? MIK_ERCCL : MIK_EHIP;
      abandon_exchange(h, j.get());
    }
    if (path == 1) h->rccl_failures = rc == MIK_OK ? 0 : h->rccl_failures + 1;
    if (rc == MIK_OK) {
      h->exchange_used = path;
      h->rccl_ranks = j->rccl_ranks;
      h->exchange_ms = std::chrono::duration<double, std::milli>(j->t_done - h->xchg_t0).count();
      h->exchange_bytes = sizeof(double) * (double)(j->xlen + j->Mp);
      mark_kids_factored(h);
      break;
    }
    ++h->exchange_fallbacks;
    const char* names[] = {"", "rccl broadcast", "peer copies"};
    h->exchange_note += std::string(h->exchange_note.empty() ? "" : "; ") + names[path] + " failed (" + why + ")";
    if (h->opt_exchange == path) {  // the caller forced this path
      (void)hipSetDevice(h->device);
      return fail(path == 1 ? MIK_ERCCL : MIK_EHIP, "factor exchange: " + h->exchange_note);
    }
    if (path == 1) {
      const int src = start_exchange(h, 2);
      if (src == MIK_OK) {
        path = 2;
        continue;
      }
      h->exchange_note += "; peer copies could not be started (" + g_err + ")";
    }
    // last resort of exchange = auto: every member assembles and inverts the (identical) matrix itself
    h->exchange_note += "; every member factors the matrix itself";
    int frc = MIK_OK;
    std::string ferr;
    {
      std::vector<std::thread> th;
      std::vector<int> rcs(h->kids.size(), MIK_OK);
      std::vector<std::string> errs(h->kids.size());
      for (size_t i = 0; i < h->kids.size(); ++i)
        th.emplace_back([&, i] {
          g_err.clear();
          rcs[i] = one_factor(h->kids[i]);
          errs[i] = g_err;
        });
      for (auto& t : th) t.join();
      for (size_t i = 0; i < rcs.size(); ++i)
        if (rcs[i] != MIK_OK && frc == MIK_OK) frc = rcs[i], ferr = errs[i];
    }
    (void)hipSetDevice(h->device);
    if (frc != MIK_OK) return fail(frc, "factor exchange: " + h->exchange_note + "; redundant factorisation failed: " + ferr);
    h->exchange_used = 3;
    h->exchange_ms = std::chrono::duration<double, std::milli>(clock::now() - h->xchg_t0).count();
    break;
  }
  h->exchange_wait_ms = std::chrono::duration<double, std::milli>(clock::now() - t_join).count();
  (void)hipSetDevice(h->device);
  return MIK_OK;
}

int mik_factor(mik_handle* h) {
  if (!h) return fail(MIK_ESTATE, "mik_factor: NULL handle");
  MIKC(join_exchange(h));  // an exchange nobody waited for yet still reads the leader's matrix
  h->exchange_used = 0;
  h->exchange_ms = h->exchange_wait_ms = h->exchange_bytes = 0.0;
  h->exchange_fallbacks = 0;
  h->rccl_ranks = 0;
  h->exchange_note.clear();
  if (h->kids.empty()) return one_factor(h);
  for (mik_handle* k : h->kids) k->have_factor = false;
  if (h->opt_exchange == 3) {  // no exchange at all: every member assembles and inverts the (identical) matrix itself
    h->exchange_used = 3;
    return for_each_device(h, [](int, mik_handle* d) { return one_factor(d); });
  }
  MIKC(one_factor(h));
  h->xchg_t0 = std::chrono::steady_clock::now();
  int path = h->opt_exchange == 2 ? 2 : 1;
  if (path == 1 && h->opt_exchange == 0 && g_rccl_dead.load()) {
    h->exchange_note = "rccl disabled for this process (" + g_rccl_dead_why + ")";
    ++h->exchange_fallbacks;
    path = 2;
  } else if (path == 1 && h->opt_exchange == 0 && h->rccl_failures >= 2) {
    // RCCL returned an error twice in a row on this handle (it does every time on aliased devices, and a broken set-up may take
    // seconds to say so): not tried again on it -- peer copies directly
    h->exchange_note = "rccl failed on the last two exchanges of this handle: not tried again";
    ++h->exchange_fallbacks;
    path = 2;
  }
  MIKC(start_exchange(h, path));
  // a forced path reports its failure HERE; auto cannot fail short of every path failing and may finish behind the caller's back
  if (!h->opt_async_exchange || h->opt_exchange != 0) return join_exchange(h);
  return MIK_OK;
}

// CPU-runnable check of the bounded exchange (no HIP call is made): `members` stand-in members run the RCCL path of the
// exchange against whatever librccl the process loads (MIK_RCCL_LIB: a stand-in whose calls hang / fail / succeed) under
// the given limits.  Returns MIK_OK when the exchange finished, MIK_ERCCL when it failed or a limit expired; report gets
// one line: "ok ranks=N" | "failed: ..." | "timeout phase=init|bcast after S s; rccl_dead=1".
int mik_selftest_exchange(int members, double init_limit_s, double bcast_limit_s, char* report, int report_len) {
  if (members < 2 || members > 64 || !report || report_len < 8) return fail(MIK_EINVAL, "mik_selftest_exchange: bad argument");
  auto j = std::make_shared<XchgJob>();
  j->path = 1;
  j->dry = true;
  j->Mp = 128;
  j->mem.resize(members);
  for (int i = 0; i < members; ++i) j->mem[i].device = i;
  j->t_start = j->t_phase1 = std::chrono::steady_clock::now();
  std::thread(xchg_worker, j).detach();
  const bool finished = xchg_wait(j.get(), init_limit_s, bcast_limit_s);
  const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - j->t_start).count();
  if (!finished) {
    const bool bc = j->phase.load() >= 1;
    g_rccl_dead_why = std::string("a bounded wait on ") + (bc ? "the grouped ncclBroadcast" : "ncclCommInitAll") + " ran out";
    g_rccl_dead.store(true);
    snprintf(report, report_len, "timeout phase=%s after %.2f s; rccl_dead=1", bc ? "bcast" : "init", waited);
    return fail(MIK_ERCCL, report);
  }
  if (j->rc != MIK_OK) {
    snprintf(report, report_len, "failed: %s", j->err.c_str());
    return fail(MIK_ERCCL, report);
  }
  snprintf(report, report_len, "ok ranks=%d", j->rccl_ranks);
  return MIK_OK;
}

const char* mik_exchange_note(mik_handle* h) {
  static thread_local std::string note;
  note = h ? h->exchange_note : std::string();
  return note.c_str();
}

int mik_get_matrix(mik_handle* h, int which, double* out) {
  if (!h || !out) return fail(MIK_EINVAL, "mik_get_matrix: NULL argument");
  if (!h->T.p) return fail(MIK_ESTATE, "mik_get_matrix: nothing assembled");
  if (which == 1 && !h->have_factor) return fail(MIK_ESTATE, "mik_get_matrix: not factored");
  MIKC(join_exchange(h));
  HIPC(hipSetDevice(h->device));
  if (which == 1 && (h->factor_sorted || h->factor_eq)) {
    // the factor is in Hilbert-curve station order and / or of the matrix with equilibrated drift rows A' = S A S^T: hand out
    // A^-1 = S^T A'^-1 S in the caller's station order.  S = I except S[N + j][N + j] = s_j, S[N + j][M - 1] = -s_j c_j.
    const long M = h->M, N = h->N;
    std::vector<double> tmp((size_t)M * M);
    HIPC(hipMemcpy2D(tmp.data(), sizeof(double) * M, h->T.p, sizeof(double) * h->Mp, sizeof(double) * M, M, hipMemcpyDeviceToHost));
    if (h->factor_eq) {
      for (long i = 0; i < M; ++i) {  // R = T S: column N + j scaled, the last column takes the centres
        double* r = tmp.data() + (size_t)i * M;
        double add = 0.0;
        for (int j = 0; j < h->p; ++j) {
          add -= r[N + j] * h->hdsc[2 * j + 1] * h->hdsc[2 * j];
          r[N + j] *= h->hdsc[2 * j + 1];
        }
        r[M - 1] += add;
      }
      double* last = tmp.data() + (size_t)(M - 1) * M;  // S^T R: the same on the rows
      for (int j = 0; j < h->p; ++j) {
        double* r = tmp.data() + (size_t)(N + j) * M;
        for (long k = 0; k < M; ++k) {
          last[k] -= r[k] * h->hdsc[2 * j + 1] * h->hdsc[2 * j];
        }
      }
      for (int j = 0; j < h->p; ++j) {
        double* r = tmp.data() + (size_t)(N + j) * M;
        for (long k = 0; k < M; ++k) r[k] *= h->hdsc[2 * j + 1];
      }
    }
    auto orig = [&](long i) { return (h->factor_sorted && i < N) ? (long)h->sort_perm[(size_t)i] : i; };
    for (long i = 0; i < M; ++i) {
      double* dst = out + orig(i) * M;
      const double* src = tmp.data() + (size_t)i * M;
      for (long j = 0; j < M; ++j) dst[orig(j)] = src[j];
    }
    return MIK_OK;
  }
  HIPC(hipMemcpy2D(out, sizeof(double) * h->M, h->T.p, sizeof(double) * h->Mp, sizeof(double) * h->M, h->M,
                   hipMemcpyDeviceToHost));
  return MIK_OK;
}

// Upload points [lo, lo + n) of the sequence of unmasked points to ONE device.  idx == nullptr: that sequence is the caller's
// arrays themselves; otherwise idx[i] is the position of the i-th unmasked point in them (np.nonzero(~mask), ok.py:700).
// The coordinates are staged through page-locked memory: the host copy (or mask compaction) of one coordinate overlaps the
// DMA of the previous one.
static int one_set_points(mik_handle* h, const mik_points* g, const long* idx, long lo, long n) {
  HIPC(hipSetDevice(h->device));
  HIPC(hipStreamSynchronize(h->stream_d2h));  // result copies of an earlier predict still read z / ss
  h->npt = n;
  h->out_off = lo;
  if (idx) h->scatter.assign(idx + lo, idx + lo + n);
  else h->scatter.clear();
  h->scatter32 = nullptr;
  const long cap = std::max<long>(n, 1);
  const int rows = h->ndim + h->nextra;
  MIKC(h->pin_in.ensure(sizeof(double) * (size_t)cap * rows));
  const double* src[3] = {g->px, g->py, g->pz};
  DevBuf* dst[3] = {&h->px, &h->py, &h->pz};
  if (h->nextra) MIKC(h->extra_rows.ensure(sizeof(double) * (size_t)cap * h->nextra));
  for (int r = 0; r < rows; ++r) {
    const double* from = r < h->ndim ? src[r] : g->extra_rows + (size_t)(r - h->ndim) * g->npt;
    double* stage = h->pin_in.as<double>() + (size_t)r * cap;
    double* to;
    if (r < h->ndim) {
      MIKC(dst[r]->ensure(sizeof(double) * cap));
      to = dst[r]->as<double>();
    } else {
      to = h->extra_rows.as<double>() + (size_t)(r - h->ndim) * n;
    }
    if (n == 0) continue;
    if (idx) {
      const long* ix = idx + lo;
      parallel_chunks(n, [&](int, long b, long e) {
        for (long i = b; i < e; ++i) stage[i] = from[ix[i]];
      });
    } else {
      host_copy(stage, from + lo, sizeof(double) * n);
    }
    HIPC(hipMemcpyAsync(to, stage, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
  }
  MIKC(h->z.ensure(sizeof(double) * cap));
  MIKC(h->ss.ensure(sizeof(double) * cap));
  MIKC(h->pin_out.ensure(sizeof(double) * 2 * (size_t)cap));
  HIPC(hipStreamSynchronize(h->stream));
  {  // median step between consecutive points, from <= 1024 sampled pairs (see pts_step); the sample's extent
    h->pts_step = h->pts_extent = -1.0;
    if (n >= 2) {
      const long ns = std::min<long>(1024, n - 1);
      std::vector<double> st((size_t)ns);
      double elo[3] = {1e300, 1e300, 1e300}, ehi[3] = {-1e300, -1e300, -1e300};
      for (long q = 0; q < ns; ++q) {
        const long i = (long)((double)q * (double)(n - 1) / (double)ns);
        const long a0 = idx ? idx[lo + i] : lo + i, a1 = idx ? idx[lo + i + 1] : lo + i + 1;
        double m = 0.0;
        for (int d = 0; d < h->ndim; ++d) {
          m = std::max(m, std::fabs(src[d][a1] - src[d][a0]));
          if (std::isfinite(src[d][a0])) elo[d] = std::min(elo[d], src[d][a0]), ehi[d] = std::max(ehi[d], src[d][a0]);
        }
        st[(size_t)q] = std::isfinite(m) ? m : 1e300;
      }
      for (int d = 0; d < h->ndim; ++d) h->pts_extent = std::max(h->pts_extent, ehi[d] - elo[d]);
      std::nth_element(st.begin(), st.begin() + ns / 2, st.end());
      h->pts_step = st[(size_t)(ns / 2)];
    }
  }
  h->have_points = true;
  h->points_from_grid = false;
  h->points_adjusted = false;
  h->ps_valid = false;
  h->have_results = false;
  return MIK_OK;
}

// contiguous slabs of the n unmasked points, one per group member, cut at multiples of 128 points (the contraction's tile)
static void slab_of(long n, int members, int i, long* lo, long* cnt) {
  auto cut = [&](int k) {
    if (k >= members) return n;
    const long c = (long)((double)n * k / members);
    return std::min(n, (c / 128) * 128);
  };
  *lo = cut(i);
  *cnt = cut(i + 1) - *lo;
}

int mik_slab_of(int64_t n, int members, int i, int64_t* lo, int64_t* count) {
  if (n < 0 || members < 1 || i < 0 || i >= members || !lo || !count) return fail(MIK_EINVAL, "mik_slab_of: bad argument");
  long l, c;
  slab_of((long)n, members, i, &l, &c);
  *lo = l;
  *count = c;
  return MIK_OK;
}

int mik_set_points(mik_handle* h, const mik_points* g) {
  if (!h || !g) return fail(MIK_EINVAL, "mik_set_points: NULL argument");
  if (!h->have_problem) return fail(MIK_ESTATE, "mik_set_points: set the problem first");
  if (g->npt < 0) return fail(MIK_EINVAL, "npt < 0");
  if (g->npt > 0 && (!g->px || !g->py || (h->ndim == 3 && !g->pz))) return fail(MIK_EINVAL, "point arrays missing");
  if (h->nextra > 0 && g->npt > 0 && !g->extra_rows) return fail(MIK_EINVAL, "extra_rows missing for host-evaluated drifts");
  h->npt_total = g->npt;  // (an exchange in flight touches neither the points nor their buffers: not joined here)
  std::vector<long> idx;
  long n = g->npt;
  h->masked = false;
  if (g->mask) {
    unmasked_positions(g->mask, g->npt, idx);
    n = (long)idx.size();
    h->masked = n != g->npt;
  }
  const long* ip = h->masked ? idx.data() : nullptr;
  const int members = (int)h->kids.size() + 1;
  return for_each_device(h, [&](int i, mik_handle* d) {
    long lo, cnt;
    slab_of(n, members, i, &lo, &cnt);
    return one_set_points(d, g, ip, lo, cnt);
  });
}

// The slab [lo, lo + n) of the unmasked cells of a grid given by its axes, generated on ONE device (k_grid_points).  idx as
// in one_set_points.  H2D: the axes, the slab's compacted cell numbers (4 bytes per point, masked style only) and the
// host-evaluated drift rows if the problem has any.
// `idx`: the leader's page-locked list of the unmasked cells (nullptr = no mask); the leader's device copy of it is already in
// its grid_idx (compact_mask), the other members upload their slab
static int one_set_grid(mik_handle* h, bool leader, const mik_grid* g, const unsigned* idx, long lo, long n, long first, long ncells) {
  HIPC(hipSetDevice(h->device));
  HIPC(hipStreamSynchronize(h->stream_d2h));  // result copies of an earlier predict still read z / ss
  h->npt = n;
  h->out_off = lo;
  h->scatter.clear();
  h->scatter32 = idx ? idx + lo : nullptr;
  const long cap = std::max<long>(n, 1);
  const int three = g->ndim == 3;
  const long nax = g->nx + g->ny + (three ? g->nz : 0);
  MIKC(h->grid_axes.ensure(sizeof(double) * (size_t)nax));
  double* ax = h->grid_axes.as<double>();
  HIPC(hipMemcpyAsync(ax, g->gx, sizeof(double) * g->nx, hipMemcpyHostToDevice, h->stream));
  HIPC(hipMemcpyAsync(ax + g->nx, g->gy, sizeof(double) * g->ny, hipMemcpyHostToDevice, h->stream));
  if (three) HIPC(hipMemcpyAsync(ax + g->nx + g->ny, g->gz, sizeof(double) * g->nz, hipMemcpyHostToDevice, h->stream));
  MIKC(h->px.ensure(sizeof(double) * cap));
  MIKC(h->py.ensure(sizeof(double) * cap));
  if (three) MIKC(h->pz.ensure(sizeof(double) * cap));
  if (h->nextra) MIKC(h->pin_in.ensure(sizeof(double) * (size_t)cap * (size_t)h->nextra));
  if (idx && n > 0 && !leader) {
    MIKC(h->grid_idx.ensure(sizeof(unsigned) * (size_t)cap));
    HIPC(hipMemcpyAsync(h->grid_idx.p, idx + lo, sizeof(unsigned) * (size_t)n, hipMemcpyHostToDevice, h->stream));
  }
  if (h->nextra) {
    MIKC(h->extra_rows.ensure(sizeof(double) * (size_t)cap * h->nextra));
    for (int r = 0; r < h->nextra && n > 0; ++r) {
      const double* from = g->extra_rows + (size_t)r * ncells;
      double* stage = h->pin_in.as<double>() + (size_t)r * cap;
      if (idx) {
        const unsigned* ix = idx + lo;
        parallel_chunks(n, [&](int, long b, long e) {
          for (long i = b; i < e; ++i) stage[i] = from[ix[i]];
        });
      } else {
        host_copy(stage, from + lo, sizeof(double) * n);
      }
      HIPC(hipMemcpyAsync(h->extra_rows.as<double>() + (size_t)r * n, stage, sizeof(double) * n, hipMemcpyHostToDevice, h->stream));
    }
  }
  if (n > 0) {
    GridArgs a{};
    a.gx = ax;
    a.gy = ax + g->nx;
    a.gz = three ? ax + g->nx + g->ny : nullptr;
    a.nx = g->nx, a.ny = g->ny, a.nz = three ? g->nz : 1;
    a.cell0 = idx ? first : first + lo, a.n = n;
    a.idx = idx ? h->grid_idx.as<unsigned>() + (leader ? lo : 0) : nullptr;
    a.ndim = g->ndim, a.adjust = g->adjust ? 1 : 0;
    for (int i = 0; i < 3; ++i) a.c[i] = g->center[i], a.st[i] = g->stretch[i];
    for (int i = 0; i < 9; ++i) a.rot[i] = g->rot[i];
    a.px = h->px.as<double>(), a.py = h->py.as<double>(), a.pz = three ? h->pz.as<double>() : nullptr;
    hipLaunchKernelGGL(k_grid_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, a);
    HIPC(hipGetLastError());
  }
  MIKC(h->z.ensure(sizeof(double) * cap));
  MIKC(h->ss.ensure(sizeof(double) * cap));
  MIKC(h->pin_out.ensure(sizeof(double) * 2 * (size_t)cap));
  HIPC(hipStreamSynchronize(h->stream));
  // consecutive points of a grid are one x step apart (meshgrid order; compacted cells of a masked grid mostly so)
  h->pts_step = g->nx > 1 ? std::fabs(g->gx[g->nx / 2] - g->gx[g->nx / 2 - 1]) : (g->ny > 1 ? std::fabs(g->gy[g->ny / 2] - g->gy[g->ny / 2 - 1]) : 0.0);
  h->pts_extent = std::max(g->nx > 1 ? std::fabs(g->gx[g->nx - 1] - g->gx[0]) : 0.0, g->ny > 1 ? std::fabs(g->gy[g->ny - 1] - g->gy[0]) : 0.0);
  if (g->ndim == 3 && g->nz > 1) h->pts_extent = std::max(h->pts_extent, std::fabs(g->gz[g->nz - 1] - g->gz[0]));
  h->have_points = true;
  h->points_from_grid = true;
  h->ps_valid = false;
  h->have_results = false;
  return MIK_OK;
}

// the unmasked cells of mik_set_grid's range, ascending, as 32-bit offsets from its first cell: in the leader's grid_idx (device)
// and scatter_pin (host, for the other members' slabs, the gathers of host-evaluated drift rows and the scatter of the results)
static int compact_mask(mik_handle* h, const int8_t* mask, long ncells, long* n_out) {
  HIPC(hipSetDevice(h->device));
  HIPC(hipStreamSynchronize(h->stream_d2h));
  const long nblk = (ncells + MIK_MASK_CELLS - 1) / MIK_MASK_CELLS;
  const size_t padded = (size_t)nblk * MIK_MASK_CELLS;
  MIKC(h->pin_in.ensure(padded));
  MIKC(h->mask_dev.ensure(padded));
  MIKC(h->mask_cnt.ensure(sizeof(unsigned) * (size_t)(nblk + 1)));
  host_copy(h->pin_in.p, mask, (size_t)ncells);
  memset(h->pin_in.as<char>() + ncells, 1, padded - (size_t)ncells);
  HIPC(hipMemcpyAsync(h->mask_dev.p, h->pin_in.p, padded, hipMemcpyHostToDevice, h->stream));
  unsigned* cnt = h->mask_cnt.as<unsigned>();
  hipLaunchKernelGGL(k_mask_count, dim3((unsigned)nblk), dim3(256), 0, h->stream, h->mask_dev.as<uint4>(), cnt);
  hipLaunchKernelGGL(k_mask_scan, dim3(1), dim3(1024), 0, h->stream, cnt, nblk);
  HIPC(hipGetLastError());
  unsigned total = 0;
  HIPC(hipMemcpyAsync(&total, cnt + nblk, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  *n_out = (long)total;
  if (total == 0 || (long)total == ncells) return MIK_OK;
  MIKC(h->grid_idx.ensure(sizeof(unsigned) * (size_t)total));
  MIKC(h->scatter_pin.ensure(sizeof(unsigned) * (size_t)total));
  hipLaunchKernelGGL(k_mask_write, dim3((unsigned)nblk), dim3(256), 0, h->stream, h->mask_dev.as<uint4>(), cnt, h->grid_idx.as<unsigned>());
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(h->scatter_pin.p, h->grid_idx.p, sizeof(unsigned) * (size_t)total, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  return MIK_OK;
}

int mik_set_grid(mik_handle* h, const mik_grid* g) {
  if (!h || !g) return fail(MIK_EINVAL, "mik_set_grid: NULL argument");
  if (!h->have_problem) return fail(MIK_ESTATE, "mik_set_grid: set the problem first");
  if (g->ndim != h->ndim) return fail(MIK_EINVAL, "mik_set_grid: ndim differs from the problem's");
  if (g->nx < 1 || g->ny < 1 || (g->ndim == 3 && g->nz < 1)) return fail(MIK_EINVAL, "mik_set_grid: empty axis");
  if (!g->gx || !g->gy || (g->ndim == 3 && !g->gz)) return fail(MIK_EINVAL, "mik_set_grid: axis arrays missing");
  const double cells = (double)g->nx * (double)g->ny * (g->ndim == 3 ? (double)g->nz : 1.0);
  if (cells >= 9.0e15) return fail(MIK_EINVAL, "mik_set_grid: grid too large");
  // cell_count < 0 (write -1): the whole grid; 0: an EMPTY range (a rank of a sharded run with more ranks than cells) -- nothing is
  // kriged, mik_get_results writes nothing
  const bool whole = g->cell_count < 0;
  const long first = whole ? 0 : g->cell_first;
  const long ncells = whole ? (long)cells : g->cell_count;
  if (first < 0 || (double)first + (double)ncells > cells) return fail(MIK_EINVAL, "mik_set_grid: cell range outside the grid");
  if (ncells >= 4294967296L) return fail(MIK_EINVAL, "mik_set_grid: more than 2^32 - 1 cells in one call (use cell_first / cell_count)");
  if (h->nextra > 0 && !g->extra_rows && ncells > 0) return fail(MIK_EINVAL, "extra_rows missing for host-evaluated drifts");
  h->npt_total = ncells;
  long n = ncells;
  h->masked = false;
  if (g->mask && ncells > 0) {
    MIKC(compact_mask(h, g->mask, ncells, &n));
    h->masked = n != ncells;
  }
  const unsigned* ip = h->masked ? h->scatter_pin.as<unsigned>() : nullptr;
  const int members = (int)h->kids.size() + 1;
  return for_each_device(h, [&](int i, mik_handle* d) {
    long lo, cnt;
    slab_of(n, members, i, &lo, &cnt);
    return one_set_grid(d, i == 0, g, ip, lo, cnt, first, ncells);
  });
}

// style='points': the anisotropy adjustment (core.py:120-193) of coordinates mik_set_points uploaded RAW, in place on every member's
// slab -- k_grid_points' arithmetic (the reference's order, dot products accumulated like np.dot) without the meshgrid
int mik_adjust_points(mik_handle* h, const double center[3], const double rot[9], const double stretch[3]) {
  if (!h || !center || !rot || !stretch) return fail(MIK_EINVAL, "mik_adjust_points: NULL argument");
  if (!h->have_points || h->points_from_grid) return fail(MIK_ESTATE, "mik_adjust_points: set the points with mik_set_points first");
  if (h->points_adjusted) return fail(MIK_ESTATE, "mik_adjust_points: the resident points have already been adjusted");
  h->points_adjusted = true;
  h->ps_valid = false;
  return for_each_device(h, [&](int, mik_handle* m) -> int {
    HIPC(hipSetDevice(m->device));
    if (m->npt == 0) return MIK_OK;
    GridArgs a{};
    a.from_points = 1, a.adjust = 1, a.ndim = m->ndim, a.n = m->npt;
    for (int i = 0; i < 3; ++i) a.c[i] = center[i], a.st[i] = stretch[i];
    for (int i = 0; i < 9; ++i) a.rot[i] = rot[i];
    a.px = m->px.as<double>(), a.py = m->py.as<double>(), a.pz = m->ndim == 3 ? m->pz.as<double>() : nullptr;
    hipLaunchKernelGGL(k_grid_points, dim3((unsigned)((m->npt + 255) / 256)), dim3(256), 0, m->stream, a);
    HIPC(hipGetLastError());
    HIPC(hipStreamSynchronize(m->stream));
    return MIK_OK;
  });
}

int mik_get_points(mik_handle* h, double* px_out, double* py_out, double* pz_out) {
  if (!h || !px_out || !py_out) return fail(MIK_EINVAL, "mik_get_points: NULL argument");
  if (!h->have_points) return fail(MIK_ESTATE, "mik_get_points: set points first");
  if (h->ndim == 3 && !pz_out) return fail(MIK_EINVAL, "mik_get_points: pz_out missing");
  long off = 0;  // the members' slabs follow each other in the unmasked sequence
  for (int i = 0; i <= (int)h->kids.size(); ++i) {
    mik_handle* d = member(h, i);
    HIPC(hipSetDevice(d->device));
    if (d->npt > 0) {
      HIPC(hipMemcpy(px_out + off, d->px.p, sizeof(double) * d->npt, hipMemcpyDeviceToHost));
      HIPC(hipMemcpy(py_out + off, d->py.p, sizeof(double) * d->npt, hipMemcpyDeviceToHost));
      if (h->ndim == 3) HIPC(hipMemcpy(pz_out + off, d->pz.p, sizeof(double) * d->npt, hipMemcpyDeviceToHost));
    }
    off += d->npt;
  }
  HIPC(hipSetDevice(h->device));
  return MIK_OK;
}

int64_t mik_points_resident(mik_handle* h) {
  if (!h || !h->have_points) return 0;
  long n = 0;
  for (int i = 0; i <= (int)h->kids.size(); ++i) n += member(h, i)->npt;
  return n;
}

int mik_predict(mik_handle* h) {
  if (!h) return fail(MIK_ESTATE, "mik_predict: NULL handle");
  // The exchange mik_factor started may still be in flight.  A transfer only reads the leader's matrix, so the leader can
  // krige its slab while it runs -- but only a COPY-ENGINE transfer (peer copies) is overlapped by default: an RCCL
  // broadcast needs compute units on the root, and once the leader's persistent contraction launch (all VGPRs of every
  // SIMD for 47 - 480 ms) is resident the root's part would queue behind it and every member would wait for that.
  // "async_exchange" 2 overlaps any path (for A/B runs on a real node: bench.py's overlap trial).
  if (h->xjob && !(h->xjob->path == 2 || h->opt_async_exchange == 2)) MIKC(join_exchange(h));
  if (h->kids.empty() || !h->xjob) return for_each_device(h, [](int, mik_handle* d) { return one_predict(d); });
  // the leader kriges its slab NOW, the members start as soon as their copies have arrived and been verified
  int lrc = MIK_OK;
  std::string lerr;
  std::thread leader([&] {
    g_err.clear();
    lrc = one_predict(h);
    lerr = g_err;
  });
  int rc = join_exchange(h);
  std::string err = g_err;
  if (rc == MIK_OK) {
    rc = for_each_device(h, [](int i, mik_handle* d) { return i == 0 ? MIK_OK : one_predict(d); });
    err = g_err;
  }
  leader.join();
  (void)hipSetDevice(h->device);
  if (lrc != MIK_OK) return fail(lrc, "device " + std::to_string(h->device) + " (group member 0): " + lerr);
  if (rc != MIK_OK) return fail(rc, err);
  return MIK_OK;
}

int mik_predict_moving_window(mik_handle* h, int n_closest) {
  if (!h) return fail(MIK_ESTATE, "mik_predict_moving_window: NULL handle");
  MIKC(join_exchange(h));
  return for_each_device(h, [n_closest](int, mik_handle* d) { return one_predict_mw(d, n_closest); });
}

int mik_statistics(mik_handle* h, double* k_out, double* ss_out) {
  if (!h || !k_out || !ss_out) return fail(MIK_EINVAL, "mik_statistics: NULL argument");
  if (!h->have_problem) return fail(MIK_ESTATE, "mik_statistics: set the problem first");
  if (h->p != 0) return fail(MIK_EINVAL, "statistics use the ordinary-kriging system (core.py:654-756): no drift terms");
  MIKC(join_exchange(h));
  HIPC(hipSetDevice(h->device));
  const int N = h->N, Ns = N + 1;
  const long ld = ((Ns + 63) / 64) * 64;
  MIKC(ensure_factor_buffers(h));
  MIKC(h->flag.ensure(sizeof(int)));
  HIPC(hipMemsetAsync(h->flag.p, 0, sizeof(int), h->stream));
  if (!h->geo) {  // coincident stations make the growing systems singular (np.linalg.solve would raise)
    if (h->ndim == 3)
      hipLaunchKernelGGL(k_stat_dupes<3>, dim3((N + 255) / 256), dim3(256), 0, h->stream, (const double*)h->xs.as<double>(),
                         (const double*)h->ys.as<double>(), (const double*)h->zs.as<double>(), N, h->flag.as<int>());
    else
      hipLaunchKernelGGL(k_stat_dupes<2>, dim3((N + 255) / 256), dim3(256), 0, h->stream, (const double*)h->xs.as<double>(),
                         (const double*)h->ys.as<double>(), (const double*)nullptr, N, h->flag.as<int>());
  }
  if (h->t_state != 1) {
    MIKC(launch_assemble(h, 0.0));
    h->t_state = 1;
    h->have_factor = false;
  }
  MIKC(h->stat_S.ensure(sizeof(double) * (size_t)ld * ld));
  MIKC(h->stat_x.ensure(sizeof(double) * (size_t)ld + 64));
  MIKC(h->stat_out.ensure(sizeof(double) * 2 * (size_t)N));
  HIPC(hipMemsetAsync(h->stat_S.p, 0, h->stat_S.bytes, h->stream));
  HIPC(hipMemsetAsync(h->stat_out.p, 0, h->stat_out.bytes, h->stream));
  double* S = h->stat_S.as<double>();
  double* x = h->stat_x.as<double>();
  double* scal = x + ld;  // 1/s lives behind the vector
  double* kd = h->stat_out.as<double>();
  double* sd = kd + N;
  const double init[4] = {0.0, 1.0, 1.0, 0.0};  // inverse of [[0,1],[1,0]] (Lagrange row + station 0)
  HIPC(hipMemcpy2DAsync(S, sizeof(double) * ld, init, sizeof(double) * 2, sizeof(double) * 2, 2, hipMemcpyHostToDevice, h->stream));
  const double* T = h->T.as<double>();
  for (int i = 1; i < N; ++i) {
    const int m = i + 1;
    const double* Trow = T + (long)i * h->Mp;
    hipLaunchKernelGGL(k_stat_matvec, dim3((m + 3) / 4), dim3(256), 0, h->stream, (const double*)S, ld, m, Trow, x);
    hipLaunchKernelGGL(k_stat_reduce, dim3(1), dim3(256), 0, h->stream, (const double*)x, m, Trow,
                       (const double*)h->vals.as<double>(), kd + i, sd + i, scal);
    if (i + 1 < N)
      hipLaunchKernelGGL(k_stat_update, dim3((m + 64) / 64, (m + 64) / 64), dim3(256), 0, h->stream, S, ld, m,
                         (const double*)x, (const double*)scal);
  }
  HIPC(hipGetLastError());
  int flag = 0;
  HIPC(hipMemcpyAsync(&flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipMemcpyAsync(k_out, kd, sizeof(double) * N, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipMemcpyAsync(ss_out, sd, sizeof(double) * N, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  if (flag) return fail(MIK_ESINGULAR, "Singular matrix");
  return MIK_OK;
}

int mik_experimental_variogram(mik_handle* h, int nlags, double* lags_out, double* semi_out, int32_t* n_out) {
  if (!h || !lags_out || !semi_out || !n_out) return fail(MIK_EINVAL, "mik_experimental_variogram: NULL argument");
  if (!h->have_problem) return fail(MIK_ESTATE, "mik_experimental_variogram: set the problem first");
  if (nlags < 1 || nlags > MIK_VG_MAXLAGS) return fail(MIK_EINVAL, "nlags must be in 1..64");
  if (h->N < 2) return fail(MIK_EINVAL, "need at least two stations");
  HIPC(hipSetDevice(h->device));
  const int N = h->N, nt = (N + 63) / 64;
  const long nblocks = (long)nt * nt;
  DevBuf mm, edges, part;
  MIKC(mm.ensure(sizeof(double) * 2 * nblocks));
  MIKC(edges.ensure(sizeof(double) * (nlags + 1)));
  MIKC(part.ensure(sizeof(double) * 3 * nlags * nblocks));
  const double *xs = h->xs.as<double>(), *ys = h->ys.as<double>(), *zs = h->zs.as<double>(), *vv = h->vals.as<double>();
  const int kd = h->geo ? 1 : h->ndim;
  dim3 grid(nt, nt);
  if (kd == 1) hipLaunchKernelGGL(k_vg_minmax<1>, grid, dim3(256), 0, h->stream, xs, ys, zs, N, mm.as<double>());
  else if (kd == 3) hipLaunchKernelGGL(k_vg_minmax<3>, grid, dim3(256), 0, h->stream, xs, ys, zs, N, mm.as<double>());
  else hipLaunchKernelGGL(k_vg_minmax<2>, grid, dim3(256), 0, h->stream, xs, ys, zs, N, mm.as<double>());
  std::vector<double> hmm(2 * nblocks);
  HIPC(hipMemcpyAsync(hmm.data(), mm.p, sizeof(double) * 2 * nblocks, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  double dmin = 1e300, dmax = -1e300;
  for (long b = 0; b < nblocks; ++b) {
    dmin = std::min(dmin, hmm[2 * b]);
    dmax = std::max(dmax, hmm[2 * b + 1]);
  }
  // core.py:466-471: bins = [dmin + n*dd for n in range(nlags)] + [dmax + 0.001]
  const double dd = (dmax - dmin) / nlags;
  std::vector<double> he(nlags + 1);
  for (int n = 0; n < nlags; ++n) he[n] = dmin + n * dd;
  he[nlags] = dmax + 0.001;
  HIPC(hipMemcpyAsync(edges.p, he.data(), sizeof(double) * (nlags + 1), hipMemcpyHostToDevice, h->stream));
  if (kd == 1) hipLaunchKernelGGL(k_vg_bin<1>, grid, dim3(256), 0, h->stream, xs, ys, zs, vv, N, nlags, (const double*)edges.as<double>(), part.as<double>());
  else if (kd == 3) hipLaunchKernelGGL(k_vg_bin<3>, grid, dim3(256), 0, h->stream, xs, ys, zs, vv, N, nlags, (const double*)edges.as<double>(), part.as<double>());
  else hipLaunchKernelGGL(k_vg_bin<2>, grid, dim3(256), 0, h->stream, xs, ys, zs, vv, N, nlags, (const double*)edges.as<double>(), part.as<double>());
  HIPC(hipGetLastError());
  std::vector<double> hp((size_t)3 * nlags * nblocks);
  HIPC(hipMemcpyAsync(hp.data(), part.p, sizeof(double) * hp.size(), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  mm.release();
  edges.release();
  part.release();
  int nv = 0;
  for (int n = 0; n < nlags; ++n) {
    double sdv = 0.0, sgv = 0.0, scv = 0.0;
    for (long b = 0; b < nblocks; ++b) {
      const double* o = hp.data() + (size_t)b * 3 * nlags;
      sdv += o[n];
      sgv += o[nlags + n];
      scv += o[2 * nlags + n];
    }
    if (scv > 0.0) {  // empty bins are dropped (core.py:498-499)
      lags_out[nv] = sdv / scv;
      semi_out[nv] = sgv / scv;
      ++nv;
    }
  }
  *n_out = nv;
  return MIK_OK;
}

int mik_synchronize(mik_handle* h) {
  if (!h) return fail(MIK_EINVAL, "mik_synchronize: NULL handle");
  MIKC(join_exchange(h));
  for (int i = 0; i <= (int)h->kids.size(); ++i) {
    mik_handle* d = member(h, i);
    HIPC(hipSetDevice(d->device));
    HIPC(hipStreamSynchronize(d->stream));
    HIPC(hipStreamSynchronize(d->stream_d2h));
  }
  HIPC(hipSetDevice(h->device));
  return MIK_OK;
}

// z and sigma^2 of one device's slab, from its page-locked landing zone (filled chunk by chunk while mik_predict ran) into
// the caller's arrays at the slab's place
static int one_get_results(mik_handle* h, double* z_out, double* ss_out) {
  if (!h->have_results) return fail(MIK_ESTATE, "mik_get_results: predict first");
  HIPC(hipSetDevice(h->device));
  HIPC(hipEventSynchronize(h->ev_d2h));
  const long n = h->npt;
  const double* hz = h->pin_out.as<double>();
  const double* hs = hz + n;
  if (n == 0) return MIK_OK;
  if (h->scatter32) {
    const unsigned* ix = h->scatter32;
    parallel_chunks(n, [&](int, long b, long e) {
      for (long i = b; i < e; ++i) {
        z_out[ix[i]] = hz[i];
        ss_out[ix[i]] = hs[i];
      }
    });
    return MIK_OK;
  }
  if (h->scatter.empty()) {
    host_copy(z_out + h->out_off, hz, sizeof(double) * n);
    host_copy(ss_out + h->out_off, hs, sizeof(double) * n);
    return MIK_OK;
  }
  const long* ix = h->scatter.data();
  parallel_chunks(n, [&](int, long b, long e) {
    for (long i = b; i < e; ++i) {
      z_out[ix[i]] = hz[i];
      ss_out[ix[i]] = hs[i];
    }
  });
  return MIK_OK;
}

int mik_get_results(mik_handle* h, double* z_out, double* ss_out) {
  if (!h || !z_out || !ss_out) return fail(MIK_EINVAL, "mik_get_results: NULL argument");
  if (!h->have_results) return fail(MIK_ESTATE, "mik_get_results: predict first");
  if (h->masked) {  // masked points keep 0.0 (cok.pyx:25-26)
    parallel_chunks(h->npt_total, [&](int, long b, long e) {
      memset(z_out + b, 0, sizeof(double) * (size_t)(e - b));
      memset(ss_out + b, 0, sizeof(double) * (size_t)(e - b));
    });
  }
  return for_each_device(h, [=](int, mik_handle* d) { return one_get_results(d, z_out, ss_out); });
}

int mik_take_results(mik_handle* h, double** z_out, double** ss_out) {
  if (!h || !z_out || !ss_out) return fail(MIK_EINVAL, "mik_take_results: NULL argument");
  if (!h->have_results) return fail(MIK_ESTATE, "mik_take_results: predict first");
  if (!h->kids.empty() || h->masked || !h->scatter.empty() || h->scatter32 || h->out_off != 0 || h->npt != h->npt_total || h->npt == 0)
    return fail(MIK_ESTATE, "mik_take_results: only for one device and unmasked points (use mik_get_results)");
  {  // page-locked memory out on loan is bounded (a caller that keeps many results -- time steps, CV folds -- would otherwise pin
     // without limit): beyond MIK_PIN_LENT_CAP bytes (default 4 GiB) the copying mik_get_results is the call
    static const double cap = env_seconds("MIK_PIN_LENT_CAP", 4294967296.0);
    size_t lent = 0;
    {
      std::lock_guard<std::mutex> lk(g_pin_mutex);
      for (const auto& kv : g_pin_lent) lent += kv.second;
    }
    if ((double)lent + (double)h->pin_out.bytes > cap)
      return fail(MIK_ESTATE, "mik_take_results: page-locked memory on loan would exceed MIK_PIN_LENT_CAP (use mik_get_results)");
  }
  HIPC(hipSetDevice(h->device));
  HIPC(hipEventSynchronize(h->ev_d2h));
  const size_t bytes = h->pin_out.bytes;
  double* base = static_cast<double*>(h->pin_out.lend());
  *z_out = base;
  *ss_out = base + h->npt;
  h->have_results = false;  // they have left the handle
  // the landing zone of the NEXT predict now (recycled from the pool, or page-locked here): a loop of execute() calls then
  // allocates in its first call only, not in the set-up of its second one
  (void)h->pin_out.ensure(bytes);
  return MIK_OK;
}

void mik_release_results(double* z) {
  if (!z) return;
  std::lock_guard<std::mutex> lk(g_pin_mutex);
  auto it = g_pin_lent.find((void*)z);
  if (it == g_pin_lent.end()) return;
  const size_t bytes = it->second;
  g_pin_lent.erase(it);
  // the pool keeps the most recently returned buffers (at most 16, 4 GB together): a full pool gives up its OLDEST entries for the newcomer.
  // (Round 5 kept six and refused the newcomer: after a run over several grid sizes the pool held six stale sizes, and every call of the
  // current size paid a hipHostFree + hipHostMalloc of its landing zone -- both synchronise the device -- 3 ms per execute() of 10^6 points: the
  // "host side" of the k = 10 bench line, scripts/r06_diag_mw10.py.)
  if (bytes > (size_t)4 << 30) {
    (void)hipHostFree(z);
    return;
  }
  size_t pooled = bytes;
  for (auto& e : g_pin_pool) pooled += e.second;
  while (!g_pin_pool.empty() && (g_pin_pool.size() >= 16 || pooled > (size_t)4 << 30)) {
    pooled -= g_pin_pool.front().second;
    (void)hipHostFree(g_pin_pool.front().first);
    g_pin_pool.erase(g_pin_pool.begin());
  }
  g_pin_pool.emplace_back((void*)z, bytes);
}

int mik_get_timing(mik_handle* h, mik_timing* out) {
  if (!h || !out) return fail(MIK_EINVAL, "mik_get_timing: NULL argument");
  MIKC(join_exchange(h));  // the exchange figures are final only then
  *out = h->tm;
  out->n_devices = (int)h->kids.size() + 1;
  out->exchange_path = h->exchange_used;
  out->exchange_ms = h->exchange_ms;
  out->exchange_wait_ms = h->exchange_wait_ms;
  out->exchange_bytes = h->exchange_bytes;
  out->exchange_fallbacks = h->exchange_fallbacks;
  out->rccl_ranks = h->rccl_ranks;
  for (mik_handle* k : h->kids) out->predict_ms = std::max(out->predict_ms, k->tm.predict_ms);  // the group's predict = its slowest member
  return MIK_OK;
}

int mik_get_device_timing(mik_handle* h, int member_index, mik_timing* out) {
  if (!h || !out) return fail(MIK_EINVAL, "mik_get_device_timing: NULL argument");
  if (member_index < 0 || member_index > (int)h->kids.size()) return fail(MIK_EINVAL, "mik_get_device_timing: no such group member");
  MIKC(join_exchange(h));
  *out = member(h, member_index)->tm;
  out->n_devices = (int)h->kids.size() + 1;
  out->exchange_path = h->exchange_used;
  out->exchange_ms = h->exchange_ms;
  out->exchange_wait_ms = h->exchange_wait_ms;
  out->exchange_bytes = h->exchange_bytes;
  out->exchange_fallbacks = h->exchange_fallbacks;
  out->rccl_ranks = h->rccl_ranks;
  out->reserved = member(h, member_index)->device;
  return MIK_OK;
}

int mik_krige_execute(int device, const mik_problem* p, const mik_points* g, double* z_out, double* ss_out) {
  mik_handle* h = nullptr;
  int r = mik_create(device, &h);
  if (r != MIK_OK) return r;
  r = mik_set_problem(h, p);
  if (r == MIK_OK) r = mik_factor(h);
  if (r == MIK_OK) r = mik_set_points(h, g);
  if (r == MIK_OK) r = mik_predict(h);
  if (r == MIK_OK) r = mik_get_results(h, z_out, ss_out);
  std::string keep = g_err;
  mik_destroy(h);
  g_err = keep;
  return r;
}

int mik_selftest_mfma(int device) {
  HIPC(hipSetDevice(device));
  double* d = nullptr;
  HIPC(hipMalloc(&d, sizeof(double) * 256));
  hipLaunchKernelGGL(k_selftest_mfma, dim3(1), dim3(64), 0, 0, d);
  double out[256];
  HIPC(hipMemcpy(out, d, sizeof out, hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; ++i)
    for (int j = 0; j < 16; ++j) {
      double ref = 0.0;
      for (int k = 0; k < 4; ++k) ref += (double)(i * 7 + k * 3 + 1) * (double)(k * 11 + j * 5 + 2);
      if (out[i * 16 + j] != ref) {
        char b[200];
        snprintf(b, sizeof b, "mfma_f64_16x16x4 layout mismatch at (%d,%d): got %g want %g", i, j, out[i * 16 + j], ref);
        (void)hipFree(d);
        return fail(MIK_EHIP, b);
      }
    }
  hipLaunchKernelGGL(k_selftest_mfma4, dim3(1), dim3(64), 0, 0, d);
  HIPC(hipMemcpy(out, d, sizeof(double) * 64, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) {
      double ref = 0.0;
      for (int k = 0; k < 4; ++k) ref += (double)(i * 7 + k * 3 + 1) * (double)(k * 11 + j * 5 + 2);
      if (out[i * 16 + j] != ref) {
        char b[200];
        snprintf(b, sizeof b, "mfma_f64_4x4x4_4b layout mismatch at (%d,%d): got %g want %g", i, j, out[i * 16 + j], ref);
        return fail(MIK_EHIP, b);
      }
    }
  return MIK_OK;
}

int mik_selftest_exp(int device, const double* x, double* out, int n) {
  if (!x || !out || n < 0) return fail(MIK_EINVAL, "mik_selftest_exp: bad argument");
  if (n == 0) return MIK_OK;
  HIPC(hipSetDevice(device));
  DevBuf dx, dy;
  MIKC(dx.ensure(sizeof(double) * (size_t)n));
  MIKC(dy.ensure(sizeof(double) * (size_t)n));
  HIPC(hipMemcpy(dx.p, x, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_selftest_exp, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, (const double*)dx.as<double>(), dy.as<double>(), n);
  HIPC(hipGetLastError());
  HIPC(hipMemcpy(out, dy.p, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  return MIK_OK;
}

// ---- multi-GPU ---------------------------------------------------------------------------------
int mik_comm_unique_id(char id_out[128]) {
  MIKC(rccl_load());
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  NCCLC(g_rccl.GetUniqueId(&id));
  memcpy(id_out, &id, 128);
  return MIK_OK;
}

// a call that may never return (RCCL set-up, a collective), run on a worker thread under a limit; the worker owns what it
// touches through the captures of fn (by value / shared_ptr), so it can be abandoned
struct BoundedCall {
  std::mutex m;
  std::condition_variable cv;
  bool done = false;
  int rc = MIK_OK;
  std::string err;
};
extern "C++" {
template <class F>
static int run_bounded(F fn, double limit_s, const char* what, bool* timed_out) {
  auto st = std::make_shared<BoundedCall>();
  std::thread([st, fn]() mutable {
    g_err.clear();
    const int rc = fn();
    std::lock_guard<std::mutex> lk(st->m);
    st->rc = rc;
    st->err = g_err;
    st->done = true;
    st->cv.notify_all();
  }).detach();
  std::unique_lock<std::mutex> lk(st->m);
  *timed_out = !st->cv.wait_for(lk, std::chrono::duration<double>(limit_s), [&] { return st->done; });
  if (*timed_out) {
    char b[200];
    snprintf(b, sizeof b, "%s did not finish within %.1f s", what, limit_s);
    return fail(MIK_ERCCL, b);
  }
  g_err = st->err;
  return st->rc;
}
}  // extern "C++"

int mik_comm_init(mik_handle* h, int nranks, int rank, const char id[128]) {
  if (!h || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(MIK_EINVAL, "mik_comm_init: bad argument");
  if (!h->kids.empty()) return fail(MIK_EINVAL, "mik_comm_init: this handle already spans several devices (mik_set_devices); use one or the other");
  if (g_rccl_dead.load()) return fail(MIK_ERCCL, "RCCL disabled for this process: " + g_rccl_dead_why);
  MIKC(rccl_load());
  HIPC(hipSetDevice(h->device));
  ncclUniqueId uid;
  memcpy(&uid, id, 128);
  auto out = std::make_shared<ncclComm_t>(nullptr);
  const int dev = h->device;
  bool timed_out = false;
  const int rc = run_bounded([=] {
    HIPC(hipSetDevice(dev));
    NCCLC(g_rccl.CommInitRank(out.get(), nranks, uid, rank));
    return MIK_OK;
  }, h->rccl_init_limit, "ncclCommInitRank", &timed_out);
  if (timed_out) {
    g_rccl_dead_why = "a bounded wait on ncclCommInitRank ran out";
    g_rccl_dead.store(true);
  }
  MIKC(rc);
  h->comm = *out;
  h->nranks = nranks;
  h->rank = rank;
  return MIK_OK;
}

int mik_bcast_factor(mik_handle* h, int root) {
  if (!h || !h->comm) return fail(MIK_ESTATE, "mik_bcast_factor: no communicator");
  if (!h->have_problem) return fail(MIK_ESTATE, "mik_bcast_factor: set the problem on every rank first");
  if (h->rank == root && !h->have_factor) return fail(MIK_ESTATE, "mik_bcast_factor: root has not factored");
  HIPC(hipSetDevice(h->device));
  MIKC(ensure_factor_buffers(h));
  if (!h->xstream) HIPC(hipStreamCreateWithFlags(&h->xstream, hipStreamNonBlocking));
  HIPC(hipStreamSynchronize(h->stream));  // the broadcast runs on the exchange stream: the factor (root) / earlier reads are done
  const size_t Mp = h->Mp;
  const int dev = h->device;
  // every rank decides by the same rule on the same problem and options: the packed upper block triangle of an exactly symmetric inverse
  // (exchange_triangle; the half-sweep choice is a function of the model and the size, the same on every rank)
  const bool tri = exchange_triangle(h);
  const size_t xlen = tri ? tri_len(Mp) : Mp * Mp;
  if (tri) {
    MIKC(h->xpack.ensure(sizeof(double) * xlen));
    if (h->rank == root) {
      hipLaunchKernelGGL(k_tri_pack, dim3((unsigned)Mp), dim3(256), 0, h->stream, h->T.as<double>(), Mp, h->xpack.as<double>(), 0);
      HIPC(hipGetLastError());
      HIPC(hipStreamSynchronize(h->stream));
    }
  }
  double* T = h->T.as<double>();
  double* X = tri ? h->xpack.as<double>() : T;
  double* cv = h->cvec.as<double>();
  ncclComm_t comm = h->comm;
  hipStream_t xs = h->xstream;
  const bool unpack = tri && h->rank != root;
  bool timed_out = false;
  const int rc = run_bounded([=] {
    HIPC(hipSetDevice(dev));
    NCCLC(g_rccl.Broadcast(X, X, xlen, ncclDouble, root, comm, xs));
    NCCLC(g_rccl.Broadcast(cv, cv, Mp, ncclDouble, root, comm, xs));
    if (unpack) {
      hipLaunchKernelGGL(k_tri_pack, dim3((unsigned)Mp), dim3(256), 0, xs, T, Mp, X, 1);
      MIKC(launch_mirror_upper(T, (long)Mp, xs));
    }
    HIPC(hipStreamSynchronize(xs));
    return MIK_OK;
  }, h->rccl_bcast_limit, "ncclBroadcast of the factor", &timed_out);
  if (timed_out) {  // nothing the abandoned collective may still touch is reused: stream, buffers and communicator are leaked
    h->xstream = nullptr;
    if (h->rank != root) {
      h->T.leak();
      h->cvec.leak();
      h->xpack.leak();
    }
    h->comm = nullptr;
    g_rccl_dead_why = "a bounded wait on ncclBroadcast ran out";
    g_rccl_dead.store(true);
  }
  MIKC(rc);
  if (h->rank != root) {  // the root decided by the same rules on the same problem and options
    h->factor_sorted = want_sorted(h);
    h->factor_eq = h->drift_eq && h->opt_drift_eq;
  }
  h->have_factor = true;
  h->t_state = 2;
  h->have_results = false;
  h->xpack_valid = tri;
  h->exchange_bytes = sizeof(double) * (double)(xlen + Mp);
  return MIK_OK;
}

// 4 words: order-independent checksums of the handle's inverted matrix and of c (k_checksum).  One process per GPU: the
// ranks compare theirs with the root's after mik_bcast_factor (pykrige_amd.dist) -- a broken broadcast is detected, not kriged with.
int mik_factor_checksum(mik_handle* h, uint64_t out[4]) {
  if (!h || !out) return fail(MIK_EINVAL, "mik_factor_checksum: NULL argument");
  if (!h->have_factor) return fail(MIK_ESTATE, "mik_factor_checksum: no factor");
  MIKC(join_exchange(h));
  HIPC(hipSetDevice(h->device));
  MIKC(h->xsum.ensure(4 * sizeof(unsigned long long)));
  unsigned long long* sd = h->xsum.as<unsigned long long>();
  const size_t Mp = h->Mp;
  HIPC(hipMemsetAsync(sd, 0, 4 * sizeof(unsigned long long), h->stream));
  // after a triangle broadcast (mik_bcast_factor) the ranks agree on the packed upper block triangle, not on the lower one nobody sent
  if (h->xpack_valid) hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, h->stream, (const unsigned long long*)h->xpack.p, tri_len(Mp), sd);
  else hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, h->stream, (const unsigned long long*)h->T.p, Mp * Mp, sd);
  hipLaunchKernelGGL(k_checksum, dim3(4), dim3(256), 0, h->stream, (const unsigned long long*)h->cvec.p, Mp, sd + 2);
  HIPC(hipGetLastError());
  unsigned long long host[4];
  HIPC(hipMemcpyAsync(host, sd, sizeof host, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  for (int i = 0; i < 4; ++i) out[i] = host[i];
  return MIK_OK;
}

}  // extern "C"

