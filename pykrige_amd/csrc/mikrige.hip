// mikrige.hip -- host orchestration + C ABI (include/mikrige.h) of the MI355X kriging execute() path.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude pykrige_amd/csrc/mikrige.hip -ldl
// No torch, no BLAS/solver libraries: every kernel is in mik_kernels.h.  RCCL is dlopen()ed on demand.
#include "mik_kernels.h"
#include "../../include/mikrige.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace mik;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) {
  g_err = msg;
  return code;
}
#define HIPC(x)                                                                                         \
  do {                                                                                                  \
    hipError_t e_ = (x);                                                                                \
    if (e_ != hipSuccess) {                                                                             \
      char b_[512];                                                                                     \
      snprintf(b_, sizeof b_, "HIP error '%s' at %s:%d (%s)", hipGetErrorString(e_), __FILE__, __LINE__, #x); \
      return fail(MIK_EHIP, b_);                                                                        \
    }                                                                                                   \
  } while (0)
#define MIKC(x)            \
  do {                     \
    int r_ = (x);          \
    if (r_ != MIK_OK) return r_; \
  } while (0)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  int ensure(size_t need) {
    if (need <= bytes && p) return MIK_OK;
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
    if (need == 0) return MIK_OK;
    HIPC(hipMalloc(&p, need));
    bytes = need;
    return MIK_OK;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
  void leak() {  // give the memory up without freeing it (an abandoned transfer may still write it)
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// Page-locked buffers that left a handle with mik_take_results (the caller's result arrays ARE the landing zone) come back
// through mik_release_results into a small process-wide pool and are handed to the next handle that needs one: a loop of
// execute() calls whose results are dropped allocates (and page-locks) nothing in steady state.
static std::mutex g_pin_mutex;
static std::vector<std::pair<void*, size_t>> g_pin_pool;
static std::map<void*, size_t> g_pin_lent;

// page-locked host memory: staging of the point coordinates on their way in, landing zone of z / sigma^2 on their way out
struct PinBuf {
  void* p = nullptr;
  size_t bytes = 0;
  PinBuf() = default;
  PinBuf(const PinBuf&) = delete;
  PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf() { release(); }
  int ensure(size_t need) {
    if (need <= bytes && p) return MIK_OK;
    release();
    if (need == 0) return MIK_OK;
    {
      std::lock_guard<std::mutex> lk(g_pin_mutex);
      int best = -1;
      for (size_t i = 0; i < g_pin_pool.size(); ++i)
        if (g_pin_pool[i].second >= need && g_pin_pool[i].second <= 2 * need + (1u << 20) &&
            (best < 0 || g_pin_pool[i].second < g_pin_pool[(size_t)best].second))
          best = (int)i;
      if (best >= 0) {
        p = g_pin_pool[(size_t)best].first;
        bytes = g_pin_pool[(size_t)best].second;
        g_pin_pool.erase(g_pin_pool.begin() + best);
        return MIK_OK;
      }
    }
    HIPC(hipHostMalloc(&p, need, hipHostMallocPortable));
    bytes = need;
    return MIK_OK;
  }
  void* lend() {  // ownership passes to the caller (mik_take_results)
    std::lock_guard<std::mutex> lk(g_pin_mutex);
    void* q = p;
    g_pin_lent[q] = bytes;
    p = nullptr;
    bytes = 0;
    return q;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    bytes = 0;
  }
  template <class T>
  T* as() const { return reinterpret_cast<T*>(p); }
};

// host-side copies between the caller's pageable arrays and the page-locked staging buffers: one core moves ~8 GB/s, which at
// 2 x 10^6 points is several per cent of a whole execute(); large copies are cut over a few threads
static void host_copy(void* dst, const void* src, size_t bytes) {
  constexpr size_t PIECE = 8u << 20;
  const size_t nthr = std::min<size_t>(4, bytes / PIECE);
  if (nthr < 2) {
    memcpy(dst, src, bytes);
    return;
  }
  std::vector<std::thread> th;
  const size_t per = ((bytes / nthr + 63) / 64) * 64;
  for (size_t t = 1; t < nthr; ++t) {
    const size_t off = t * per, len = (t + 1 == nthr) ? bytes - off : per;
    th.emplace_back([=] { memcpy((char*)dst + off, (const char*)src + off, len); });
  }
  memcpy(dst, src, per);
  for (auto& t : th) t.join();
}

// O(npt) host loops of the masked styles (index list of the unmasked cells, gathers, the scatter of the results): cut over a few
// threads from ~10^6 elements on (one core does 0.3 - 0.5 ns-bound passes at 2 - 4 ns per element: 50 ms per pass at 1.7e7 cells)
extern "C++" {
template <class F>
static void parallel_chunks(long n, F fn) {  // fn(chunk index, begin, end) over at most 8 contiguous chunks
  const long nthr = std::min<long>(8, n / (1L << 20));
  if (nthr < 2) {
    fn(0, 0L, n);
    return;
  }
  std::vector<std::thread> th;
  const long per = (n + nthr - 1) / nthr;
  for (long t = 1; t < nthr; ++t) th.emplace_back([=] { fn((int)t, t * per, std::min(n, (t + 1) * per)); });
  fn(0, 0L, std::min(n, per));
  for (auto& t : th) t.join();
}
}  // extern "C++"
static int chunks_of(long n) { return (int)std::max<long>(1, std::min<long>(8, n / (1L << 20))); }

// np.nonzero(~mask) (ok.py:700) / `if mask[i]: continue` (cok.pyx:57-58): positions of the unmasked cells, ascending
static void unmasked_positions(const int8_t* mask, long ncells, std::vector<long>& idx) {
  const int nc = chunks_of(ncells);
  std::vector<long> cnt(nc + 1, 0);
  parallel_chunks(ncells, [&](int c, long b, long e) {
    long k = 0;
    for (long i = b; i < e; ++i) k += mask[i] == 0;
    cnt[c + 1] = k;
  });
  for (int c = 0; c < nc; ++c) cnt[c + 1] += cnt[c];
  idx.resize((size_t)cnt[nc]);
  parallel_chunks(ncells, [&](int c, long b, long e) {
    long k = cnt[c];
    for (long i = b; i < e; ++i)
      if (!mask[i]) idx[(size_t)k++] = i;
  });
}

// --- RCCL, loaded lazily so the single-GPU path has no link-time dependency on it ---------------
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  // single-process multi-device use (mik_set_devices): one communicator per device, calls fused in a group
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
  if (g_rccl.lib) return MIK_OK;
  // MIK_RCCL_LIB: load this library instead (the tests' stand-ins whose calls hang, fail or copy)
  const char* names[] = {getenv("MIK_RCCL_LIB") ? getenv("MIK_RCCL_LIB") : "librccl.so", "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* lib = nullptr;
  for (const char* n : names) {
    lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (lib) break;
  }
  if (!lib) return fail(MIK_ERCCL, std::string("cannot dlopen librccl.so: ") + dlerror());
  g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
  g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(lib, "ncclCommInitRank");
  g_rccl.Broadcast = (decltype(g_rccl.Broadcast))dlsym(lib, "ncclBroadcast");
  g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(lib, "ncclCommDestroy");
  g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(lib, "ncclGetErrorString");
  g_rccl.CommInitAll = (decltype(g_rccl.CommInitAll))dlsym(lib, "ncclCommInitAll");
  g_rccl.GroupStart = (decltype(g_rccl.GroupStart))dlsym(lib, "ncclGroupStart");
  g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))dlsym(lib, "ncclGroupEnd");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.Broadcast || !g_rccl.CommDestroy)
    return fail(MIK_ERCCL, "librccl.so lacks an expected symbol");
  g_rccl.lib = lib;
  return MIK_OK;
}
#define NCCLC(x)                                                                                   \
  do {                                                                                             \
    ncclResult_t r_ = (x);                                                                         \
    if (r_ != ncclSuccess) {                                                                       \
      char b_[512];                                                                                \
      snprintf(b_, sizeof b_, "RCCL error '%s' at %s:%d (%s)",                                     \
               g_rccl.GetErrorString ? g_rccl.GetErrorString(r_) : "?", __FILE__, __LINE__, #x);   \
      return fail(MIK_ERCCL, b_);                                                                  \
    }                                                                                              \
  } while (0)

struct mik_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  // problem
  bool have_problem = false, have_factor = false, have_points = false, have_results = false;
  int ndim = 2, model = 0, exact = 1, rl = 0, nwells = 0, nextra = 0;
  int geo = 0;  // coordinates_type == 'geographic' (2-D lon/lat in degrees; kernels are instantiated with NDIM = 1)
  int N = 0, p = 0, M = 0, Mp = 0;
  Vario v{};
  double eps = 1e-10, shift_guess = 0.0;
  bool host_inv = false;
  int pinv = 0;  // pseudo_inv: 0 no, 1 'pinv', 2 'pinvh'
  mik_variogram_fn custom_fn = nullptr;  // variogram_model == 'custom' (model 6): host map d -> gamma(d)
  void* custom_user = nullptr;
  std::vector<double> host_ainv;
  DevBuf xs, ys, zs, vals, wells, extra_cols;
  // range-aware contraction (compact-support variogram, round 4): the dense path keeps a second copy of the stations laid out
  // along a Hilbert curve (sort_perm[i] = caller's index of the station at position i) and the bounding boxes of its
  // 128-station blocks.  factor_sorted says which order the factor in T (and c) is in.
  bool sort_ok = false, factor_sorted = false;
  // drift equilibration (AsmArgs::dsc): per drift term (centre, scale) from the station values; the factor path assembles with it
  // (not with a pseudo-inverse -- pinv(S A S^T) is not S^-T pinv(A) S^-1 -- nor with a caller's inverse); factor_eq = T is in that form
  bool drift_eq = false, factor_eq = false;
  std::vector<double> hdsc;
  DevBuf dsc;
  int opt_drift_eq = 1;  // "drift_eq": 0 = assemble the drift columns as the reference does
  std::vector<int> sort_perm;
  bool stations_same = false;  // mik_set_problem: the station coordinates are the previous problem's (sort_perm is kept)
  std::vector<double> hvals_s;
  DevBuf xs_s, ys_s, zs_s, vals_s, extra_cols_s, sbox;
  int opt_sparse = -1;  // "sparse": -1 = auto (= 1: on for compact-support models), 0 = off, 1 = on, 2 = sorted stations, dense contraction
  DevBuf sp_cand, sp_flags, sp_klist, sp_kcount, sp_nrows, sp_rows, sp_rstart, sp_tiles, sp_xoff, sp_stats, sp_recs;
  int opt_sort_points = -1;  // "sort_points": range-aware contraction over the points of every launch in Hilbert-curve order (k_ps_*): -1 = auto = 1, 0 = off
  DevBuf ps_key[2], ps_idx[2], ps_table, ps_box, ps_x, ps_y, ps_z, ps_zs, ps_sss;
  bool ps_valid = false;     // ps_idx[0] holds the order of the resident points for launches of ps_chunk points
  long ps_chunk = 0;
  int opt_sparse_group = 4;  // "sparse_group": point blocks per group of k_sp_tiles_g's queue order (a group's tiles run on one XCD, tile position
                             // ascending, point block fast): 1 .. 16
  int opt_sparse_epi = 0;    // "sparse_epilogue": k_contract_spg forms a group's term of the quadratic form from global memory after the K loop (0,
                             // default) or from the B tile in LDS at the group's own K step (1: no operand reads in the epilogue -- measured 1.7 %
                             // SLOWER at config 5, 43.1 against 42.4 ms of contraction: the extra registers of the triangle loop cost more)
  int opt_sparse_rows = -1;  // "sparse_rows": 16 = tiles of gathered 16-row groups (k_contract_spg), 128 = aligned row blocks (k_contract_sp),
                             // -1 = auto: 16 wherever 32-bit offsets address the inverse (Mp * Mp * 8 < 2^32)
  // second set (with Bt2): the launches of the range-aware contraction alternate between two lanes on two streams, so that the
  // candidate / right-hand-side / list kernels of one launch and the tail of the previous launch's tile queue overlap
  DevBuf sp2_cand, sp2_flags, sp2_klist, sp2_kcount, sp2_nrows, sp2_rows, sp2_rstart, sp2_tiles, sp2_xoff, part2, queue2, sp2_recs;
  int opt_sparse_lanes = 1;  // "sparse_lanes": 1 = one launch after the other on one stream (default), 2 = two lanes.  Measured
                             // (profiles/r04_sparse_lanes_ab.txt): config-5 slab 64.6 -> 63.4 ms, bench grid 96.3 -> 94.0 ms -- 2 % for a
                             // second 8.4 GB panel and per-launch times that no longer add up: off
  std::vector<double> hxs, hys, hzs;  // host copies of the station coordinates (the moving-window cell grid is built on the host)
  // moving-window neighbour search: stations sorted into a uniform grid of cells
  struct MwGrid {
    int target = -1;  // stations-per-cell target the grid was built for (-1 = none)
    int nx = 1, ny = 1, nz = 1;
    int live = 0;          // axes along which the stations spread (a flat 3-D set has 2): the dimension of their density
    double per_cell = 0;   // mean stations per cell of the grid as built
    double x0 = 0, y0 = 0, z0 = 0, cell = 1;
    DevBuf gx, gy, gz, orig, cstart;
  } grid;
  // factor
  DevBuf T, cvec, Cold, Cnew, Rt, TKt, Dinv, DinvT, P0, P1, cand0, cand1, pivall, flag;
  DevBuf Cold2, Cnew2, Rt2, Dinv2, DinvT2;  // second panel set of the look-ahead sweep
  DevBuf Dinv3, DinvT3;                     // third diagonal-inverse set (panel-stream schedule)
  DevBuf tilemap;                           // k_update's tile order (update_tile_map)
  int tilemap_key[3] = {0, 0, 0};
  hipStream_t stream3 = nullptr;            // panel stream of the sweep (panel kernel + block-column update), high priority
  std::vector<hipEvent_t> ps_events;
  DevBuf Dnext, Dcopy, Cb, Rb;              // early-diagonal chain: 128 x 128 scratch (next diagonal block, its source tile, one block of panel rows)
  hipStream_t stream2 = nullptr;            // the look-ahead branch (next panel) runs here
  std::vector<hipEvent_t> la_events;
  int opt_lookahead = -1;  // -1 = where it pays (>= 24 block columns), 0 = off, 1 = on
  // unpivoted sweep maintaining only the upper block triangle (half the update tiles: -9 % at N=5000, -30 % at N=8000).  The two
  // triangles of the in-place inverse carry different rounding histories, and z / sigma^2 formed from a mirrored triangle
  // lose the small residual of the full sweep on ill-conditioned systems (power variogram with drift terms, cond 3e5: |dz|
  // 3e-9 -> 8e-7).  AUTO (default): on for exponential / spherical models from 24 block columns on (where it pays and where
  // its measured error stays three orders inside the bar) AND only as long as the probe of the result passes
  // (verify_inverse) -- an ill-conditioned set-up of those models falls back to the full sweep by itself.
  int opt_symsweep = -1;  // -1 = auto, 0 = off, 1 = on
  int opt_pinv_fast = 1;   // pseudo_inv: try the deflated regular inverse (duplicated stations) before the Jacobi pseudo-inverse
  int opt_pinv_block = -1; // the Jacobi pseudo-inverse in its block form (k_bj_*: round 4): -1 = from 1536 rows on, 1 = always, 0 = one row pair per workgroup (rounds 1-3)
  // every inverse the device computes is PROBED before it is used (verify_inverse): A c against the data vector (bounds the
  // error of z) and X A e_j against e_j for three station columns (the sigma^2 side).  A failed probe sends the factorisation
  // to the next more careful path: half sweep -> full sweep -> partial pivoting.
  int opt_verify = 1;
  double verify_tol_z = 5e-10, verify_tol_inv = 1e-8;  // calibrated: profiles/r03_inverse_probe_calibration.txt (true |dz| <= 9 res_z, |dss| <= 50 res_inv over 481 runs)
  bool no_half_sweep = false;  // transient: this attempt must not use the half sweep
  bool last_half_sweep = false;
  bool points_from_grid = false;  // the resident points were generated by mik_set_grid (mik_adjust_points refuses them)
  double pts_extent = -1.0;       // largest coordinate extent of the resident points (from the same sample / the grid's axes; -1 = unknown)
  double pts_step = -1.0;         // median step between consecutive resident points (largest coordinate difference; -1 = unknown):
                                  // tells the moving-window search whether 64 consecutive points are neighbours in space
  bool points_adjusted = false;   // mik_adjust_points has transformed the resident points (a second call would transform them twice)
  DevBuf Averify, vbuf;
  std::vector<double> hvals;   // host copy of the station values (the probe compares A c with them)
  int opt_fuse_chain = 1;  // look-ahead sweep: the column update writes the next panel copy too (no copy kernel on the chain)
  int opt_early_diag = -1; // look-ahead sweep: the next diagonal block is built and inverted ahead of the panel / update stream (-1 = with the look-ahead)
  int opt_gate = -1;       // look-ahead sweep: the trailing update waits until the next diagonal inverse has started and leaves
                           // it a CU of its own (k_gate); -1 = where the serial chain, not the update, is the step period
  // round 3: the panel kernel and the update of the NEXT block column run on a third stream beside the trailing update of the
  // step before (events only): -1 = from 24 block columns on, 0 = off, 1 = wherever the early-diagonal schedule runs
  int opt_panel_stream = -1;
  // tile order of the trailing update: 0 = the kernel's own (column by column; default), n > 1 = n x n super-blocks (the tiles an
  // XCD has in flight share n + n operand panels in its L2).  Measured a tie at every size (profiles/r03_k2_panel_stream_ab.txt):
  // the update is not bound by its panel reads.
  int opt_update_map = 0;
  int opt_update_rev = -1;  // "update_rev": the half sweep's trailing update walks its tiles backwards on odd steps (k_update): -1 = auto =
                            // from 45 block columns on (the upper triangle no longer fits half of the 256 MB memory-side cache), 0 / 1
  // trailing update: tiles without a panel / diagonal copy go to memory as fp64 atomic adds (k_update atomic_rmw; same bits).
  // Measured SLOWER (N=5000 4.39 -> 4.84 ms, N=8000 13.98 -> 15.96 ms: the L2's fp64 atomic rate, not latency, is the bound): off.
  int opt_update_atomic = 0;
  int opt_panel_rows = 32;  // rows of the column panel one block of k_panel forms: 32 (round 3), 64 or 128 (one tile, the round-1 form)
  int opt_update_waves = 8; // trailing-update kernel of the block sweep: 4 waves (wave tile 64 x 64) or 8 (32 x 64, default since round 3:
                            // -5 % at N=5000 / 8000, same bits: profiles/r03_update_waves_ab.txt) per 128 x 128 tile
  int opt_diag = 4;        // diagonal-block inverse variant: 0 = 1024 threads (16 waves x 8 rows), 1 = 16x16 grid, 2 = 16x32, 3 = 32x32 (all four:
                           // 128 barrier-separated pivots, the same bits), 4 = blocked, 8 x 16 pivots (round 3; equal to rounding)
  // points
  long npt_total = 0, npt = 0;
  bool masked = false;  // the caller's mask skipped at least one point: outputs are zero-filled before the scatter
  std::vector<long> scatter;  // empty = identity (mik_set_points under a mask)
  const unsigned* scatter32 = nullptr;  // mik_set_grid under a mask: this member's slab of the leader's page-locked index list
  DevBuf mask_dev, mask_cnt;  // the byte mask (padded to whole blocks) and the per-block counts / offsets of its compaction
  PinBuf scatter_pin;         // the compacted index list on the host (leader)
  DevBuf px, py, pz, extra_rows, z, ss;
  DevBuf grid_axes, grid_idx;  // mik_set_grid: the axes and (masked style) the slab's compacted cell numbers
  // work
  DevBuf Bt, Bt2, part, mw_idx, mw_dist, stat_S, stat_x, stat_out, queue;
  int n_cu = 256;
  int t_state = 0;  // what T holds: 0 nothing, 1 the kriging matrix A (shift 0), 2 its inverse
  // options
  int opt_waves = 8;  // waves per contraction block: 4 (wave tile 64x64) or 8 (32x64)
  // symmetric contraction: the queue can hand out equal-length PAIRS of row blocks instead of single tiles.  Measured
  // (profiles/r02_contract_pairs_vs_tiles.txt): L2 hit rate 28 % -> 47 %, fabric reads -19 %, and 2.7 % SLOWER -- co-resident
  // blocks then reach their epilogues together and stop covering each other's bubbles; the kernel is not traffic-bound.  Off.
  int opt_pairs = 0;
  // symmetric contraction (8-wave form): the diagonal block of a tile is contracted as a triangle of 16-row groups -- 36 of
  // its 64 (group, K tile) products (round 3; gemm_core TRI).  0 = the whole diagonal block.
  int opt_tri = 1;
  // symmetric contraction with triangular diagonal blocks: the next tile is popped, and its first K tile sent to LDS, before the
  // epilogue of the current one (k_contract PRE)
  int opt_prefetch = 0;
  int opt_symmetrize = 1;  // T <- (T + T^T) / 2 after a full sweep / the pivoted elimination (k_symmetrize); 0 = as eliminated
  int opt_factor = 0, opt_sym = 1, opt_engine = 0;  // engine: 0 = v_mfma_f64 contraction, 1 = v_fma_f64 (VALU) contraction
  long opt_chunk = 131072;
  int opt_mw_pivot = 0;       // 1 = always solve the moving-window systems with partial pivoting
  int opt_mw_solver = 0;      // 0 = LDL^T of the shifted system in registers (default), 1 = the Gauss-Jordan kernels
  bool mw_force_piv = false;
  int opt_mw_lds_cap = 8192;  // largest candidate buffer the moving-window neighbour search keeps in LDS
  int opt_mw_knn_bound = 1;   // neighbour search: first pass over the 3 x 3 cells with a distance bound (see k_mw_knn)
  int opt_mw_static = 1;      // k_mw_chol instantiated with the variogram model as a compile-time constant where possible (0: the dynamic form, for A/B)
  int opt_mw_knn_lane = 1;    // neighbour search, windows <= 32: one lane per point first (k_mw_knn_lane), k_mw_knn for what it leaves
  int opt_mw_class = 0;       // 100 G + RI: force one thread-grid / register-tile class of k_mw_chol (0 = by window size)
  mik_timing tm{};
  std::vector<hipEvent_t> evpool;
  std::vector<hipEvent_t> pr_events;  // predict: per chunk "right-hand sides written" / "contraction done" (two RHS panels)
  hipEvent_t ev_sort = nullptr;       // predict: the points of every launch are in order (k_ps_*: timed, and the second lane waits for it)
  hipEvent_t ev_chunk = nullptr;      // predict: chunk finished on the compute stream (the result copies wait for it)
  // "rhs_overlap": k_rhs of the next chunk on a second stream while the current chunk is contracted (two RHS panels).
  // Measured (profiles/r03_chunk_and_rhs_overlap_sweep_c2.txt): it does run concurrently -- and the contraction slows down by
  // exactly the time k_rhs takes (362.8 + 9.1 ms serial = 372.6 ms per 10^6 points; 372.8 ms overlapped): fp64 VALU / HBM-write
  // work does not hide under fp64 MFMAs on this part.  Off by default; kept as an option for the record.
  int opt_rhs_overlap = 0;
  // comm
  ncclComm_t comm = nullptr;
  int nranks = 1, rank = 0;
  // host path: pinned staging in, pinned landing zone out; results leave the device chunk by chunk on their own stream
  PinBuf pin_in, pin_out;
  hipStream_t stream_d2h = nullptr;
  hipEvent_t ev_d2h = nullptr;  // recorded on stream_d2h after the last result copy of a predict
  bool results_on_host = false;
  long out_off = 0;  // where this handle's (unmasked) slab starts in the caller's arrays (device groups)
  // single-process device group (mik_set_devices): this handle is device 0 of the group and owns the others
  std::vector<mik_handle*> kids;
  bool is_kid = false;
  bool alias_ok = false;       // "alias_devices": a group may put several logical devices on one physical GPU (1-GPU test boxes)
  int opt_exchange = 0;        // "exchange": 0 = auto (RCCL broadcast, else peer copies), 1 = RCCL, 2 = peer copies, 3 = every device factors
  int exchange_used = 0;       // what the last mik_factor did (same codes; 0 = single device)
  double exchange_ms = 0.0;
  std::string exchange_note;
  std::vector<std::vector<hipStream_t>> xstreams; // xstreams[i][k]: stream on device i for the copy to device k (peer exchange)
  std::vector<hipEvent_t> xevents;
  // the exchange in flight (see "the factor exchange of a device group" below)
  std::shared_ptr<struct XchgJob> xjob;
  hipStream_t xstream = nullptr;  // this member's exchange stream (RCCL broadcast, checksums)
  DevBuf xsum;                 // 4 x u64: checksums of T and c after an exchange
  std::chrono::steady_clock::time_point xchg_t0;
  double exchange_wait_ms = 0.0;  // of exchange_ms, what a caller really waited for (the rest overlapped the leader's prediction)
  int exchange_fallbacks = 0, rccl_ranks = 0;
  int rccl_failures = 0;  // consecutive RCCL exchanges of this handle that FAILED (returned an error; a stall disables RCCL process-wide)
  int opt_async_exchange = 1;  // "async_exchange": mik_factor returns after the leader's K1 + K2; the exchange is joined by the next call
  double rccl_init_limit = 120.0, rccl_bcast_limit = 30.0, peer_limit = 30.0;  // seconds; MIK_RCCL_INIT_TIMEOUT, MIK_RCCL_BCAST_TIMEOUT, MIK_PEER_TIMEOUT
};

static int get_events(mik_handle* h, size_t n) {
  while (h->evpool.size() < n) {
    hipEvent_t e;
    HIPC(hipEventCreate(&e));
    h->evpool.push_back(e);
  }
  return MIK_OK;
}

static double host_vario(const Vario& v, double d) {
  switch (v.model) {
    case 0: return v.p0 * d + v.p1;
    case 1: return v.p0 * std::pow(d, v.p1) + v.p2;
    case 2: return v.p0 * (1.0 - std::exp(-(d * d) / v.c0)) + v.p2;
    case 3: return d <= v.p1 ? v.p0 * ((3.0 * d) / (2.0 * v.p1) - (d * d * d) / (2.0 * v.p1 * v.p1 * v.p1)) + v.p2 : v.p0 + v.p2;
    case 4: return v.p0 * (1.0 - std::exp(-d / v.c0)) + v.p2;
    default: {
      double q = d / v.c0;
      return v.p0 * (1.0 - (1.0 - q) * std::exp(-q)) + v.p2;
    }
  }
}

#define DISPATCH_MODEL_NDIM(model, ndim, KERNEL, grid, block, stream, args)                                 \
  do {                                                                                                      \
    if ((ndim) == 1) { /* geographic lon/lat */                                                             \
      switch (model) {                                                                                      \
        case 0: hipLaunchKernelGGL((KERNEL<0, 1>), grid, block, 0, stream, args); break;                    \
        case 1: hipLaunchKernelGGL((KERNEL<1, 1>), grid, block, 0, stream, args); break;                    \
        case 2: hipLaunchKernelGGL((KERNEL<2, 1>), grid, block, 0, stream, args); break;                    \
        case 3: hipLaunchKernelGGL((KERNEL<3, 1>), grid, block, 0, stream, args); break;                    \
        case 4: hipLaunchKernelGGL((KERNEL<4, 1>), grid, block, 0, stream, args); break;                    \
        default: hipLaunchKernelGGL((KERNEL<5, 1>), grid, block, 0, stream, args); break;                   \
      }                                                                                                     \
    } else if ((ndim) == 3) {                                                                                      \
      switch (model) {                                                                                      \
        case 0: hipLaunchKernelGGL((KERNEL<0, 3>), grid, block, 0, stream, args); break;                    \
        case 1: hipLaunchKernelGGL((KERNEL<1, 3>), grid, block, 0, stream, args); break;                    \
        case 2: hipLaunchKernelGGL((KERNEL<2, 3>), grid, block, 0, stream, args); break;                    \
        case 3: hipLaunchKernelGGL((KERNEL<3, 3>), grid, block, 0, stream, args); break;                    \
        case 4: hipLaunchKernelGGL((KERNEL<4, 3>), grid, block, 0, stream, args); break;                    \
        default: hipLaunchKernelGGL((KERNEL<5, 3>), grid, block, 0, stream, args); break;                   \
      }                                                                                                     \
    } else {                                                                                                \
      switch (model) {                                                                                      \
        case 0: hipLaunchKernelGGL((KERNEL<0, 2>), grid, block, 0, stream, args); break;                    \
        case 1: hipLaunchKernelGGL((KERNEL<1, 2>), grid, block, 0, stream, args); break;                    \
        case 2: hipLaunchKernelGGL((KERNEL<2, 2>), grid, block, 0, stream, args); break;                    \
        case 3: hipLaunchKernelGGL((KERNEL<3, 2>), grid, block, 0, stream, args); break;                    \
        case 4: hipLaunchKernelGGL((KERNEL<4, 2>), grid, block, 0, stream, args); break;                    \
        default: hipLaunchKernelGGL((KERNEL<5, 2>), grid, block, 0, stream, args); break;                   \
      }                                                                                                     \
    }                                                                                                       \
  } while (0)

template <int GY, int GX, int RI, int CJ>
static int launch_mw_solve(mik_handle* h, const MwArgs& a, long pc, bool piv) {
  constexpr int T = GY * GX, PPB = 256 / T, CJP = (CJ + 1) & ~1;
  const int nb = a.K + 1;
  if (nb > GY * RI || nb + 1 > GX * CJ) return fail(MIK_EINVAL, "moving-window solve class too small for this window");
  const size_t per = (2 * ((size_t)GX * CJP + (size_t)GY * RI) + 16 + 5 * (size_t)nb + (2 * (size_t)nb + 1) / 2 + 1) & ~(size_t)1;
  const size_t lds = sizeof(double) * per * PPB;
  const dim3 grid((unsigned)((pc + PPB - 1) / PPB));
  if (piv) {
    HIPC(hipFuncSetAttribute((const void*)k_mw_solve<GY, GX, RI, CJ, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mw_solve<GY, GX, RI, CJ, true>), grid, dim3(256), lds, h->stream, a);
  } else {
    HIPC(hipFuncSetAttribute((const void*)k_mw_solve<GY, GX, RI, CJ, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_mw_solve<GY, GX, RI, CJ, false>), grid, dim3(256), lds, h->stream, a);
  }
  return MIK_OK;
}

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

// thread-grid / register-tile classes of k_mw_solve, {GY, GX, RI, CJ} covers nb <= GY*RI and nb + 1 <= GX*CJ.  Measured
// on MI355X (scripts/mw_classes.py history in DESIGN.md): the classes whose tile fits the VGPR file without AGPR spills
// win, and among those the one with the fewest threads per point.
static int dispatch_mw_solve(mik_handle* h, const MwArgs& a, long pc, bool piv) {
  const int nb = a.K + 1;
  if (nb <= 16) return launch_mw_solve<4, 4, 4, 5>(h, a, pc, piv);   // 16 threads per point
  if (nb <= 32) return launch_mw_solve<8, 8, 4, 5>(h, a, pc, piv);   // 64
  if (nb <= 48) return launch_mw_solve<8, 8, 6, 7>(h, a, pc, piv);   // 64
  if (nb <= 64) return launch_mw_solve<8, 8, 8, 9>(h, a, pc, piv);   // 64
  if (nb <= 96) return launch_mw_solve<16, 16, 6, 7>(h, a, pc, piv); // 256
  return launch_mw_solve<16, 16, 8, 9>(h, a, pc, piv);               // 256, nb <= 128
}

template <int G, int RI>
static int launch_mw_chol(mik_handle* h, const MwArgs& a, long pc) {
  constexpr int T = G * G, NT = T < 256 ? 256 : T, PPB = NT / T, NB = G * RI;
  if (a.K > NB) return fail(MIK_EINVAL, "moving-window LDL^T class too small for this window");
  const size_t lds = sizeof(double) * (size_t)(2 * (NB + 4) + 9 * NB) * PPB;
  const dim3 grid((unsigned)((pc + PPB - 1) / PPB));
  // the variogram model as a compile-time constant where the problem allows it (Euclidean coordinates; the four models whose
  // shifted station block is positive definite and cheap): the set-up code of the kernel shrinks 20-fold (mw_entry_t)
#define MWC_LAUNCH(MODEL)                                                                                                   \
  do {                                                                                                                      \
    HIPC(hipFuncSetAttribute((const void*)k_mw_chol<G, RI, MODEL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
    hipLaunchKernelGGL((k_mw_chol<G, RI, MODEL>), grid, dim3(NT), lds, h->stream, a);                                       \
  } while (0)
  const int sm = (a.mode == 1 || !h->opt_mw_static) ? -1 : a.v.model;
  switch (sm) {
    case 0: MWC_LAUNCH(0); break;
    case 2: MWC_LAUNCH(2); break;
    case 3: MWC_LAUNCH(3); break;
    case 4: MWC_LAUNCH(4); break;
    default: MWC_LAUNCH(-1); break;
  }
#undef MWC_LAUNCH
  return MIK_OK;
}

// thread-grid / register-tile classes of k_mw_chol: {G, RI} covers K <= G * RI
#define MIK_MW_CHOL_KMAX 256
static int dispatch_mw_chol(mik_handle* h, const MwArgs& a, long pc) {
  const int K = a.K;
  if (h->opt_mw_class) {  // "mw_class" = 100 G + RI: a class forced for A/B runs (scripts/mw_classes.py)
    switch (h->opt_mw_class) {
#define MWC(G, RI) case 100 * G + RI: return launch_mw_chol<G, RI>(h, a, pc);
      // (round 4: the classes that lost every A/B of rounds 2-3 -- {8,14}, {8,16}, {16,4..6}, {16,15}, {16,16}, {32,5..7} -- are no longer
      // built: each was a 100 000-instruction kernel; profiles/r03_mw_classes_*.txt keep their measurements)
      MWC(4, 4) MWC(4, 6) MWC(4, 8) MWC(4, 10) MWC(4, 13) MWC(8, 4) MWC(8, 6) MWC(8, 8) MWC(8, 10) MWC(8, 11) MWC(8, 12) MWC(8, 13)
      MWC(16, 7) MWC(16, 8) MWC(16, 9) MWC(16, 10) MWC(16, 11) MWC(16, 12) MWC(16, 13) MWC(16, 14)
      MWC(32, 8)
#undef MWC
      default: return fail(MIK_EINVAL, "mw_class: no such LDL^T class");
    }
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

extern "C" {

const char* mik_last_error(void) { return g_err.c_str(); }

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
  env = getenv("MIK_ENGINE");
  if (env) h->opt_engine = (!strcmp(env, "valu") || !strcmp(env, "1")) ? 1 : 0;
  env = getenv("MIK_WAVES");
  if (env && (atoi(env) == 4 || atoi(env) == 8)) h->opt_waves = atoi(env);
  env = getenv("MIK_CHUNK");
  if (env && atol(env) >= 128) h->opt_chunk = (atol(env) / 128) * 128;
  env = getenv("MIK_SYMSWEEP");
  if (env) h->opt_symsweep = atoi(env) < 0 ? -1 : atoi(env) ? 1 : 0;
  env = getenv("MIK_PAIRS");
  if (env) h->opt_pairs = atoi(env) ? 1 : 0;
  env = getenv("MIK_SPARSE");
  if (env && atoi(env) >= -1 && atoi(env) <= 2) h->opt_sparse = atoi(env);
  env = getenv("MIK_SORT_POINTS");
  if (env && atoi(env) >= -1 && atoi(env) <= 1) h->opt_sort_points = atoi(env);
  env = getenv("MIK_SPARSE_GROUP");
  if (env && atoi(env) >= 1 && atoi(env) <= 16) h->opt_sparse_group = atoi(env);
  env = getenv("MIK_SPARSE_EPILOGUE");
  if (env) h->opt_sparse_epi = atoi(env) ? 1 : 0;
  env = getenv("MIK_SPARSE_ROWS");
  if (env && (atoi(env) == -1 || atoi(env) == 16 || atoi(env) == 128)) h->opt_sparse_rows = atoi(env);
  env = getenv("MIK_UPDATE_ATOMIC");
  if (env) h->opt_update_atomic = atoi(env) ? 1 : 0;
  env = getenv("MIK_UPDATE_REV");
  if (env) h->opt_update_rev = atoi(env) < 0 ? -1 : atoi(env) ? 1 : 0;
  env = getenv("MIK_UPDATE_MAP");
  if (env) h->opt_update_map = atoi(env);
  env = getenv("MIK_PANEL_STREAM");
  if (env) h->opt_panel_stream = atoi(env) < 0 ? -1 : atoi(env) ? 1 : 0;
  env = getenv("MIK_TRI");
  if (env) h->opt_tri = atoi(env) ? 1 : 0;
  env = getenv("MIK_PREFETCH");
  if (env) h->opt_prefetch = atoi(env) ? 1 : 0;
  env = getenv("MIK_EXCHANGE");
  if (env) h->opt_exchange = !strcmp(env, "rccl") ? 1 : !strcmp(env, "peer") ? 2 : !strcmp(env, "redundant") ? 3 : 0;
  env = getenv("MIK_ALIAS_DEVICES");
  if (env) h->alias_ok = atoi(env) != 0;
  env = getenv("MIK_EARLY_DIAG");
  if (env) h->opt_early_diag = atoi(env) < 0 ? -1 : atoi(env);
  env = getenv("MIK_UPDATE_WAVES");
  if (env && (atoi(env) == 4 || atoi(env) == 8)) h->opt_update_waves = atoi(env);
  env = getenv("MIK_PANEL_ROWS");
  if (env && (atoi(env) == 32 || atoi(env) == 64 || atoi(env) == 128)) h->opt_panel_rows = atoi(env);
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
                    &h->Cold2, &h->Cnew2, &h->Rt2, &h->Dinv2, &h->DinvT2, &h->Dinv3, &h->DinvT3, &h->tilemap, &h->Dnext, &h->Dcopy, &h->Cb, &h->Rb, &h->grid.gx, &h->grid.gy, &h->grid.gz, &h->grid.orig,
                    &h->grid.cstart,
                    &h->px, &h->py, &h->pz, &h->grid_axes, &h->grid_idx, &h->Averify, &h->vbuf, &h->extra_rows, &h->z, &h->ss, &h->Bt, &h->Bt2, &h->part, &h->mw_idx, &h->mw_dist, &h->stat_S, &h->stat_x, &h->stat_out, &h->queue,
                    &h->xs_s, &h->ys_s, &h->zs_s, &h->vals_s, &h->extra_cols_s, &h->sbox, &h->sp_cand, &h->sp_flags, &h->sp_klist, &h->sp_kcount,
                    &h->sp_nrows, &h->sp_rows, &h->sp_rstart, &h->sp_tiles, &h->sp_xoff, &h->sp_stats, &h->sp2_cand, &h->sp2_flags, &h->sp2_klist, &h->sp2_kcount,
                    &h->sp2_nrows, &h->sp2_rows, &h->sp2_rstart, &h->sp2_tiles, &h->sp2_xoff, &h->part2, &h->queue2, &h->dsc, &h->sp_recs, &h->sp2_recs, &h->ps_key[0], &h->ps_key[1], &h->ps_idx[0], &h->ps_idx[1], &h->ps_table, &h->ps_box, &h->ps_x, &h->ps_y, &h->ps_z, &h->ps_zs, &h->ps_sss};
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
    k->opt_factor = h->opt_factor, k->opt_sym = h->opt_sym, k->opt_engine = h->opt_engine, k->opt_waves = h->opt_waves;
    k->opt_chunk = h->opt_chunk, k->opt_symsweep = h->opt_symsweep, k->opt_diag = h->opt_diag, k->opt_update_waves = h->opt_update_waves, k->opt_panel_rows = h->opt_panel_rows, k->opt_panel_stream = h->opt_panel_stream, k->opt_update_map = h->opt_update_map, k->opt_update_rev = h->opt_update_rev, k->opt_update_atomic = h->opt_update_atomic, k->opt_lookahead = h->opt_lookahead, k->opt_gate = h->opt_gate, k->opt_fuse_chain = h->opt_fuse_chain, k->opt_early_diag = h->opt_early_diag, k->opt_pinv_fast = h->opt_pinv_fast, k->opt_rhs_overlap = h->opt_rhs_overlap, k->opt_verify = h->opt_verify, k->verify_tol_z = h->verify_tol_z, k->verify_tol_inv = h->verify_tol_inv;
    k->opt_mw_class = h->opt_mw_class, k->opt_mw_knn_bound = h->opt_mw_knn_bound, k->opt_mw_pivot = h->opt_mw_pivot, k->opt_mw_lds_cap = h->opt_mw_lds_cap, k->opt_pairs = h->opt_pairs, k->opt_tri = h->opt_tri, k->opt_prefetch = h->opt_prefetch, k->opt_symmetrize = h->opt_symmetrize, k->opt_mw_solver = h->opt_mw_solver;
    k->opt_sparse = h->opt_sparse, k->opt_sparse_lanes = h->opt_sparse_lanes, k->opt_sparse_rows = h->opt_sparse_rows, k->opt_sparse_epi = h->opt_sparse_epi, k->opt_sparse_group = h->opt_sparse_group, k->opt_sort_points = h->opt_sort_points, k->opt_drift_eq = h->opt_drift_eq;
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
  } else if (!strcmp(key, "alias_devices")) {
    h->alias_ok = value != 0.0;
  } else if (!strcmp(key, "factor")) {
    if (value < 0 || value > 2) return fail(MIK_EINVAL, "factor must be 0 (auto), 1 (sweep) or 2 (pivoted)");
    h->opt_factor = (int)value;
  } else if (!strcmp(key, "symmetric")) {
    h->opt_sym = value != 0.0;
  } else if (!strcmp(key, "engine")) {
    if (value != 0.0 && value != 1.0) return fail(MIK_EINVAL, "engine must be 0 (mfma) or 1 (valu)");
    h->opt_engine = (int)value;
  } else if (!strcmp(key, "sparse")) {
    if (value != -1.0 && value != 0.0 && value != 1.0 && value != 2.0) return fail(MIK_EINVAL, "sparse must be -1 (auto), 0, 1 or 2");
    h->opt_sparse = (int)value;
  } else if (!strcmp(key, "sort_points")) {
    if (value != -1.0 && value != 0.0 && value != 1.0) return fail(MIK_EINVAL, "sort_points must be -1 (auto), 0 or 1");
    h->opt_sort_points = (int)value;
  } else if (!strcmp(key, "sparse_group")) {
    if (!(value >= 1.0 && value <= 16.0)) return fail(MIK_EINVAL, "sparse_group must be 1 .. 16");
    h->opt_sparse_group = (int)value;
  } else if (!strcmp(key, "sparse_epilogue")) {
    h->opt_sparse_epi = value != 0.0;
  } else if (!strcmp(key, "sparse_rows")) {
    if (value != -1.0 && value != 16.0 && value != 128.0) return fail(MIK_EINVAL, "sparse_rows must be -1 (auto), 16 or 128");
    h->opt_sparse_rows = (int)value;
  } else if (!strcmp(key, "sparse_lanes")) {
    if (value != 1.0 && value != 2.0) return fail(MIK_EINVAL, "sparse_lanes must be 1 or 2");
    h->opt_sparse_lanes = (int)value;
  } else if (!strcmp(key, "drift_eq")) {
    h->opt_drift_eq = value != 0.0;
  } else if (!strcmp(key, "pairs")) {
    h->opt_pairs = value != 0.0;
  } else if (!strcmp(key, "tri")) {
    h->opt_tri = value != 0.0;
  } else if (!strcmp(key, "prefetch")) {
    h->opt_prefetch = value != 0.0;
  } else if (!strcmp(key, "symmetrize")) {
    h->opt_symmetrize = value != 0.0;
  } else if (!strcmp(key, "waves")) {
    if (value != 4.0 && value != 8.0) return fail(MIK_EINVAL, "waves must be 4 or 8");
    h->opt_waves = (int)value;
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
  } else if (!strcmp(key, "fuse_chain")) {
    h->opt_fuse_chain = value != 0.0;
  } else if (!strcmp(key, "early_diag")) {
    h->opt_early_diag = value < 0.0 ? -1 : (int)value;  // 1 / 3 = on (streams ordered by events), 2 = with the one-block tile kernels, 4 / 5 = ordered by flags (both sides / update stream only)
  } else if (!strcmp(key, "gate")) {
    h->opt_gate = value < 0.0 ? -1 : (value != 0.0);
  } else if (!strcmp(key, "update_waves")) {
    if (value != 4.0 && value != 8.0) return fail(MIK_EINVAL, "update_waves must be 4 or 8");
    h->opt_update_waves = (int)value;
  } else if (!strcmp(key, "update_atomic")) {
    h->opt_update_atomic = value != 0.0;
  } else if (!strcmp(key, "update_rev")) {
    h->opt_update_rev = value < 0.0 ? -1 : value != 0.0 ? 1 : 0;
  } else if (!strcmp(key, "update_map")) {
    h->opt_update_map = (int)value;
  } else if (!strcmp(key, "panel_stream")) {
    h->opt_panel_stream = value < 0.0 ? -1 : (value != 0.0);
  } else if (!strcmp(key, "panel_rows")) {
    if (value != 32.0 && value != 64.0 && value != 128.0) return fail(MIK_EINVAL, "panel_rows must be 32, 64 or 128");
    h->opt_panel_rows = (int)value;
  } else if (!strcmp(key, "diag")) {
    h->opt_diag = (int)value;
  } else if (!strcmp(key, "lookahead")) {
    h->opt_lookahead = value < 0.0 ? -1 : (value != 0.0);
  } else if (!strcmp(key, "mw_solver")) {
    if (value != 0.0 && value != 1.0) return fail(MIK_EINVAL, "mw_solver must be 0 (LDL^T) or 1 (Gauss-Jordan)");
    h->opt_mw_solver = (int)value;
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

int mik_station_order(const mik_problem* p, int32_t* order_out) {
  if (!p || !order_out) return fail(MIK_EINVAL, "mik_station_order: NULL argument");
  if ((p->ndim != 2 && p->ndim != 3) || p->n < 1 || !p->xs || !p->ys || (p->ndim == 3 && !p->zs))
    return fail(MIK_EINVAL, "mik_station_order: station arrays missing");
  std::vector<int> order;
  hilbert_order(p->ndim, p->n, p->xs, p->ys, p->zs, order);
  for (long i = 0; i < p->n; ++i) order_out[i] = order[(size_t)i];
  return MIK_OK;
}

static int upload_sorted_stations(mik_handle* h, const mik_problem* p) {
  const long N = h->N;
  if (!(h->stations_same && (long)h->sort_perm.size() == N)) hilbert_order(h->ndim, N, p->xs, p->ys, p->zs, h->sort_perm);
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
  const int nIblk = h->Mp / 16;
  std::vector<double> box((size_t)nIblk * 6);
  const double* c[3] = {p->xs, p->ys, p->zs};
  for (int b = 0; b < nIblk; ++b) {
    double* q = box.data() + (size_t)b * 6;
    for (int d = 0; d < 3; ++d) {
      q[d] = d < h->ndim ? 1e300 : 0.0;
      q[3 + d] = d < h->ndim ? -1e300 : 0.0;
    }
    for (long i = (long)b * 16; i < std::min<long>(N, (long)(b + 1) * 16); ++i)
      for (int d = 0; d < h->ndim; ++d) {
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

// the order the next factor will be in: "sparse" -1 (auto) / 1 / 2 = Hilbert-curve order wherever the problem allows it
// (measured even at N = 100: the four small list kernels per launch cost less than the dense tiles they save)
static bool want_sorted(const mik_handle* h) { return h->sort_ok && h->opt_sparse != 0; }

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
  // the range-aware contraction (k_contract_sp).  Not with a pseudo-inverse (A+ u != e_last), a caller's inverse (its order is the
  // caller's) or geographic coordinates (lon / lat boxes are not distance boxes).
  h->sort_ok = h->model == MIK_MODEL_SPHERICAL && !h->geo && !h->pinv && !h->host_inv && h->Mp / 16 <= MIK_SP_MAXK16 &&
               std::isfinite(v.p1) && v.p1 > 0.0 && std::isfinite(v.p0 + v.p2);
  h->factor_sorted = false;
  if (h->sort_ok) MIKC(upload_sorted_stations(h, p));
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

#define DISPATCH_NDIM_FIXED(MODEL, ndim, KERNEL, grid, block, stream, args)                    \
  do {                                                                                        \
    if ((ndim) == 1) hipLaunchKernelGGL((KERNEL<MODEL, 1>), grid, block, 0, stream, args);    \
    else if ((ndim) == 3) hipLaunchKernelGGL((KERNEL<MODEL, 3>), grid, block, 0, stream, args); \
    else hipLaunchKernelGGL((KERNEL<MODEL, 2>), grid, block, 0, stream, args);                \
  } while (0)

// custom variogram: bring `rows` rows of a device array (row length ld, `cols` meaningful columns) to the host, let the
// caller's function turn distances into gamma in place, send them back
static int custom_roundtrip(mik_handle* h, double* dev, long rows, long cols, long ld) {
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

static int launch_assemble(mik_handle* h, double shift, double* dst = nullptr, bool sorted = false, bool eq = false) {
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

static int ensure_factor_buffers(mik_handle* h) {
  const size_t Mp = h->Mp;
  MIKC(h->T.ensure(sizeof(double) * Mp * Mp));
  MIKC(h->cvec.ensure(sizeof(double) * Mp));
  return MIK_OK;
}

static void launch_diag_inv(mik_handle* h, hipStream_t st, const double* T, long ld, int k0, int nspd, double* dinv, double* dinvT,
                            bool own_cu = false) {
  int* flag = h->flag.as<int>();
  if (h->opt_diag == 4) {  // blocked (round 3): 86 KB of LDS of its own, padded like the others' when it wants the CU to itself
    const int lds = own_cu ? 100 * 1024 : (int)(sizeof(double) * MIK_DIAGB_LDS_DOUBLES);
    (void)hipFuncSetAttribute((const void*)k_diag_inv_b<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(k_diag_inv_b<0>, dim3(1), dim3(256), lds, st, T, ld, k0, nspd, dinv, dinvT, flag);
    return;
  }
  if (own_cu && h->opt_diag == 1) {  // keep trailing-update blocks (64 KB of LDS each) off this block's CU: see k_gate
    constexpr int pad = 100 * 1024;
    // per launch: the attribute belongs to the function object of the CURRENT device (device groups factor on several)
    (void)hipFuncSetAttribute((const void*)k_diag_inv_t<16, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, pad);
    hipLaunchKernelGGL((k_diag_inv_t<16, 16>), dim3(1), dim3(256), pad, st, T, ld, k0, nspd, dinv, dinvT, flag);
    return;
  }
  switch (h->opt_diag) {
    case 1: hipLaunchKernelGGL((k_diag_inv_t<16, 16>), dim3(1), dim3(256), 0, st, T, ld, k0, nspd, dinv, dinvT, flag); break;
    case 2: hipLaunchKernelGGL((k_diag_inv_t<16, 32>), dim3(1), dim3(512), 0, st, T, ld, k0, nspd, dinv, dinvT, flag); break;
    case 3: hipLaunchKernelGGL((k_diag_inv_t<32, 32>), dim3(1), dim3(1024), 0, st, T, ld, k0, nspd, dinv, dinvT, flag); break;
    default: hipLaunchKernelGGL(k_diag_inv, dim3(1), dim3(1024), 0, st, T, ld, k0, nspd, dinv, dinvT, flag); break;
  }
}

// k_update's tilemap: the tiles of the (upper triangle of the) block grid, super-block by super-block (sb x sb tiles, rows of
// super-blocks, inside one column by column), invalid positions skipped -- a plain permutation of the kernel's own enumeration,
// so xcd_tile() still hands every XCD an equal, contiguous share.  Cached per (nblk, sym, sb).
static int update_tile_map(mik_handle* h, int nblk, bool sym, int sb) {
  if (h->tilemap_key[0] == nblk && h->tilemap_key[1] == (int)sym && h->tilemap_key[2] == sb) return MIK_OK;
  std::vector<int2> map;
  map.reserve(sym ? (size_t)nblk * (nblk + 1) / 2 : (size_t)nblk * nblk);
  const int ns = (nblk + sb - 1) / sb;
  for (int I = 0; I < ns; ++I)
    for (int J = sym ? I : 0; J < ns; ++J)
      for (int dj = 0; dj < sb; ++dj)
        for (int di = 0; di < sb; ++di) {
          const int i = I * sb + di, j = J * sb + dj;
          if (i >= nblk || j >= nblk || (sym && i > j)) continue;
          map.push_back(make_int2(i, j));
        }
  MIKC(h->tilemap.ensure(sizeof(int2) * map.size()));
  HIPC(hipMemcpyAsync(h->tilemap.p, map.data(), sizeof(int2) * map.size(), hipMemcpyHostToDevice, h->stream));
  HIPC(hipStreamSynchronize(h->stream));  // (map is a local)
  h->tilemap_key[0] = nblk, h->tilemap_key[1] = (int)sym, h->tilemap_key[2] = sb;
  return MIK_OK;
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
  const bool upd8 = h->opt_update_waves == 8;
  const int uatomic = (pivoted ? 0 : h->opt_update_atomic) | ((h->opt_update_rev < 0 ? nblk >= 45 : h->opt_update_rev != 0) ? 2 : 0);  // bit 0: plain tiles of the trailing update as
                                                                                           // fp64 atomic adds, bit 1: odd steps backwards (k_update)
  // tile order of the trailing update: optionally n x n super-blocks (k_update's tilemap)
  const int2* tmap = nullptr;
  if (!pivoted && h->opt_update_map > 1) {
    MIKC(update_tile_map(h, nblk, symsweep, h->opt_update_map));
    tmap = h->tilemap.as<int2>();
  }
  // the panel kernel over all Mp rows: 32 * NAI rows per block (k_panel)
#define PANEL(STREAM, ...)                                                                                                   \
  do {                                                                                                                       \
    if (h->opt_panel_rows == 32) hipLaunchKernelGGL((k_panel<1>), dim3(4 * nblk), dim3(256), 0, STREAM, __VA_ARGS__);        \
    else if (h->opt_panel_rows == 64) hipLaunchKernelGGL((k_panel<2>), dim3(2 * nblk), dim3(256), 0, STREAM, __VA_ARGS__);   \
    else hipLaunchKernelGGL((k_panel<4>), dim3(nblk), dim3(256), 0, STREAM, __VA_ARGS__);                                    \
  } while (0)
#define UPDK(SYMV, GRID, STREAM, ...)                                                                                       \
  do {                                                                                                                      \
    if (upd8) hipLaunchKernelGGL((k_update<SYMV, 2>), GRID, dim3(512), 0, STREAM, __VA_ARGS__, tmap, uatomic);              \
    else hipLaunchKernelGGL((k_update<SYMV, 4>), GRID, dim3(256), 0, STREAM, __VA_ARGS__, tmap, uatomic);                   \
  } while (0)
#define UPDX(GRID, STREAM, CO, CN, R, D, PART, COL, POUT, DCOPY)                                                             \
  do {                                                                                                                       \
    if (symsweep)                                                                                                            \
      UPDK(true, GRID, STREAM, T, ld, nblk, kb, (const double*)(CO), (const double*)(CN),                                    \
           (const double*)(R), (const double*)(D), PART, COL, POUT, DCOPY, (int*)nullptr);                                   \
    else                                                                                                                     \
      UPDK(false, GRID, STREAM, T, ld, nblk, kb, (const double*)(CO), (const double*)(CN),                                   \
           (const double*)(R), (const double*)(D), PART, COL, POUT, DCOPY, (int*)nullptr);                                   \
  } while (0)
#define UPD(GRID, STREAM, CO, CN, R, D, PART, COL, POUT) UPDX(GRID, STREAM, CO, CN, R, D, PART, COL, POUT, (double*)nullptr)
  const bool early_ok = h->opt_early_diag != 0;
  const bool lookahead = h->opt_lookahead < 0 ? nblk >= (early_ok ? 3 : 24) : h->opt_lookahead != 0;
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
      PANEL(st, (const double*)cold[set], 128L, (const double*)dinvT[set], -1.0, cnew[set], rt[set], k0, 0, 0, (int*)nullptr, -1, -1);
    };
    const bool early = h->opt_early_diag < 0 ? true : h->opt_early_diag != 0;
    panel_chain(h->stream, 0, 0, false);
    if (early) {
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
      MIKC(h->Cb.ensure(sizeof(double) * 128 * 128));
      MIKC(h->Rb.ensure(sizeof(double) * 128 * 128));
      double* dnext = h->Dnext.as<double>();
      double* dcopy[2] = {h->Dcopy.as<double>(), h->Dcopy.as<double>() + 128 * 128};  // [kb & 1] is read by step kb's chain
      double* cb = h->Cb.as<double>();
      double* rb = h->Rb.as<double>();
      HIPC(hipMemcpy2DAsync(dcopy[0], sizeof(double) * 128, T + 128L * ld + 128, sizeof(double) * ld, sizeof(double) * 128, 128,
                            hipMemcpyDeviceToDevice, h->stream));  // tile (1, 1) as assembled: the second stream never reads T
      HIPC(hipEventRecord(h->la_events[0], h->stream));  // "update -1": the first panel set and diagonal inverse are there
      HIPC(hipStreamWaitEvent(h->stream2, h->la_events[0], 0));
      // The two streams are ordered by events (default).  A satisfied hipStreamWaitEvent still costs ~12 us of barrier-packet
      // latency per step and stream, so two opt-in modes order them through the flag buffer (MIK_F_*) instead:
      //   early_diag = 5: update stream <- "diagonal inverse kb finished" by a flag the inverse releases, polled inside k_panel
      //                   (N=5000: 5.35 -> 5.0 ms, N=8000: 15.1 -> 14.6 ms);
      //   early_diag = 4: also chain stream <- "update kb-1 finished" by a count of finished blocks behind k_wait_ge -- every
      //                   block's release writes its XCD's L2 back: good for small sweeps only (N=2000: 1.86 -> 1.76 ms; N=8000:
      //                   15.1 -> 20.8 ms).
      // They are NOT the default because a kernel that waits for a kernel of another stream needs both to be able to run
      // concurrently: under tools that serialise dispatches (rocprofv3 --pmc, debuggers) the wait runs out (bounded: an error,
      // not a hang).  (Also not beyond 128 block columns: k_panel's waiting blocks hold LDS, and with two of them on every CU
      // a diagonal inverse that has not been placed yet could never start.)  What IS folded into k_panel in every mode is
      // k_gate's poll: a hint with a bounded wait, harmless when serialised.
      const bool flags_s1 = (h->opt_early_diag == 4 || h->opt_early_diag == 5) && nblk <= 128;
      const bool flags_s2 = flags_s1 && h->opt_early_diag == 4;
      int* fl = h->flag.as<int>();
      const bool pstream = !flags_s1 && h->opt_early_diag != 2 && (h->opt_panel_stream < 0 ? nblk >= 24 : h->opt_panel_stream != 0);
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
            PANEL(s3, (const double*)cold[set], 128L, (const double*)dvT[d3], -1.0, cnew[set], rt[set], k0, 0, 0, (int*)nullptr, -1, -1);
            HIPC(hipEventRecord(evP(kb), s3));
          }
          if (kb + 1 < nblk) {  // s3: column part of update kb
            if (kb > 0) HIPC(hipStreamWaitEvent(s3, evR(kb - 1), 0));
            if (symsweep)
              UPDK(true, dim3(nblk + 1), s3, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set], (const double*)rt[set],
                   (const double*)dv[d3], 3, kb + 1, cold[set ^ 1], dcopy[set ^ 1], (int*)nullptr);
            else
              UPDK(false, dim3(nblk + 1), s3, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set], (const double*)rt[set],
                   (const double*)dv[d3], 3, kb + 1, cold[set ^ 1], dcopy[set ^ 1], (int*)nullptr);
            HIPC(hipEventRecord(evC(kb), s3));
          }
          if (kb > 0) HIPC(hipStreamWaitEvent(s1, evP(kb), 0));  // s1: the rest (last step: everything)
          const int part = kb + 1 < nblk ? 4 : 0, colarg = kb + 1 < nblk ? kb + 1 : -2;
          if (symsweep)
            UPDK(true, dim3(ug), s1, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set], (const double*)rt[set],
                 (const double*)dv[d3], part, colarg, (double*)nullptr, (double*)nullptr, (int*)nullptr);
          else
            UPDK(false, dim3(ug), s1, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set], (const double*)rt[set],
                 (const double*)dv[d3], part, colarg, (double*)nullptr, (double*)nullptr, (int*)nullptr);
          if (kb + 1 < nblk) HIPC(hipEventRecord(evR(kb), s1));
        }
      } else
      for (int kb = 0; kb < nblk; ++kb) {
        const int set = kb & 1, k0 = kb * 128, k1 = k0 + 128;
        if (kb + 1 < nblk) {
          hipStream_t s2 = h->stream2;
          if (kb > 0) {  // update kb-1 has left cold[set], dcopy[set]
            if (flags_s2) hipLaunchKernelGGL(k_wait_ge, dim3(1), dim3(1), 0, s2, fl, MIK_F_UCNT + kb - 1, (int)ug);
            else HIPC(hipStreamWaitEvent(s2, h->la_events[2 * kb], 0));
          }
          if (h->opt_early_diag == 2) {  // the library's one-block tile kernels (22 us each: a CU's MFMA rate), kept for comparison
            hipLaunchKernelGGL((k_panel<4>), dim3(1), dim3(256), 0, s2, (const double*)cold[set], 128L, (const double*)dinvT[set], -1.0, cb, rb,
                               k0, kb + 1, k1, (int*)nullptr, -1, -1);
            hipLaunchKernelGGL(k_next_diag, dim3(1), dim3(256), 0, s2, (const double*)dcopy[set], 128L,
                               (const double*)(cold[set] + (long)k1 * 128), (const double*)rb, dnext);
          } else {  // the same accumulation streams, one per wavefront, over 64 blocks
            hipLaunchKernelGGL(k_gemm128<0>, dim3(64), dim3(256), 0, s2, (const double*)(cold[set] + (long)k1 * 128), (const double*)dinvT[set],
                               -1.0, (const double*)nullptr, 0L, rb);
            hipLaunchKernelGGL(k_gemm128<1>, dim3(64), dim3(256), 0, s2, (const double*)(cold[set] + (long)k1 * 128), (const double*)rb, 0.0,
                               (const double*)dcopy[set], 128L, dnext);
          }
          // the diagonal-inverse kernels address T[(k0 + r) * ld + k0 + c]: hand them the 128 x 128 copy under that indexing
          const double* dview = (const double*)((uintptr_t)dnext - sizeof(double) * ((size_t)k1 * 128 + (size_t)k1));
          launch_diag_inv(h, s2, dview, 128L, k1, nspd, dinv[set ^ 1], dinvT[set ^ 1], gate);
          if (!flags_s1) HIPC(hipEventRecord(h->la_events[2 * kb + 1], s2));
        }
        const bool gate_here = gate && kb + 1 < nblk;
        if (kb > 0) {
          if (!flags_s1) HIPC(hipStreamWaitEvent(h->stream, h->la_events[2 * kb - 1], 0));  // diagonal inverse kb
          // (the per-wavefront form of k_gemm128 for ALL panel rows was tried here: 30 us against 26 us -- its strided fragment
          // loads do not coalesce -- and its 640 blocks delay the chain's 64)
          // k_panel, leaving, polls for diagonal inverse kb+1 to have started (the gate); with flags_s1 it first waits for
          // diagonal inverse kb itself
          PANEL(h->stream, (const double*)cold[set], 128L, (const double*)dinvT[set], -1.0, cnew[set], rt[set], k0, 0, 0, fl, flags_s1 ? kb : -1,
                gate_here ? kb + 1 : -1);
        }
        if (kb + 1 < nblk) {
          if (gate_here && kb == 0) hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, h->stream, (const int*)fl, kb + 1, 20000);
          if (symsweep)
            UPDK(true, dim3(ug), h->stream, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set],
                 (const double*)rt[set], (const double*)dinv[set], 0, kb + 1, cold[set ^ 1], dcopy[set ^ 1], flags_s2 ? fl + MIK_F_UCNT + kb : (int*)nullptr);
          else
            UPDK(false, dim3(ug), h->stream, T, ld, nblk, kb, (const double*)cold[set], (const double*)cnew[set],
                 (const double*)rt[set], (const double*)dinv[set], 0, kb + 1, cold[set ^ 1], dcopy[set ^ 1], flags_s2 ? fl + MIK_F_UCNT + kb : (int*)nullptr);
          if (!flags_s2) HIPC(hipEventRecord(h->la_events[2 * kb + 2], h->stream));
        } else {
          UPD(dim3(ug), h->stream, cold[set], cnew[set], rt[set], dinv[set], 0, -2, (double*)nullptr);
        }
      }
      if (flags_s1) {  // once per inverse: the second stream has drained before this one goes on (and before the next call's memset)
        HIPC(hipEventRecord(h->la_events[1], h->stream2));
        HIPC(hipStreamWaitEvent(h->stream, h->la_events[1], 0));
      }
    } else
    for (int kb = 0; kb < nblk; ++kb) {
      const int set = kb & 1;
      if (kb + 1 < nblk) {
        // the column update leaves the updated block column in the other panel set as well (its last reader, the rest of step
        // kb-1, is earlier on this very stream): the chain below starts with the diagonal inverse
        UPD(dim3(nblk), h->stream, cold[set], cnew[set], rt[set], dinv[set], 1, kb + 1, h->opt_fuse_chain ? cold[set ^ 1] : (double*)nullptr);
        HIPC(hipEventRecord(h->la_events[2 * kb], h->stream));
        HIPC(hipStreamWaitEvent(h->stream2, h->la_events[2 * kb], 0));
        panel_chain(h->stream2, kb + 1, set ^ 1, h->opt_fuse_chain != 0);
        HIPC(hipEventRecord(h->la_events[2 * kb + 1], h->stream2));
        if (gate)  // hold the big update back until the next diagonal inverse sits on a CU (see k_gate)
          hipLaunchKernelGGL(k_gate, dim3(1), dim3(1), 0, h->stream, (const int*)h->flag.as<int>(), kb + 1, 20000);
        UPD(dim3(ug), h->stream, cold[set], cnew[set], rt[set], dinv[set], 2, kb + 1, (double*)nullptr);
        HIPC(hipStreamWaitEvent(h->stream, h->la_events[2 * kb + 1], 0));
      } else {
        UPD(dim3(ug), h->stream, cold[set], cnew[set], rt[set], dinv[set], 0, 0, (double*)nullptr);
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
          pivoted ? (double*)nullptr : h->Rt.as<double>(), k0, 0, 0, (int*)nullptr, -1, -1);
    if (pivoted) {
      hipLaunchKernelGGL(k_transpose_rows, dim3(Mp / 64, 2), dim3(256), 0, h->stream, (const double*)T, ld, k0, Mp,
                         h->TKt.as<double>());
      PANEL(h->stream, (const double*)h->TKt.as<double>(), 128L, (const double*)h->Dinv.as<double>(), 1.0, h->Rt.as<double>(), (double*)nullptr, 0, 0, 0,
            (int*)nullptr, -1, -1);
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
  int flag = 0, lost = 0;
  HIPC(hipMemcpyAsync(&flag, h->flag.p, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipMemcpyAsync(&lost, h->flag.as<int>() + MIK_F_ERR, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  if (lost) return fail(MIK_EHIP, "block sweep: a cross-stream wait ran out (a producer kernel never finished)");
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

static int one_factor(mik_handle* h) {
  if (!h || !h->have_problem) return fail(MIK_ESTATE, "mik_factor: no problem set");
  HIPC(hipSetDevice(h->device));
  h->t_state = 0;
  h->have_factor = false;
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
struct XchgMember {
  int device = 0;
  double* T = nullptr;
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
    hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, d.xs, (const unsigned long long*)d.T, Mp * Mp, d.sum_dev);
    hipLaunchKernelGGL(k_checksum, dim3(4), dim3(256), 0, d.xs, (const unsigned long long*)d.cvec, Mp, d.sum_dev + 2);
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
    first_bad = g_rccl.Broadcast(d.T, d.T, j->Mp * j->Mp, ncclDouble, 0, (*comms)[i], d.xs);
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
  const size_t Mp = j->Mp, S = Mp * Mp;
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
    MIKC(copy_between(dk.T + off, dk.device, d0.T + off, d0.device, sizeof(double) * len, st));
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
      MIKC(copy_between(dk.T + off, dk.device, dq.T + off, dq.device, sizeof(double) * len, st));
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
  j->mem.resize(n);
  for (int i = 0; i < n; ++i) {
    mik_handle* d = member(h, i);
    j->mem[i].device = d->device;
    j->mem[i].T = d->T.as<double>();
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
    hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, h->stream, (const unsigned long long*)h->T.p, j->Mp * j->Mp, sd);
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
  h->exchange_ms = h->exchange_wait_ms = 0.0;
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

// Hilbert-curve order of the resident points inside every launch of `chunk` points (k_ps_*, mik_kernels.h): ps_idx[0][s] = index of
// the point at sorted position s.  On the handle's stream; two radix passes of 10-bit digits, all segments side by side.
static int sort_points(mik_handle* h, long chunk, long nchunks) {
  const long npt = h->npt;
  const int bits = ps_bits(h->ndim), bps = (int)((chunk + MIK_PS_TILE - 1) / MIK_PS_TILE);
  for (int q = 0; q < 2; ++q) {
    MIKC(h->ps_key[q].ensure(sizeof(unsigned) * (size_t)npt));
    MIKC(h->ps_idx[q].ensure(sizeof(unsigned) * (size_t)npt));
  }
  MIKC(h->ps_table.ensure(sizeof(unsigned) * (size_t)nchunks * (1u << MIK_PS_DB) * (size_t)bps));
  MIKC(h->ps_box.ensure(sizeof(double) * 4 * (size_t)nchunks));
  const double *px = h->px.as<double>(), *py = h->py.as<double>(), *pz = h->ndim == 3 ? h->pz.as<double>() : nullptr;
  hipStream_t st = h->stream;
  hipLaunchKernelGGL(k_ps_bbox, dim3((unsigned)nchunks), dim3(1024), 0, st, px, py, pz, npt, chunk, bits, h->ps_box.as<double>());
  hipLaunchKernelGGL(k_ps_keys, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, st, px, py, pz, npt, chunk, h->ndim, bits,
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

static int one_predict(mik_handle* h) {
  if (!h || !h->have_factor) return fail(MIK_ESTATE, "mik_predict: factor first");
  if (!h->have_points) return fail(MIK_ESTATE, "mik_predict: set points first");
  HIPC(hipSetDevice(h->device));
  const long npt = h->npt;
  const int Mp = h->Mp, nIblk = Mp / 128;
  h->tm.rhs_ms = h->tm.contract_ms = h->tm.predict_ms = 0.0;
  h->tm.contract_launches = 0;
  h->tm.contract_flops_executed = 0.0;
  h->tm.symmetric = h->opt_sym;
  h->tm.engine = h->opt_engine;
  h->tm.mw_kernel = 0;
  if (npt == 0) {
    h->have_results = true;
    return MIK_OK;
  }
  long chunk = std::min<long>(h->opt_chunk, ((npt + 127) / 128) * 128);
  if (h->model == MIK_MODEL_CUSTOM) chunk = std::min<long>(chunk, 16384);  // each chunk's distances visit the host
  // range-aware contraction (k_contract_sp): the factor is in Hilbert-curve station order and the variogram has compact support
  const bool sparse = h->factor_sorted && h->opt_sparse != 2 && h->opt_sparse != 0 && h->opt_engine == 0;
  if (sparse) chunk = std::min<long>(chunk, 131072);  // k_sp_tiles: at most 1024 point blocks per launch
  const int nK16 = Mp / 16;
  // tiles of gathered 16-row groups (k_contract_spg) wherever 32-bit LDS-DMA offsets reach every row of the inverse
  const bool gathered = sparse && h->opt_sparse_rows != 128 && (double)Mp * (double)Mp * 8.0 < 4294967296.0;
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
      MIKC(lane[L].flags->ensure(nTb * nK16));
      MIKC(lane[L].klist->ensure(sizeof(unsigned short) * nTb * nK16));
      MIKC(lane[L].kcount->ensure(sizeof(int) * nTb));
      MIKC(lane[L].nrows->ensure(sizeof(int) * nTb));
      if (gathered) {
        MIKC(lane[L].recs->ensure(32 * nTb * nIblk));  // ceil(nk / 8) <= nK16 / 8 = nIblk tiles per point block
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
      hipLaunchKernelGGL(k_sp_cand, dim3(palloc / 128), dim3(128), 0, ss, a.px, a.py, a.pz, nvalid, (const double*)h->sbox.as<double>(), nK16,
                         h->N / 16, (h->M + 15) / 16, std::max(h->v.p1, h->eps), ln.cand->as<unsigned char>(), a.perm, gathered ? 0 : 1);
      HIPC(hipMemsetAsync(ln.flags->p, 0, (size_t)(palloc / 128) * nK16, ss));
      HIPC(hipEventRecord(h->evpool[2 + 4 * c], ss));
      if (h->ndim == 3) hipLaunchKernelGGL((k_rhs<3, 3, true>), dim3(palloc / MIK_TP), dim3(256), 0, ss, a);
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
        hipLaunchKernelGGL(k_sp_lists_g, dim3(nTb), dim3(64), 0, sc, (const unsigned char*)ln.flags->as<unsigned char>(), nK16,
                           ln.klist->as<unsigned short>(), ln.kcount->as<int>(), ln.nrows->as<int>());
        hipLaunchKernelGGL(k_sp_tiles_g, dim3(1), dim3(1024), 0, sc, (const int*)ln.nrows->as<int>(), (const int*)ln.kcount->as<int>(),
                           (const unsigned short*)ln.klist->as<unsigned short>(), nK16, nTb, ln.recs->as<uint4>(), ln.xoff->as<int>(),
                           h->sp_stats.as<unsigned long long>() + 4 * c, h->opt_sparse_group);
      } else {
        hipLaunchKernelGGL(k_sp_lists, dim3(nTb), dim3(64), 0, sc, (const unsigned char*)ln.flags->as<unsigned char>(), nK16, nIblk,
                           ln.klist->as<unsigned short>(), ln.kcount->as<int>(), ln.rows->as<unsigned short>(),
                           ln.rstart->as<unsigned short>(), ln.nrows->as<int>());
        hipLaunchKernelGGL(k_sp_tiles, dim3(1), dim3(1024), 0, sc, (const int*)ln.nrows->as<int>(), (const int*)ln.kcount->as<int>(),
                           (const unsigned short*)ln.rstart->as<unsigned short>(), nIblk, nTb, ln.tiles->as<unsigned>(),
                           ln.xoff->as<int>(), h->sp_stats.as<unsigned long long>() + 4 * c);
      }
      HIPC(hipEventRecord(h->evpool[3 + 4 * nchunks + 2 * c], sc));
      HIPC(hipMemsetAsync(ln.queue->p, 0, 8 * sizeof(unsigned long long), sc));
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
      HIPC(hipEventRecord(e1, sc));
      if (gathered) {
        SpgArgs ga{};
        ga.Ainv = sa.Ainv;
        ga.lda = Mp;
        ga.Bt = sa.Bt;
        ga.ldb = Mp;
        ga.part = sa.part;
        ga.palloc = palloc;
        ga.nK16 = nK16;
        ga.klist = sa.klist;
        ga.recs = ln.recs->as<uint4>();
        ga.xoff = sa.xoff;
        ga.queue = sa.queue;
        if (h->opt_sparse_epi) hipLaunchKernelGGL((k_contract_spg<2, true>), dim3((unsigned)std::min<long>(2L * h->n_cu, (long)nTb * nIblk)), dim3(512), 0, sc, ga);
        else hipLaunchKernelGGL((k_contract_spg<2, false>), dim3((unsigned)std::min<long>(2L * h->n_cu, (long)nTb * nIblk)), dim3(512), 0, sc, ga);
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
      if (h->opt_engine == 1) {
        if (h->opt_sym) hipLaunchKernelGGL(k_contract_valu<true>, dim3(grid), dim3(256), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend);
        else hipLaunchKernelGGL(k_contract_valu<false>, dim3(grid), dim3(256), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend);
      } else {
        // persistent launch: 2 blocks per CU pop tiles from per-XCD sequences (8 counters, zeroed per launch)
        MIKC(h->queue.ensure(8 * sizeof(unsigned long long)));
        HIPC(hipMemsetAsync(h->queue.p, 0, 8 * sizeof(unsigned long long), sc));
        unsigned long long* qp = h->queue.as<unsigned long long>();
        const unsigned pgrid = (unsigned)std::min<long>(2L * h->n_cu, (long)sgrid);
        if (h->opt_waves == 8 && h->opt_sym && h->opt_pairs) {
          hipLaunchKernelGGL((k_contract<true, 2, true, true>), dim3(pgrid), dim3(512), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend, qp);
        } else if (h->opt_waves == 8) {
          if (h->opt_sym && h->opt_tri && h->opt_prefetch) hipLaunchKernelGGL((k_contract<true, 2, true, false, true, true>), dim3(pgrid), dim3(512), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend, qp);
          else if (h->opt_sym && h->opt_tri) hipLaunchKernelGGL((k_contract<true, 2, true, false, true>), dim3(pgrid), dim3(512), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend, qp);
          else if (h->opt_sym) hipLaunchKernelGGL((k_contract<true, 2>), dim3(pgrid), dim3(512), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend, qp);
          else hipLaunchKernelGGL((k_contract<false, 2>), dim3(pgrid), dim3(512), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend, qp);
        } else {
          if (h->opt_sym) hipLaunchKernelGGL((k_contract<true, 4>), dim3(pgrid), dim3(256), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend, qp);
          else hipLaunchKernelGGL((k_contract<false, 4>), dim3(pgrid), dim3(256), 0, sc, Ai, ldm, Bi, ldm, pp, palloc, nIblk, kend, qp);
        }
      }
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
    const bool tri = h->opt_engine != 1 && h->opt_waves == 8 && h->opt_sym && !h->opt_pairs && h->opt_tri;
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


static int one_predict_mw(mik_handle* h, int n_closest) {
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
  // with or without implicit partial pivoting (opt_mw_solver = 1, or when the model cannot promise a positive definite
  // station block; windows up to 127), LU with partial pivoting in HBM scratch (any window)
  const bool chol = !mw_piv && h->opt_mw_solver == 0 && K <= MIK_MW_CHOL_KMAX && h->opt_mw_class != 1;
  // beyond the register classes: blocked Cholesky of the shifted system (one block per point, panels of 64 in LDS, the matrix
  // in an L2-resident scratch slot); "mw_class" 1 forces it for smaller windows too (A/B runs)
  const bool cholb = !mw_piv && h->opt_mw_solver == 0 && !chol && K >= 8;
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
      MIKC(dispatch_mw_solve(h, a, pc, mw_piv));
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
  size_t pooled = 0;
  for (auto& e : g_pin_pool) pooled += e.second;
  if (g_pin_pool.size() < 6 && pooled + bytes <= (size_t)2 << 30) g_pin_pool.emplace_back((void*)z, bytes);
  else (void)hipHostFree(z);
}

int mik_get_timing(mik_handle* h, mik_timing* out) {
  if (!h || !out) return fail(MIK_EINVAL, "mik_get_timing: NULL argument");
  MIKC(join_exchange(h));  // the exchange figures are final only then
  *out = h->tm;
  out->n_devices = (int)h->kids.size() + 1;
  out->exchange_path = h->exchange_used;
  out->exchange_ms = h->exchange_ms;
  out->exchange_wait_ms = h->exchange_wait_ms;
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
  double* T = h->T.as<double>();
  double* cv = h->cvec.as<double>();
  ncclComm_t comm = h->comm;
  hipStream_t xs = h->xstream;
  bool timed_out = false;
  const int rc = run_bounded([=] {
    HIPC(hipSetDevice(dev));
    NCCLC(g_rccl.Broadcast(T, T, Mp * Mp, ncclDouble, root, comm, xs));
    NCCLC(g_rccl.Broadcast(cv, cv, Mp, ncclDouble, root, comm, xs));
    HIPC(hipStreamSynchronize(xs));
    return MIK_OK;
  }, h->rccl_bcast_limit, "ncclBroadcast of the factor", &timed_out);
  if (timed_out) {  // nothing the abandoned collective may still touch is reused: stream, buffers and communicator are leaked
    h->xstream = nullptr;
    if (h->rank != root) {
      h->T.leak();
      h->cvec.leak();
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
  hipLaunchKernelGGL(k_checksum, dim3(1024), dim3(256), 0, h->stream, (const unsigned long long*)h->T.p, Mp * Mp, sd);
  hipLaunchKernelGGL(k_checksum, dim3(4), dim3(256), 0, h->stream, (const unsigned long long*)h->cvec.p, Mp, sd + 2);
  HIPC(hipGetLastError());
  unsigned long long host[4];
  HIPC(hipMemcpyAsync(host, sd, sizeof host, hipMemcpyDeviceToHost, h->stream));
  HIPC(hipStreamSynchronize(h->stream));
  for (int i = 0; i < 4; ++i) out[i] = host[i];
  return MIK_OK;
}

}  // extern "C"
