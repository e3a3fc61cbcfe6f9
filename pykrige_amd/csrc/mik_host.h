// mikrige.hip -- host orchestration + C ABI (include/mikrige.h) of the MI355X kriging execute() path.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude pykrige_amd/csrc/mikrige.hip -ldl
// No torch, no BLAS/solver libraries: every kernel is in mik_kernels.h.  RCCL is dlopen()ed on demand.
#pragma once
#include "mik_dev.h"
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

inline thread_local std::string g_err;
inline int fail(int code, const std::string& msg) {
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
inline std::mutex g_pin_mutex;
inline std::vector<std::pair<void*, size_t>> g_pin_pool;
inline std::map<void*, size_t> g_pin_lent;

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
inline void host_copy(void* dst, const void* src, size_t bytes) {
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
inline void parallel_chunks(long n, F fn) {  // fn(chunk index, begin, end) over at most 8 contiguous chunks
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
inline int chunks_of(long n) { return (int)std::max<long>(1, std::min<long>(8, n / (1L << 20))); }

// np.nonzero(~mask) (ok.py:700) / `if mask[i]: continue` (cok.pyx:57-58): positions of the unmasked cells, ascending
inline void unmasked_positions(const int8_t* mask, long ncells, std::vector<long>& idx) {
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
inline RcclApi g_rccl;
inline int rccl_load() {
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
  DevBuf gu;  // geographic problems, range-aware contraction: unit vectors of the resident points (3 x npt)
  bool ps_valid = false;     // ps_idx[0] holds the order of the resident points for launches of ps_chunk points
  long ps_chunk = 0;
  int opt_sparse_group = 16; // "sparse_group": point blocks per group of k_sp_tiles_g's queue order (a group's tiles run on one XCD, tile position
                             // ascending, point block fast): 1 .. 16 (round 5: 4 -> 16, contraction 35.7 -> 35.3 ms at config 5)
  int opt_sparse_rows = -1;  // "sparse_rows": 16 = tiles of gathered 16-row groups (k_contract_spg), 128 = aligned row blocks (k_contract_sp),
                             // -1 = auto: 16 wherever 32-bit offsets address the inverse (Mp * Mp * 8 < 2^32)
  // second set (with Bt2): the launches of the range-aware contraction alternate between two lanes on two streams, so that the
  // candidate / right-hand-side / list kernels of one launch and the tail of the previous launch's tile queue overlap
  DevBuf sp2_cand, sp2_flags, sp2_klist, sp2_kcount, sp2_nrows, sp2_rows, sp2_rstart, sp2_tiles, sp2_xoff, part2, queue2, sp2_recs;
  int opt_sparse_lanes = 2;  // "sparse_lanes": 2 = two lanes (default since round 5), 1 = one launch after the other on one stream.  Round 4
                             // (profiles/r04_sparse_lanes_ab.txt): config-5 slab 64.6 -> 63.4 ms, 2 % for a second 8.4 GB panel: off.  Round 5, with
                             // the contraction 15 % shorter, what runs beside it weighs more: prediction 43.1 -> 41.6 ms (bench 35.3 -> 36.3 M points/s)
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
  hipStream_t stream3 = nullptr;            // panel stream of the sweep (panel kernel + block-column update), high priority
  std::vector<hipEvent_t> ps_events;
  DevBuf Dnext, Dcopy, Rb;                  // early-diagonal chain: 128 x 128 scratch (next diagonal block, its source tile, one block of R^T rows)
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
  // look-ahead sweep (from 3 block columns on): the next diagonal block is built and inverted ahead of the panel / update stream
  // ("early diagonal" schedule; the schedules it replaced -- rounds 1-2 -- and its flag-ordered variants left the library in round 6)
  int opt_gate = -1;       // look-ahead sweep: the trailing update waits until the next diagonal inverse has started and leaves
                           // it a CU of its own (k_gate); -1 = where the serial chain, not the update, is the step period
  // round 3: the panel kernel and the update of the NEXT block column run on a third stream beside the trailing update of the
  // step before (events only): -1 = from 24 block columns on, 0 = off, 1 = wherever the early-diagonal schedule runs
  int opt_panel_stream = -1;
  int opt_update_rev = -1;  // "update_rev": the half sweep's trailing update walks its tiles backwards on odd steps (k_update): -1 = auto =
                            // from 45 block columns on (the upper triangle no longer fits half of the 256 MB memory-side cache), 0 / 1
  // (Round 6 prune: the 256-column pivot sweep, the deep / prefetching / token-passing trailing updates, the tile map, the 4-wave update and
  // the one-tile panel kernel, the scalar-pivot diagonal inverses -- every one bit- or LAPACK-exact, none faster than what is here -- are no
  // longer in the library: DESIGN_HISTORY.md section 10, tools/mik_k_experiments.h, git history.)
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
  // symmetric contraction: the diagonal block of a tile is contracted as a triangle of 16-row groups -- 36 of
  // its 64 (group, K tile) products (round 3; gemm_core TRI).  0 = the whole diagonal block.
  int opt_tri = 1;
  int opt_symmetrize = 1;  // T <- (T + T^T) / 2 after a full sweep / the pivoted elimination (k_symmetrize); 0 = as eliminated
  int opt_factor = 0, opt_sym = 1;
  long opt_chunk = 131072;
  int opt_mw_pivot = 0;       // 1 = always solve the moving-window systems with partial pivoting
  bool mw_force_piv = false;
  int opt_mw_lds_cap = 8192;  // largest candidate buffer the moving-window neighbour search keeps in LDS
  int opt_mw_knn_bound = 1;   // neighbour search: first pass over the 3 x 3 cells with a distance bound (see k_mw_knn)
  int opt_mw_static = 1;      // k_mw_chol instantiated with the variogram model as a compile-time constant where possible (0: the dynamic form, for A/B)
  int opt_mw_knn_lane = 1;    // neighbour search, windows <= 16: one lane per point first (k_mw_knn_lane), k_mw_knn for what it leaves
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
  // round 6: the exchange moves the packed upper block triangle of the inverse (k_tri_pack) wherever the inverse is exactly symmetric
  DevBuf xpack;                // tri_len(Mp) doubles: packed on the leader / root; received, unpacked into T and mirrored on the others
  int opt_exchange_tri = 1;    // "exchange_tri": 1 (default) = the triangle wherever the device computed the inverse (exactly symmetric), 0 = always the whole square
  bool xpack_valid = false;    // one process per GPU: xpack holds the packed triangle of the current factor (mik_factor_checksum sums it)
  double exchange_bytes = 0.0; // payload one member / rank received in the last exchange
  std::chrono::steady_clock::time_point xchg_t0;
  double exchange_wait_ms = 0.0;  // of exchange_ms, what a caller really waited for (the rest overlapped the leader's prediction)
  int exchange_fallbacks = 0, rccl_ranks = 0;
  int rccl_failures = 0;  // consecutive RCCL exchanges of this handle that FAILED (returned an error; a stall disables RCCL process-wide)
  int opt_async_exchange = 1;  // "async_exchange": mik_factor returns after the leader's K1 + K2; the exchange is joined by the next call
  double rccl_init_limit = 120.0, rccl_bcast_limit = 30.0, peer_limit = 30.0;  // seconds; MIK_RCCL_INIT_TIMEOUT, MIK_RCCL_BCAST_TIMEOUT, MIK_PEER_TIMEOUT
};

inline int get_events(mik_handle* h, size_t n) {
  while (h->evpool.size() < n) {
    hipEvent_t e;
    HIPC(hipEventCreate(&e));
    h->evpool.push_back(e);
  }
  return MIK_OK;
}

inline double host_vario(const Vario& v, double d) {
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

#define DISPATCH_NDIM_FIXED(MODEL, ndim, KERNEL, grid, block, stream, args)                    \
  do {                                                                                        \
    if ((ndim) == 1) hipLaunchKernelGGL((KERNEL<MODEL, 1>), grid, block, 0, stream, args);    \
    else if ((ndim) == 3) hipLaunchKernelGGL((KERNEL<MODEL, 3>), grid, block, 0, stream, args); \
    else hipLaunchKernelGGL((KERNEL<MODEL, 2>), grid, block, 0, stream, args);                \
  } while (0)

// the order the next factor will be in: "sparse" -1 (auto) / 1 / 2 = Hilbert-curve order wherever the problem allows it
// (measured even at N = 100: the four small list kernels per launch cost less than the dense tiles they save)
inline bool want_sorted(const mik_handle* h) { return h->sort_ok && h->opt_sparse != 0; }

// ---- functions shared between the library's translation units (C++ linkage; definitions: the file named) ----
int custom_roundtrip(mik_handle* h, double* dev, long rows, long cols, long ld);                                 // mikrige.hip
int launch_assemble(mik_handle* h, double shift, double* dst = nullptr, bool sorted = false, bool eq = false);  // mikrige.hip
int ensure_factor_buffers(mik_handle* h);                                                                        // mik_inverse.hip
int launch_mirror_upper(double* T, long Mp, hipStream_t st);                                                         // mik_inverse.hip
int one_factor(mik_handle* h);                                                                                   // mik_inverse.hip
int sort_points(mik_handle* h, long chunk, long nchunks);                                                        // mik_predict.hip
int one_predict(mik_handle* h);                                                                                  // mik_predict.hip
int one_predict_mw(mik_handle* h, int n_closest);
// mik_mw_chol.hip, part N: launches class 100 G + RI of k_mw_chol if it holds it, else returns MIK_MWC_NOCLASS
#define MIK_MWC_PARTS 5
#define MIK_MWC_NOCLASS (-9999)
namespace mik { struct MwArgs; }
int dispatch_mw_solve(mik_handle* h, const mik::MwArgs& a, long pc);                                               // mik_mw_solve.hip
int mw_chol_part0(int cls, hipStream_t stream, bool use_static, const mik::MwArgs& a, long pc);
int mw_chol_part1(int cls, hipStream_t stream, bool use_static, const mik::MwArgs& a, long pc);
int mw_chol_part2(int cls, hipStream_t stream, bool use_static, const mik::MwArgs& a, long pc);
int mw_chol_part3(int cls, hipStream_t stream, bool use_static, const mik::MwArgs& a, long pc);
int mw_chol_part4(int cls, hipStream_t stream, bool use_static, const mik::MwArgs& a, long pc);
inline int mw_chol_part(int part, int cls, hipStream_t stream, bool use_static, const mik::MwArgs& a, long pc) {
  switch (part) {
    case 0: return mw_chol_part0(cls, stream, use_static, a, pc);
    case 1: return mw_chol_part1(cls, stream, use_static, a, pc);
    case 2: return mw_chol_part2(cls, stream, use_static, a, pc);
    case 3: return mw_chol_part3(cls, stream, use_static, a, pc);
    default: return mw_chol_part4(cls, stream, use_static, a, pc);
  }
}                                                                // mik_mw.hip
