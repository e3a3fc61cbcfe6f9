/* mikrige.h -- C ABI of libmikrige.so: the MI355X (gfx950) kriging execute() path.
 *
 * This is the drop-in boundary for the native half of PyKrige's execute() hot path.  What it
 * replaces in the reference (paths relative to /root/reference/src/pykrige):
 *
 *   lib/cok.pyx:14-96    cpdef _c_exec_loop(a_all, bd_all, mask, n, pars)      -> mik_factor + mik_predict
 *   lib/cok.pyx:98-193   cpdef _c_exec_loop_moving_window(...) + cKDTree.query  -> mik_predict_moving_window
 *   lib/cok.pyx:196-203  check_b_vect (|bd| <= eps  ->  b = 0)                 -> inside mik_predict
 *   lib/variogram_models.pyx:6-84  C variogram kernels selected by name        -> mik_problem.model_id
 *   ok.py:626-648, uk.py:861-920, ok3d.py:603-622, uk3d.py:688-737
 *                        _get_kriging_matrix (cdist + -gamma + border)         -> mik_factor (K1 assemble)
 *   ok.py:663, uk.py:935, ok3d.py:637, uk3d.py:752, cok.pyx:53
 *                        scipy.linalg.inv(a)                                    -> mik_factor (K2 invert)
 *   ok.py:989, uk.py:1293, ok3d.py:899, uk3d.py:1122   cdist(points, data)      -> inside mik_predict
 *   ok.py:650-683, uk.py:922-1009, ok3d.py:624-657, uk3d.py:739-811
 *                        _exec_vector (b build, A_inv.b, z / sigma^2 sums)      -> mik_predict (K3)
 *
 * Differences from _c_exec_loop's signature, on purpose: the library takes COORDINATES, not the
 * npt x N distance matrix `bd` (40 GB at N=5000 / 1000x1000; it never exists here), and the kriging
 * matrix is assembled and inverted on the device instead of being passed in.
 *
 * Conventions: every pointer is a host pointer to C-contiguous float64 (int8 for the mask), borrowed
 * for the duration of the call only.  Outputs are caller-allocated.  All calls are blocking.  A
 * handle is bound to one HIP device (or to a group of them, mik_set_devices) and is not thread-safe; distinct handles
 * are independent.
 * Return value 0 = success; negative = error (mik_last_error() has the text for this thread).
 */
#ifndef MIKRIGE_H
#define MIKRIGE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ABI version: bumped whenever the meaning of a field or an argument changes under an unchanged signature.
 *   4 -> 5  mik_grid.cell_count == 0 is an EMPTY range (it used to mean "the whole grid", which is now -1): a caller that
 *           zero-initialises mik_grid must set cell_count = -1.  mik_abi_version() returns the library's value; the Python
 *           loader (pykrige_amd/_lib.py) refuses a library whose version differs from the header it was written against. */
#define MIK_ABI_VERSION 7
/*   5 -> 6  mik_timing grew by sparse_ktile + reserved2 (8 bytes appended; earlier fields unchanged). */
/*   6 -> 7  mik_timing grew by exchange_bytes (8 bytes appended); the factor exchange moves the packed upper block triangle of the inverse
 *           (option "exchange_tri"); the option keys of the experiments of rounds 2-5 are refused (see the option list). */

#define MIK_OK          0
#define MIK_EINVAL     (-1) /* bad argument            -> Python ValueError                  */
#define MIK_ESINGULAR  (-2) /* singular kriging matrix -> numpy.linalg.LinAlgError (what scipy.linalg.inv raises) */
#define MIK_EHIP       (-3) /* HIP runtime failure     -> RuntimeError                       */
#define MIK_ERCCL      (-4) /* RCCL failure            -> RuntimeError                       */
#define MIK_ESTATE     (-5) /* call order violated (e.g. predict before factor) -> RuntimeError */

/* variogram_models.py:25-81 / lib/variogram_models.pyx:25-84 (hole-effect exists only in the .py) */
#define MIK_MODEL_LINEAR      0 /* params = [slope, nugget]            */
#define MIK_MODEL_POWER       1 /* params = [scale, exponent, nugget]  */
#define MIK_MODEL_GAUSSIAN    2 /* params = [psill, range, nugget]     */
#define MIK_MODEL_SPHERICAL   3
#define MIK_MODEL_EXPONENTIAL 4
#define MIK_MODEL_HOLE_EFFECT 5
#define MIK_MODEL_CUSTOM      6 /* variogram_model='custom': gamma is a host callable (mik_set_custom_variogram) */

typedef struct mik_handle mik_handle;

/* What a constructed kriging object carries into execute() (the `pars` dict of ok.py:916-927 plus
 * the adjusted station coordinates and the drift description of uk.py:861-920). */
typedef struct mik_problem {
  int32_t ndim;             /* 2 or 3 */
  int32_t model_id;         /* MIK_MODEL_* */
  int64_t n;                /* number of stations */
  const double *xs, *ys, *zs; /* anisotropy-ADJUSTED station coordinates (X_ADJUSTED...), zs NULL if ndim==2 */
  const double *values;     /* self.Z / self.VALUES, length n */
  double params[3];         /* variogram_model_parameters (internal form: psill, not sill) */
  double eps;               /* self.eps (1e-10) */
  int32_t exact_values;     /* self.exact_values */
  int32_t regional_linear;  /* UK: drift columns x, y[, z] (uk.py:877-883, uk3d.py:708-717) */
  int32_t n_wells;          /* UK 2D point_log wells (uk.py:884-896) */
  int32_t n_extra;          /* host-evaluated drift terms: external_Z, specified, functional (in that order) */
  const double *wells;      /* n_wells x 3 row-major: adjusted x, adjusted y, strength */
  const double *extra_cols; /* n_extra x n row-major: those drift terms evaluated at the stations */
  const double *a_inv;      /* optional (M x M, M = n + ndrift + 1): an inverse supplied by the caller, used as is;
                               NULL = invert on device.  It is the inverse of a symmetric matrix: the symmetric
                               contraction (option "symmetric", default) reads its upper triangle only */
  int32_t geographic;       /* coordinates_type == 'geographic' (ordinary 2D only): xs/ys and px/py are lon/lat in
                               degrees, distances are great-circle degrees (core.py:36-97; ok.py:634-640, 990-996) */
  int32_t pseudo_inv;       /* self.pseudo_inv: 0 = inverse; 1 = 'pinv', 2 = 'pinvh' (core.py:33 P_INV): Moore-Penrose
                               pseudo-inverse on the device (the two types coincide on a symmetric matrix): the regular inverse of
                               the matrix deflated by the null space of duplicated stations when that provably is it, else a
                               one-sided Jacobi SVD */
} mik_problem;

/* The prediction points handed to _exec_vector: adjusted coordinates (SoA), mask, drift rows. */
typedef struct mik_points {
  int64_t npt;
  const double *px, *py, *pz; /* anisotropy-ADJUSTED point coordinates (or raw ones followed by mik_adjust_points); pz NULL if ndim==2 */
  const int8_t *mask;         /* nullable; nonzero = skip the point (outputs stay 0.0, cok.pyx:25-26,57-58) */
  const double *extra_rows;   /* n_extra x npt row-major: host-evaluated drift terms at the points; nullable if n_extra==0 */
} mik_points;

/* The prediction points of style='grid' / 'masked' given by their AXES: what execute() builds with np.meshgrid and
 * adjusts point by point on the host (ok.py:863-885, uk.py:1263-1284, ok3d.py:866-883, uk3d.py:1089-1106;
 * core.py:120-193 _adjust_for_anisotropy) is generated on the device, so the H2D traffic of a grid is O(nx + ny + nz)
 * instead of npt x d doubles.  Point order = the reference's flattened meshgrid: 2-D cell iy*nx + ix, 3-D cell
 * (iz*ny + iy)*nx + ix.  Arithmetic in the reference's order (X -= c ; rot . X ; stretch . (..) ; += c). */
typedef struct mik_grid {
  int32_t ndim;                /* 2 or 3 (must equal the problem's) */
  int32_t adjust;              /* 1 = apply the anisotropy transform below; 0 = take the axes' values as they are
                                  (geographic coordinates: ok.py:892-896) */
  int64_t nx, ny, nz;          /* axis lengths; nz ignored when ndim == 2 */
  const double *gx, *gy, *gz;  /* xpoints, ypoints[, zpoints] as handed to execute(), fp64 */
  double center[3];            /* XCENTER, YCENTER[, ZCENTER] */
  double rot[9];               /* rotation matrix of core.py:150-154 / 166-187, row-major ndim x ndim (leading entries) */
  double stretch[3];           /* diagonal of the stretch matrix: 1, scaling  /  1, scaling_y, scaling_z */
  int64_t cell_first, cell_count; /* krige only cells cell_first .. cell_first + cell_count - 1 of the flattened grid (one
                                  rank's slab when the caller shards: pykrige_amd.dist); cell_count = -1: the whole grid
                                  (cell_first ignored); cell_count = 0: an EMPTY range -- nothing is kriged (a rank of a run
                                  with more ranks than cells).  mask, extra_rows and the outputs of mik_get_results are
                                  relative to this range */
  const int8_t *mask;          /* nullable; one byte per cell of the range, flattened like the cells; nonzero = skip */
  const double *extra_rows;    /* n_extra x (cells of the range) row-major host-evaluated drift terms; nullable if n_extra == 0 */
} mik_grid;

/* Per-phase device times of the last mik_factor / mik_predict (HIP events on the handle's stream). */
typedef struct mik_timing {
  double assemble_ms;     /* K1: kriging-matrix assembly */
  double invert_ms;       /* K2: block inverse (all launches) */
  double rhs_ms;          /* K3a: RHS/variogram assembly launches, summed */
  double contract_ms;     /* K3b: the dense contraction kernel (dominant), summed over launches */
  double predict_ms;      /* whole mik_predict on the stream */
  int64_t contract_launches;
  double contract_flops_executed; /* flops the contraction kernel really executed (symmetric form: ~M^2/pt) */
  int32_t factor_path;    /* 1 = unpivoted symmetric block sweep on the SPD-shifted matrix, 2 = pivoted block Gauss-Jordan, 3 = caller-supplied inverse, 4 = device pseudo-inverse (one-sided Jacobi), 5 = pseudo-inverse as the regular inverse of the matrix deflated by the duplicated stations' null space (verified with probe vectors), 6 = pseudo-inverse as the regular inverse of the matrix deflated by a numerically found null space (shift-and-invert subspace iteration + Rayleigh-Ritz; verified the same way) */
  int32_t symmetric;      /* 1 = contraction used the symmetric half product */
  int32_t engine;         /* 0 = v_mfma_f64_4x4x4_4b_f64 contraction, 1 = v_fma_f64 register-tiled contraction */
  int32_t reserved;       /* mik_get_device_timing: the HIP device index of that group member */
  double exchange_ms;     /* device groups: host wall time of the factor exchange of the last mik_factor (0 for one device) */
  int32_t exchange_path;  /* 0 = none (one device), 1 = RCCL broadcast, 2 = peer copies (scatter + all-gather), 3 = every device factored */
  int32_t n_devices;      /* members of the handle's device group */
  double exchange_wait_ms;     /* of exchange_ms, the part a caller was blocked for (the rest overlapped the leader's prediction) */
  int32_t exchange_fallbacks;  /* exchange paths that failed or timed out before exchange_path succeeded (mik_exchange_note says why) */
  int32_t rccl_ranks;          /* communicators (= ranks = devices) the RCCL broadcast ran over; 0 if RCCL was not used */
  int32_t mw_kernel;           /* mik_predict_moving_window: the per-point solver that ran (contract_ms is its time): 1 = k_mw_chol
                                  (LDL^T in registers), 2 = k_mw_solve (Gauss-Jordan in registers), 3 = k_mw_solve_big (LU in HBM
                                  scratch), 4 = k_mw_chol_blocked (blocked Cholesky, panels in LDS); 0 after mik_predict */
  int32_t half_sweep;          /* 1 = the block sweep maintained only the upper block triangle */
  int32_t factor_attempts;     /* factorisations mik_factor ran: 1, or more when a bad pivot / a failed probe sent it to a more
                                  careful path (half sweep -> full sweep -> partial pivoting) */
  int32_t null_dim;            /* factor_path 6: dimension of the null space that was found and deflated */
  int32_t rhs_overlapped;      /* 1 = two right-hand-side panels: k_rhs of chunk c + 1 ran on a second stream under chunk c's contraction */
  double verify_ms;            /* the probe of the inverse (all attempts) */
  double verify_res_z;         /* max |A c - [Z; 0]| / max(1, max|Z|), c = A_inv[:, :n] Z: bounds the error of z (last attempt) */
  double verify_res_inv;       /* max_j max |A_inv A e_j - e_j| over three station columns (last attempt) */
  int32_t sparse;              /* 1 = mik_predict ran the range-aware contraction (compact-support variogram: option "sparse") */
  int32_t stations_sorted;     /* 1 = the factor holds the stations in Hilbert-curve order (mik_get_matrix un-permutes) */
  double sparse_tiles;         /* range-aware contraction: tiles (active row block x point block) contracted, all launches */
  double sparse_tiles_dense;   /*   ... and the tiles the dense symmetric contraction runs for the same points */
  double sparse_ktiles;        /* off-diagonal K tiles (16 stations x 128 rows x 128 points) contracted, summed over the tiles */
  double sparse_ktiles_dense;  /*   ... and of the dense symmetric contraction */
  double sparse_lists_ms;      /* candidate test + flag -> list kernels (k_sp_cand, k_sp_lists, k_sp_tiles), summed */
  double sparse_diag_products; /* (16-row group x 16-station K tile x 128 points) products of the tiles' triangular parts, all launches */
  int32_t sparse_rows;         /* rows of a tile of the range-aware contraction: 16 = eight gathered 16-row groups (k_contract_spg),
                                  128 = aligned row blocks (k_contract_sp), 0 = dense contraction (option "sparse_rows") */
  int32_t points_sorted;       /* 1 = the range-aware contraction ran over the points of every launch in Hilbert-curve order (option
                                  "sort_points") */
  double sort_points_ms;       /* the device sort of the points (k_ps_*), when this mik_predict had to run it (part of predict_ms) */
  int32_t sparse_ktile;        /* stations per candidate / list tile of the range-aware contraction: 16, or 8 (a K step is then a pair of
                                  list-adjacent 8-station tiles: option "sparse_ktile"; ABI 5 -> 6: appended) ; 0 = dense */
  int32_t reserved2;
  double exchange_bytes;       /* payload one group member / rank received in the last factor exchange: 8 (tri_len + Mp) with the packed upper block
                                  triangle (Mp (Mp + 128) / 2 doubles; option "exchange_tri", the default for every inverse the device computed), 8 (Mp^2 + Mp)
                                  with the whole square; 0 = no exchange.  ABI 6 -> 7: appended */
} mik_timing;

int  mik_device_count(void);
int  mik_abi_version(void); /* MIK_ABI_VERSION of the built library */
int  mik_create(int device, mik_handle **out);
void mik_destroy(mik_handle *h);

/* Single-process multi-GPU.  A handle can span n devices of the node (a "device group": the
 * handle's own device + the next n-1 visible ones); every call below then works on the group with unchanged signatures,
 * so OrdinaryKriging.execute() (ok.py:760-768) scales over the node without any change to the caller's script:
 *   mik_set_problem   stations to every member                    mik_factor   K1 + K2 on the handle's own device, then ONE
 *   mik_set_points    unmasked points cut into n contiguous slabs              broadcast of the inverted matrix and of c
 *   mik_predict       every member kriges its slab, own stream                 (RCCL over xGMI: ncclCommInitAll + grouped
 *   mik_get_results   every member copies its slab into the caller's           ncclBroadcast; or peer copies as scatter +
 *                     z_out / ss_out at its offset (page-locked staging)       all-gather; or no exchange: all factor)
 * No cross-GPU reduction, no halo.
 * mik_set_devices(n): process-wide default for handles created afterwards (n = 0: every visible device; the environment
 * variable MIK_NGPU = n | "all" sets the same default without touching the script).  mik_handle_set_devices: one handle. */
int  mik_set_devices(int n);
int  mik_handle_set_devices(mik_handle *h, int n);
int  mik_handle_devices(mik_handle *h);          /* members of the handle's device group (1 = single device) */
int  mik_slab_of(int64_t n, int members, int i, int64_t *lo, int64_t *count); /* the slab of n unmasked points member i gets
                                                    (contiguous, cut at multiples of 128 points); needs no GPU */

/* options (round 6: every option is either a documented fallback, a cross-check kernel of the parity tests, or a schedule switch of a default
 * path; the experiments of rounds 2-5 -- "engine", "waves", "update_waves", "update_deep", "update_tpb", "update_pf", "update_token",
 * "update_map", "pivot256", "wide_reserve", "wide_colstream", "panel_rows", "diag", "early_diag", "fuse_chain", "sparse_ktile",
 * "sparse_epilogue", and before them "pairs", "prefetch", "update_atomic" -- are refused as unknown; their kernels live in
 * tools/mik_k_experiments.h or git history, their measurements in profiles/ and DESIGN_HISTORY.md):
 * "factor" 0=auto 1=sweep 2=pivoted ; "symmetric" 0/1 (1 = the contraction forms b^T A_inv b over one triangle of A_inv; 0 = the reference's
 *   full product w = A_inv b, ok.py:679: the cross-check kernel of the parity tests) ;
 * "exchange_tri" 0/1 = device groups / ranks: the factor exchange moves the packed UPPER BLOCK TRIANGLE of the inverse (block row I keeps its
 *   columns from 128 I on: Mp (Mp + 128) / 2 doubles, 264 MB instead of 520 MB at N = 8000), packed on the leader / root (k_tri_pack, 0.1 ms),
 *   checksummed, and on every member unpacked into its matrix and MIRRORED into the lower block triangle (two local kernels, 0.3 ms): the member
 *   then holds the leader's matrix bit for bit.  That needs an inverse that is exactly symmetric by construction: every inverse the device computes
 *   is (the half sweep mirrors its triangle; the full sweep, the pivoted elimination and the pseudo-inverses end in k_symmetrize).  1 (default) = the
 *   triangle unless the inverse is the caller's (mik_problem.a_inv: used as handed over) or "symmetrize" is 0; 0 = always the whole square.
 *   mik_timing.exchange_bytes says what travelled [MIK_EXCHANGE_TRI] ;
 * "sparse" -1/0/1/2 = range-aware contraction for variograms with compact support (the reference's spherical model is constant
 *   beyond its range, variogram_models.py:56-70).  With u = [1_N; 0] and s = psill + nugget the right-hand side is b = -s u + delta,
 *   delta_k = s - gamma(d_k) = 0 for every station beyond the range, and because A e_last = u:  z = c . delta  and
 *   sigma^2 = 2 s - delta^T A_inv delta  exactly.  The stations are laid out along a Hilbert curve, K3a writes delta and flags the
 *   (128 points x 16 stations) tiles that hold a nonzero, K3b contracts only those (k_contract_spg; see "sparse_rows").  -1 (default) = 1 = on for the
 *   spherical model (not with pseudo_inv or a caller's a_inv: those run the dense contraction; geographic coordinates since round 5:
 *   candidates by boxes of the stations' / points' UNIT VECTORS against the chord 2 sin(range / 2) -- monotone in the great-circle
 *   distance ok.py:634-640, 990-996 take --, delta from the great-circle distance as in the dense path), 0 = off,
 *   2 = Hilbert-ordered stations with the dense contraction (A/B of the order alone).  Takes effect at the
 *   next mik_factor [MIK_SPARSE].  REPRODUCIBILITY: the set of K tiles a point meets depends on the 128-point block it falls
 *   into, so with this option on (with or without "sort_points") sigma^2 depends TO ROUNDING (~1e-13) on how the points are cut
 *   into launches, slabs and device-group members; so does z since round 5 (~1e-16 relative: k_rhs walks the block's list of candidate
 *   tiles, which lane adds which station follows the list).  Cuts at multiples of 128 points in the caller's order (device groups with
 *   "sort_points" 0) keep the blocks, hence the bits.  "sparse" 0 is bit-identical across device counts
 *   (tests/test_device_group.py::test_spherical_model_across_device_counts) ;
 * "sparse_rows" -1/16/128 = range-aware contraction: a tile's 128 rows of A_inv are eight GATHERED active 16-row groups of the point
 *   block's list (16: k_contract_spg -- the list of active K tiles is also the list of active row groups; tile r takes entries
 *   [8r, 8r + 8) as rows, the entries beyond as K tiles, then its own groups as a triangle) or an ALIGNED block of 128 rows that is
 *   contracted whole when any of its eight groups is active (128: k_contract_sp; active blocks are ~79 % full at BASELINE config
 *   5).  -1 (default) = 16 wherever 32-bit DMA offsets reach every row (Mp <= 23168), else 128 [MIK_SPARSE_ROWS] ;
 *   With gathered row groups the flags and lists are per 8 stations (round 5): a K step of the contraction is a PAIR of list-adjacent
 *   8-station tiles staged into the two halves of the 16-wide LDS tile, a 16-row group is two gathered 8-row groups, an odd last entry is
 *   half a K step (718 instead of 780 stations in active tiles at BASELINE config 5; work ~ n^2); a row group's term sum_i delta_ti W_it
 *   is formed at the K step of the group's own square from that step's B tile in LDS.  Aligned blocks keep 16-station lists ;
 * "sparse_group" 1..16 = k_contract_spg's queue order: point blocks per group (a group's tiles run on one XCD, tile position ascending
 *   = longest K loops first, point block fast; default 16) [MIK_SPARSE_GROUP] ;
 * "sort_points" -1/0/1 = range-aware contraction: the points of every launch (one chunk of the resident point list) are put in
 *   Hilbert-curve order among themselves on the device (k_ps_*: 20-bit keys, stable two-pass radix sort, all launches' segments
 *   side by side) and kriged in that order -- a block of 128 consecutive points is then a compact patch whatever order the caller's
 *   list or grid is in (a shuffled list, a row of a 3-D grid), and the contraction's cost grows with the square of the stations in
 *   range of a block.  z and sigma^2 come back in the caller's order.  -1 (default) = on with the range-aware contraction from 4096 points and
 *   512 matrix rows on (below that the seven sort launches cost more than they save), 1 = always, 0 = off.  With the sort on, sigma^2
 *   of the range-aware path depends to rounding (~1e-13) on how the points are cut into launches and slabs [MIK_SORT_POINTS] ;
 * "sparse_lanes" 1/2 = range-aware contraction: its launches alternate between two lanes (two streams, two sets of work buffers and
 *   right-hand-side panels), so that the candidate / right-hand-side / list kernels of a launch and the tail of the previous
 *   launch's tile queue overlap.  Default 2 (round 5: prediction 43.1 -> 41.6 ms per 2.1 M points at BASELINE config 5, for a second
 *   right-hand-side panel); 1 = one launch after the other.  With two lanes a kernel's event-timed duration includes what ran beside it ;
 * "drift_eq" 0/1 = drift equilibration (default 1): every drift term enters the matrix and the right-hand sides as s_j (f_j - c_j),
 *   c_j = its mean and 1 / s_j = its largest deviation over the stations (wells excepted).  With the unbiasedness row present
 *   span{1, f_j} = span{1, s_j (f_j - c_j)}: the stations' kriging weights, z and sigma^2 are those of the reference's system
 *   (uk.py:861-920, 949-981) -- A' = S A S^T, b' = S b -- but coordinates of 1e6 no longer share a matrix with semivariances of 1e2
 *   (the reference's own KT3D test case: cond(A) 3e14 -> 2e6; the unpivoted sweep's |dz| 6e-9 -> 2e-11).  Not with pseudo_inv
 *   (a pseudo-inverse is not invariant under S) or a caller's a_inv.  mik_get_matrix(1) hands out the inverse of the
 *   reference's matrix, S^T A'^-1 S ;
 * "tri" 0/1 = symmetric contraction: the diagonal block of a tile is contracted as a triangle of 16-row groups,
 *   36 of its 64 (row group, K tile) products (default 1: -1.5 % contraction time, partials equal to 1e-14) [MIK_TRI] ;
 * "symmetrize" 0/1 = after a full sweep, the pivoted elimination or a pseudo-inverse: A_inv <- (A_inv + A_inv^T) / 2 (default 1).
 *   The symmetric contraction reads one triangle of A_inv; a quadratic form sees only the symmetric part, so with the average
 *   in both triangles the half product equals b^T A_inv b of the matrix as eliminated (the half sweep mirrors its triangle
 *   anyway; an inverse the caller supplies, mik_problem.a_inv, is never touched) ;
 * "chunk" = largest number of points per contraction launch (multiple of 128; the points are cut into equal launches) ;
 * "rhs_overlap" 0/1 = two right-hand-side panels: K3a of the next chunk runs on a second stream while the current chunk is
 *   contracted (default 0: measured a tie -- the contraction slows down by what K3a takes) [MIK_RHS_OVERLAP] ;
 * "lookahead" 0/1/-1 = overlap the next panel's serial chain with the current trailing update in the block sweep
 *   (default -1: from 3 block columns on).  The look-ahead schedule is the "early diagonal" one: the next diagonal block is built from 128
 *   panel rows (two distributed 128^3 products) and inverted on the second stream AHEAD of the panel kernel and update of its step; the
 *   streams are ordered by events only; every schedule returns the bit-identical inverse ;
 * "update_rev" -1/0/1 = half sweep: on odd steps the trailing update walks every XCD's range of tiles from its end.  The sweep streams the
 *   whole upper block triangle once per step (260 MB at N = 8000: more than the 256 MB memory-side cache holds); a cyclic stream
 *   leaves nothing behind in an LRU cache, a back-and-forth one most of it.  Same tiles, same bits; measured N = 8000 14.0 -> 13.45 ms, a
 *   tie at N <= 5000: -1 (default) = on from 45 block columns [MIK_UPDATE_REV] ;
 * "panel_stream" 0/1/-1 = early-diagonal sweep: the panel kernel and the update of the next block column (+ the diagonal tile
 *   after next) run on a third stream beside the rest of the previous step's trailing update, ordered by events only (default
 *   -1 = from 24 block columns on; same bits) [MIK_PANEL_STREAM] ;
 * "pinv_fast" 0/1 = pseudo_inv: try the deflated regular inverse before the Jacobi pseudo-inverse (default 1) ;
 * "pinv_block" -1/0/1 = the Jacobi pseudo-inverse (factor_path 4) in its block form (default -1 = from 1536 rows on; 1 = always): rows in blocks of 32 sorted by norm, per
 *   pair of blocks one pass for the 64 x 64 Gram matrix, its eigenproblem by a two-sided Jacobi in LDS, one pass for the block
 *   rotation of B and W -- ~3 M / 32 passes over the matrix per sweep instead of ~2 M (0 = one row pair per workgroup, rounds 1-3) ;
 * "verify" 0/1 = probe every inverse the device computes against the matrix itself before it is used (default 1):
 *   res_z = max |A c - [Z; 0]| / max(1, max|Z|) with c = A_inv[:, :n] Z (bounds the error of z: z_g = w_g . (A c)) and
 *   res_inv = max |A_inv A e_j - e_j| over three station columns.  "verify_tol_z" (default 5e-10) / "verify_tol_inv" (1e-8):
 *   a half sweep the library chose by itself that exceeds them is redone as a full sweep, a sweep of factor = auto that
 *   exceeds them by partial pivoting; mik_timing has the residuals, the attempts and the time (0.25 ms at N = 5000) ;
 * "gate" 0/1/-1 = look-ahead sweep: the trailing update of a step starts only once the next diagonal inverse sits on a CU of its
 *   own (default -1: where the serial chain bounds the step) ;
 * "symsweep" 0/1/-1 = sweep only the upper block triangle (faster; 10-100 x the rounding error of the full sweep, which stays far
 *   inside the tolerance for the exponential and spherical models and does not for power + drift terms); default -1 = by
 *   itself for exponential / spherical from 24 block columns on ;
 * "mw_pivot" 0/1 = always solve the moving-window systems with partial pivoting (default 0: SPD-shifted, no pivot search,
 *   falling back to pivoting when a local system is not positive definite) ;
 * "mw_knn_bound" 0/1 = moving-window neighbour search: the first pass takes only stations within the radius expected to hold
 *   K + 4 sqrt(K) + 2 of them (one scan of the 3 x 3 cells, one sort), falling back to the unbounded ring walk where that
 *   finds fewer than K (default 1) ;
 * "mw_static" 0/1 = moving-window LDL^T kernel instantiated with the variogram model as a compile-time constant (linear, gaussian,
 *   spherical, exponential; Euclidean coordinates; default 1).  The dynamic form inlines the great-circle distance and all six models
 *   at every element of the register tile: 1.3 MB of set-up code per kernel, streamed through a 64 KB instruction cache by every point ;
 * "mw_knn_lane" 0/1 = moving-window neighbour search for windows <= 16 over spatially ordered points (default 1): first one LANE per point -- the 64 consecutive points
 *   of a wavefront (the rows of a grid) scan the box of station cells that covers their 3 x 3 (x 3) neighbourhoods and keep their
 *   nearest in registers by sorted insertion (no candidate buffer, no sort); points it cannot finish (sparse corners, scattered
 *   point lists) go to the wave-per-point search ;
 * "mw_class" = 100 G + RI: force one thread-grid (G x G threads per point) / register-tile (RI x RI per thread) class of the
 *   moving-window LDL^T kernel, windows up to G RI (0 = chosen by window size; 1 = the blocked Cholesky kernel of the large
 *   windows whatever the size; for A/B runs) ;
 * "mw_lds_cap" = largest moving-window candidate buffer kept in LDS (entries, default 8192; 0 forces the HBM lists) ;
 * "exchange" 0..3 = how a device group distributes the inverted matrix: 0 auto (RCCL broadcast, peer copies if RCCL is
 *   unavailable), 1 RCCL broadcast, 2 peer copies (scatter + all-gather over xGMI), 3 none (every device factors) [MIK_EXCHANGE] ;
 * "alias_devices" 0/1 = a device group may place several members on one physical GPU (1-GPU test boxes) [MIK_ALIAS_DEVICES] ;
 * "async_exchange" 0/1/2 = device groups with exchange = auto: 0 = mik_factor blocks until every member holds the inverse; 1
 *   (default) = it returns once the leader has factored, the exchange is joined by the next call, and mik_predict lets the leader
 *   krige its slab during a copy-engine transfer (peer copies) but joins an RCCL broadcast first (its root needs compute units
 *   the leader's persistent contraction launch would hold for 47 - 480 ms); 2 = the leader's prediction overlaps any exchange
 *   [MIK_ASYNC_EXCHANGE].  A forced exchange path always completes (or fails) inside mik_factor ;
 * "rccl_init_timeout", "rccl_bcast_timeout", "peer_timeout" = the bounded waits of the factor exchange, seconds (see
 *   "Bounded waits" below) [MIK_RCCL_INIT_TIMEOUT, MIK_RCCL_BCAST_TIMEOUT, MIK_PEER_TIMEOUT] */
int  mik_set_option(mik_handle *h, const char *key, double value);

/* variogram_model='custom' (a Python callable in the reference: ok.py:305-318, core.py:584-586).  Geometry stays on the
 * device: the kernels write distances (station-station, point-station) into their output slots, the library brings them
 * to the host, `fn` maps d -> gamma(d) in place over a rows x cols block with row stride ld, and they go back; matrix
 * borders, drift terms, the eps rule, the inverse and the contraction are the device's as for the named models. */
typedef void (*mik_variogram_fn)(void *user, double *d_inout, int64_t rows, int64_t cols, int64_t ld);
int  mik_set_custom_variogram(mik_handle *h, mik_variogram_fn fn, void *user);

int  mik_set_problem(mik_handle *h, const mik_problem *p); /* H2D of stations/values/drifts            */
int  mik_station_order(const mik_problem *p, int32_t *order_out); /* the Hilbert-curve order option "sparse" lays the stations out
                                                              in: order_out[i] = index (in p's arrays) of the station at position i.
                                                              Diagnostic; needs no GPU.  Only xs / ys / zs, n and ndim of p are read */
int  mik_factor(mik_handle *h);                            /* K1 + K2 (+ c = A_inv[:, :n].Z) on device   */
int  mik_set_points(mik_handle *h, const mik_points *g);   /* H2D of the (unmasked) points              */
int  mik_set_grid(mik_handle *h, const mik_grid *g);       /* the same for a grid given by its axes: H2D of the axes (and
                                                              the compacted cell indices of a mask), points generated and
                                                              anisotropy-adjusted on the device */
int  mik_adjust_points(mik_handle *h, const double center[3], const double rot[9], const double stretch[3]);
                                                           /* style='points' (round 3): mik_set_points was handed the RAW
                                                              coordinates; apply the anisotropy adjustment (core.py:120-193
                                                              _adjust_for_anisotropy, what execute() does on the host at
                                                              ok.py:879-885) to them in place on the device -- same
                                                              arithmetic and argument meaning as mik_grid's center / rot /
                                                              stretch.  Not for points generated by mik_set_grid */
int  mik_predict(mik_handle *h);                           /* K3 over the resident points; results stay in HBM */
int  mik_get_results(mik_handle *h, double *z_out, double *ss_out); /* D2H, scattered through the mask  */
int  mik_take_results(mik_handle *h, double **z, double **ss); /* the same without the last copy: *z and *ss point INTO the
                                                              page-locked landing zone the device wrote (npt doubles each, one
                                                              allocation, *ss = *z + npt), which now belongs to the caller until
                                                              mik_release_results(*z); the handle takes another buffer (recycled
                                                              from released ones) for its next predict.  One device, no mask;
                                                              otherwise MIK_ESTATE and mik_get_results is the call */
void mik_release_results(double *z);                       /* gives such a buffer back (any thread) */
int  mik_synchronize(mik_handle *h);                       /* wait until the handle's stream is idle (every call above
                                                              already blocks; this is the explicit bracket for timing) */

/* Moving-window ordinary kriging (n_closest_points): replaces cKDTree.query + _c_exec_loop_moving_window
 * (ok.py:929-986, lib/cok.pyx:98-193; ok3d.py:901-912, 697-733).  Needs mik_set_problem + mik_set_points
 * (no mik_factor: each point solves its own (k+1)x(k+1) system).  Results as for mik_predict. */
int  mik_predict_moving_window(mik_handle *h, int n_closest_points);

/* Variogram-fit statistics: replaces core._find_statistics -> core._krige (core.py:759-836, 654-756): for
 * i = 1..n-1 station i is kriged from stations 0..i-1.  k_out / ss_out have n entries (entry 0 unused = 0). */
int  mik_statistics(mik_handle *h, double *k_out, double *ss_out);

/* Experimental semivariogram of the constructor: replaces pdist + the lag-bin loop of
 * core._initialize_variogram_model (core.py:432-505).  lags_out / semi_out hold up to nlags entries (empty bins are
 * dropped); *n_out = number written.  Needs mik_set_problem only. */
int  mik_experimental_variogram(mik_handle *h, int nlags, double *lags_out, double *semi_out, int32_t *n_out);

/* One-shot convenience: create + set_problem + factor + set_points + predict + get_results + destroy. */
int  mik_krige_execute(int device, const mik_problem *p, const mik_points *g, double *z_out, double *ss_out);

/* Test/diagnostic access: which = 0 -> kriging matrix A as assembled (only valid right after
 * mik_assemble_only), 1 -> A_inv after mik_factor.  out is M x M row-major. */
int  mik_assemble_only(mik_handle *h);
int  mik_get_matrix(mik_handle *h, int which, double *out);
int64_t mik_matrix_order(mik_handle *h); /* M = n + ndrift + 1 */
int64_t mik_points_resident(mik_handle *h);  /* unmasked points on the device(s) after mik_set_points / mik_set_grid */
int  mik_get_points(mik_handle *h, double *px_out, double *py_out, double *pz_out); /* those points' adjusted coordinates
                                            (diagnostic: what mik_set_grid generated), mik_points_resident() doubles each */
int  mik_get_timing(mik_handle *h, mik_timing *out);   /* device group: the leader's phases, predict_ms = slowest member */
int  mik_get_device_timing(mik_handle *h, int member, mik_timing *out); /* one member of a device group (0 = the handle's own device) */
int  mik_selftest_exp(int device, const double *x, double *out, int n); /* out[i] = the library's 19-instruction exp (x[i] <= 0: the moving
                                                                          window's matrix set-up), for comparison with the caller's exp */
int  mik_selftest_mfma(int device); /* 0 if the v_mfma_f64_4x4x4_4b_f64 (and 16x16x4) fragment layouts are what the kernels assume */

/* Multi-GPU (one process per GPU): grid points are sharded by the caller; the factored matrix is
 * broadcast from `root` over RCCL/xGMI.  The 128-byte id is an ncclUniqueId made on rank 0 and
 * carried to the other ranks by the host launcher. */
int  mik_comm_unique_id(char id_out[128]);
int  mik_comm_init(mik_handle *h, int nranks, int rank, const char id[128]);
int  mik_bcast_factor(mik_handle *h, int root); /* ranks != root need mik_set_problem first, not mik_factor */
int  mik_factor_checksum(mik_handle *h, uint64_t out[4]); /* order-independent checksums of the inverted matrix (2 words) and
                                                  of c (2 words) as this handle's device holds them: equal on every rank
                                                  after a sound broadcast */

/* Bounded waits.  Nothing in the multi-GPU paths can block for ever: communicator set-up (ncclCommInitAll /
 * ncclCommInitRank) is given MIK_RCCL_INIT_TIMEOUT seconds (default 120; option "rccl_init_timeout"), a broadcast of the
 * factor MIK_RCCL_BCAST_TIMEOUT (default 30; "rccl_bcast_timeout"), the peer scatter + all-gather MIK_PEER_TIMEOUT (default
 * 30; "peer_timeout").  When a limit expires the stuck call is abandoned on its worker thread together with the streams
 * and buffers it may still touch (leaked on purpose), RCCL is marked unusable for the rest of the process, and
 *   - a device group with exchange = auto goes on to peer copies, then to every member factoring the matrix itself;
 *     mik_timing.exchange_path / exchange_fallbacks and mik_exchange_note() say what happened;
 *   - a forced path ("exchange" 1 / 2), mik_comm_init and mik_bcast_factor return MIK_ERCCL / MIK_EHIP.
 * Every exchange of a device group is verified: each member checksums its copy of the inverse on its device against the
 * leader's; a mismatch counts as a failed exchange.  With "async_exchange" 1 (default; MIK_ASYNC_EXCHANGE) mik_factor
 * returns when the leader has factored; the exchange is joined by the next call on the handle (see the option). */
const char *mik_exchange_note(mik_handle *h);   /* why exchange paths of the last mik_factor were given up ("" if none were) */
int  mik_selftest_exchange(int members, double init_limit_s, double bcast_limit_s, char *report, int report_len);
                                                /* the RCCL path of the group exchange with stand-in members and NO HIP
                                                   call (runs without a GPU): drives the bounded waits against whatever
                                                   MIK_RCCL_LIB names */

const char *mik_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
