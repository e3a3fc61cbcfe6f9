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
#include "mik_dev.h"
#include "mik_k_core.h"
#include "mik_k_predict.h"
#include "mik_k_inverse.h"
#include "mik_k_mw.h"
#include "mik_k_mw_solve.h"
