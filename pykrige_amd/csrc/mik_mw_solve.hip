// mik_mw_solve.hip -- moving-window kriging: the register classes of k_mw_solve (Gauss-Jordan with partial pivoting) and their dispatch.
// One translation unit of libmikrige.so (pykrige_amd/build.py compiles them in parallel).
#include "mik_k_mw_solve.h"
#include "mik_host.h"

// (Round 6: only the pivoting form is instantiated.  The six non-pivoting instantiations were reachable through the A/B option "mw_solver" alone --
// the default solver of a positive definite window is the LDL^T kernel -- and were half of this unit's compile time.)
template <int GY, int GX, int RI, int CJ>
static int launch_mw_solve(mik_handle* h, const MwArgs& a, long pc) {
  constexpr int T = GY * GX, PPB = 256 / T, CJP = (CJ + 1) & ~1;
  const int nb = a.K + 1;
  if (nb > GY * RI || nb + 1 > GX * CJ) return fail(MIK_EINVAL, "moving-window solve class too small for this window");
  const size_t per = (2 * ((size_t)GX * CJP + (size_t)GY * RI) + 16 + 5 * (size_t)nb + (2 * (size_t)nb + 1) / 2 + 1) & ~(size_t)1;
  const size_t lds = sizeof(double) * per * PPB;
  const dim3 grid((unsigned)((pc + PPB - 1) / PPB));
  HIPC(hipFuncSetAttribute((const void*)k_mw_solve<GY, GX, RI, CJ, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL((k_mw_solve<GY, GX, RI, CJ, true>), grid, dim3(256), lds, h->stream, a);
  return MIK_OK;
}

// thread-grid / register-tile classes of k_mw_solve, {GY, GX, RI, CJ} covers nb <= GY*RI and nb + 1 <= GX*CJ.  Measured
// on MI355X (scripts/mw_classes.py history in DESIGN.md): the classes whose tile fits the VGPR file without AGPR spills
// win, and among those the one with the fewest threads per point.
int dispatch_mw_solve(mik_handle* h, const MwArgs& a, long pc) {
  const int nb = a.K + 1;
  if (nb <= 16) return launch_mw_solve<4, 4, 4, 5>(h, a, pc);   // 16 threads per point
  if (nb <= 32) return launch_mw_solve<8, 8, 4, 5>(h, a, pc);   // 64
  if (nb <= 48) return launch_mw_solve<8, 8, 6, 7>(h, a, pc);   // 64
  if (nb <= 64) return launch_mw_solve<8, 8, 8, 9>(h, a, pc);   // 64
  if (nb <= 96) return launch_mw_solve<16, 16, 6, 7>(h, a, pc); // 256
  return launch_mw_solve<16, 16, 8, 9>(h, a, pc);               // 256, nb <= 128
}
