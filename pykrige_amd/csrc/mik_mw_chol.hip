// mik_mw_chol.hip -- the thread-grid / register-tile classes of k_mw_chol (mik_k_mw_chol.h), {G, RI} covers K <= G * RI, each in the
// dynamic form and with the four variogram models as compile-time constants.  Built MIK_MWC_PARTS times with -DMIK_MWC_PART=0..4
// (pykrige_amd/build.py): every class is a 10 000 .. 100 000-instruction kernel, and together they are most of the library's compile time.
#include "mik_k_mw_chol.h"
#include "mik_host.h"

#ifndef MIK_MWC_PART
#error "compile with -DMIK_MWC_PART=0 .. MIK_MWC_PARTS - 1"
#endif

template <int G, int RI>
static int launch_mw_chol(hipStream_t stream, bool use_static, const MwArgs& a, long pc) {
  constexpr int T = G * G, NT = T < 256 ? 256 : T, PPB = NT / T, NB = G * RI;
  if (a.K > NB) return fail(MIK_EINVAL, "moving-window LDL^T class too small for this window");
  const size_t lds = sizeof(double) * (size_t)(2 * (NB + 4) + 9 * NB) * PPB;
  const dim3 grid((unsigned)((pc + PPB - 1) / PPB));
  // the variogram model as a compile-time constant where the problem allows it (Euclidean coordinates; the four models whose
  // shifted station block is positive definite and cheap): the set-up code of the kernel shrinks 20-fold (mw_entry_t)
#define MWC_LAUNCH(MODEL)                                                                                                   \
  do {                                                                                                                      \
    HIPC(hipFuncSetAttribute((const void*)k_mw_chol<G, RI, MODEL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));  \
    hipLaunchKernelGGL((k_mw_chol<G, RI, MODEL>), grid, dim3(NT), lds, stream, a);                                       \
  } while (0)
  const int sm = (a.mode == 1 || !use_static) ? -1 : a.v.model;
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


#define MWC(G, RI) case 100 * G + RI: return launch_mw_chol<G, RI>(stream, use_static, a, pc);
#define MWC_PART_FN(N) int mw_chol_part##N(int cls, hipStream_t stream, bool use_static, const MwArgs& a, long pc)
// The classes are dealt out by COMPILE TIME (seconds of device code generation per class, hipcc of ROCm 7.2 on one core: 1.5 s for {8, 4} ... 9.6 s for
// {16, 14}; 106 s together), 21 s a part: the parts are the critical path of a clean build.
#if MIK_MWC_PART == 0
MWC_PART_FN(0) {
  switch (cls) {
    MWC(4, 13) MWC(8, 12) MWC(16, 11)
    default: return MIK_MWC_NOCLASS;
  }
}
#elif MIK_MWC_PART == 1
MWC_PART_FN(1) {
  switch (cls) {
    MWC(16, 14) MWC(8, 11) MWC(16, 10)
    default: return MIK_MWC_NOCLASS;
  }
}
#elif MIK_MWC_PART == 2
MWC_PART_FN(2) {
  switch (cls) {
    MWC(8, 13) MWC(16, 12) MWC(4, 10)
    default: return MIK_MWC_NOCLASS;
  }
}
#elif MIK_MWC_PART == 3
MWC_PART_FN(3) {
  switch (cls) {
    MWC(16, 13) MWC(8, 10) MWC(16, 9) MWC(4, 8)
    default: return MIK_MWC_NOCLASS;
  }
}
#else
MWC_PART_FN(4) {
  switch (cls) {
    MWC(4, 4) MWC(4, 6) MWC(8, 4) MWC(8, 6) MWC(8, 8) MWC(16, 7) MWC(16, 8) MWC(32, 8)
    default: return MIK_MWC_NOCLASS;
  }
}
#endif
