// standin_rccl.cpp -- a stand-in for librccl.so used by the tests of the bounded factor exchange (MIK_RCCL_LIB points at
// it).  TEST INFRASTRUCTURE: never loaded by the product unless that variable says so.
//   g++   -shared -fPIC standin_rccl.cpp -o libstandin_rccl_cpu.so                 (no HIP: the copies are skipped)
//   hipcc -shared -fPIC -DWITH_HIP standin_rccl.cpp -o libstandin_rccl_hip.so       (broadcast = device-to-device copy)
// STANDIN_RCCL_MODE: ok | fail_init | hang_init | hang_bcast | corrupt   (corrupt: the copy that rank 1 receives is damaged)
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <chrono>
#ifdef WITH_HIP
#include <hip/hip_runtime.h>
#endif

extern "C" {
typedef void* ncclComm_t;
typedef int ncclResult_t;
typedef struct { char internal[128]; } ncclUniqueId;

static const char* mode() {
  const char* m = getenv("STANDIN_RCCL_MODE");
  return m ? m : "ok";
}
static void hang() {
  for (;;) std::this_thread::sleep_for(std::chrono::seconds(3600));
}
static std::mutex g_m;
static std::map<size_t, const void*> g_root;  // element count -> the root's buffer of the broadcast in flight

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  memset(id, 7, sizeof *id);
  return 0;
}
ncclResult_t ncclCommInitRank(ncclComm_t* c, int, ncclUniqueId, int rank) {
  if (!strcmp(mode(), "hang_init")) hang();
  if (!strcmp(mode(), "fail_init")) return 1;
  *c = (ncclComm_t)(intptr_t)(rank + 1);
  return 0;
}
ncclResult_t ncclCommInitAll(ncclComm_t* c, int n, const int*) {
  if (!strcmp(mode(), "hang_init")) hang();
  if (!strcmp(mode(), "fail_init")) return 1;
  for (int i = 0; i < n; ++i) c[i] = (ncclComm_t)(intptr_t)(i + 1);
  return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t) { return 0; }
const char* ncclGetErrorString(ncclResult_t) { return "stand-in RCCL error"; }
ncclResult_t ncclGroupStart() { return 0; }
ncclResult_t ncclGroupEnd() {
  if (!strcmp(mode(), "hang_bcast")) hang();
  return 0;
}
ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, int /*dtype: doubles*/, int root, ncclComm_t comm, void* stream) {
  const int rank = (int)(intptr_t)comm - 1;
  std::lock_guard<std::mutex> lk(g_m);
  if (rank == root) {
    g_root[count] = send;
    return 0;
  }
  if (!g_root.count(count)) return 2;  // a member's call before the root's
  const void* src = g_root[count];
#ifdef WITH_HIP
  if (hipMemcpyAsync(recv, src, count * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return 3;
  if (!strcmp(mode(), "corrupt") && rank == 1 && count > 64)
    if (hipMemsetAsync((char*)recv + (count / 2) * 8, 0x5A, 8, (hipStream_t)stream) != hipSuccess) return 3;
#else
  (void)recv;
  (void)src;
  (void)stream;
#endif
  return 0;
}
}
