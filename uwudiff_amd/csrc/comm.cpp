// Data-parallel gradient exchange behind the C ABI (SURVEY.md section 8b "allreduce_flat"; the reference's exchange is
// Lightning DDP's bucketed NCCL all-reduce, configs/demo_training.yaml:5-7): one RCCL communicator per rank, created once,
// and a sum all-reduce of a slice of the flat fp32 gradient buffer on a caller-given stream -- no ProcessGroup stream hops.
// RCCL is resolved at run time (the librccl this process already has loaded -- PyTorch's -- else the system one), so
// libuwu_hip.so carries no link-time dependency on it and single-GPU use never touches it.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <string.h>

#include "../../include/uwu_hip.h"

void uwu_set_error(const char* fmt, ...);

namespace {
struct UniqueId { char internal[128]; };  // ncclUniqueId (rccl.h: NCCL_UNIQUE_ID_BYTES = 128)
typedef void* Comm;
struct Api {
  int (*get_unique_id)(UniqueId*);
  int (*comm_init_rank)(Comm*, int, UniqueId, int);
  int (*all_reduce)(const void*, void*, size_t, int, int, Comm, hipStream_t);
  int (*comm_destroy)(Comm);
  const char* (*get_error_string)(int);
  bool ok = false;
};
Api g_api;

bool load_api() {
  if (g_api.ok) return true;
  void* h = nullptr;
  for (const char* name : {"librccl.so", "librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_NOLOAD);  // the copy already mapped into this process (torch's), if any
    if (h) break;
  }
  if (!h)
    for (const char* name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
      h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (h) break;
    }
  if (!h) { uwu_set_error("comm: librccl.so not found (%s)", dlerror()); return false; }
  g_api.get_unique_id = reinterpret_cast<decltype(g_api.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
  g_api.comm_init_rank = reinterpret_cast<decltype(g_api.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
  g_api.all_reduce = reinterpret_cast<decltype(g_api.all_reduce)>(dlsym(h, "ncclAllReduce"));
  g_api.comm_destroy = reinterpret_cast<decltype(g_api.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
  g_api.get_error_string = reinterpret_cast<decltype(g_api.get_error_string)>(dlsym(h, "ncclGetErrorString"));
  if (!g_api.get_unique_id || !g_api.comm_init_rank || !g_api.all_reduce || !g_api.comm_destroy) {
    uwu_set_error("comm: RCCL symbols missing");
    return false;
  }
  g_api.ok = true;
  return true;
}
int fail(const char* what, int rc) {
  uwu_set_error("%s: RCCL error %d (%s)", what, rc, g_api.get_error_string ? g_api.get_error_string(rc) : "?");
  return UWU_ELAUNCH;
}
}  // namespace

// 128 bytes that rank 0 creates and every rank must receive (any side channel: torch.distributed broadcast, a file, ...)
extern "C" int uwu_comm_unique_id(void* id128) {
  if (!id128) { uwu_set_error("comm_unique_id: null"); return UWU_EINVAL; }
  if (!load_api()) return UWU_ENOTIMPL;
  UniqueId id;
  const int rc = g_api.get_unique_id(&id);
  if (rc) return fail("ncclGetUniqueId", rc);
  memcpy(id128, &id, sizeof(id));
  return UWU_OK;
}
// collective: every rank calls it once with the same id (the current HIP device is the rank's GPU)
extern "C" int uwu_comm_init(const void* id128, int rank, int world, void** comm) {
  if (!id128 || !comm || world < 1 || rank < 0 || rank >= world) { uwu_set_error("comm_init: bad argument"); return UWU_EINVAL; }
  if (!load_api()) return UWU_ENOTIMPL;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  Comm c = nullptr;
  const int rc = g_api.comm_init_rank(&c, world, id, rank);
  if (rc) return fail("ncclCommInitRank", rc);
  *comm = c;
  return UWU_OK;
}
// buf[0..n) (fp32, device) <- sum over ranks, in place, enqueued on `stream`
extern "C" int uwu_allreduce_flat(void* comm, float* buf, int64_t n, void* stream) {
  if (!comm || !buf || n <= 0) { uwu_set_error("allreduce_flat: bad argument"); return UWU_EINVAL; }
  if (!g_api.ok) { uwu_set_error("allreduce_flat: no communicator was created through uwu_comm_init"); return UWU_EINVAL; }
  const int rc = g_api.all_reduce(buf, buf, (size_t)n, /*ncclFloat32*/ 7, /*ncclSum*/ 0, comm, static_cast<hipStream_t>(stream));
  if (rc) return fail("ncclAllReduce", rc);
  return UWU_OK;
}
extern "C" int uwu_comm_destroy(void* comm) {
  if (!comm) return UWU_OK;
  if (!g_api.ok) { uwu_set_error("comm_destroy: RCCL not loaded"); return UWU_EINVAL; }
  const int rc = g_api.comm_destroy(comm);
  return rc ? fail("ncclCommDestroy", rc) : UWU_OK;
}
