// sfm_comm.hip — library-owned RCCL communicators (gfx950, one process per GPU).
//
// the only exchange step of sharded bundle adjustment is one all-reduce (SUM, double) of the
// packed reduced camera system [S | rhs] per iteration (0.54 MB at 50 cameras, 8.1 MB at 200).  With a communicator
// attached (sfm_ba_set_comm) sfm_ba_iterate issues it itself, on the problem's stream, between the partial reduce and the
// replicated solve: the whole K-iteration loop of a rank is one C call.
//
// RCCL is loaded with dlopen on first use and called through function pointers: the library keeps no link-time
// dependency on it (a Python process usually carries torch's own copy already; two copies bound at link time would
// fight over the same symbol names).
#include <dlfcn.h>

#include <cstring>

#include "sfm_ba.h"

struct sfm_comm {
  unsigned magic;
  void* nccl;        // ncclComm_t
  int world, rank;
  int attached;      // problems that hold it (sfm_ba_set_comm): it cannot be destroyed under them
};

namespace sfm {
namespace {

constexpr unsigned kCommMagic = 0x5F3C0221u;
// the few RCCL types and constants this file needs (rccl.h: ncclUniqueId is 128 opaque bytes, ncclFloat64 = 8, ncclSum = 0)
struct NcclUniqueId { char internal[128]; };
constexpr int kNcclFloat64 = 8, kNcclSum = 0;

struct Rccl {
  void* lib = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(void**, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool tried = false;
};

Rccl& rccl() { static Rccl r; return r; }

int load_rccl() {
  Rccl& r = rccl();
  if (r.lib) return SFM_OK;
  if (!r.tried) {
    r.tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.lib) break;
    }
    if (r.lib) {
      r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(r.lib, "ncclGetUniqueId"));
      r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(r.lib, "ncclCommInitRank"));
      r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(r.lib, "ncclCommDestroy"));
      r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(r.lib, "ncclAllReduce"));
      r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(r.lib, "ncclGetErrorString"));
      if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce) { dlclose(r.lib); r.lib = nullptr; }
    }
  }
  if (!r.lib) {
    set_error("RCCL could not be loaded (librccl.so.1 / librccl.so): %s", dlerror() ? dlerror() : "symbols missing");
    return SFM_E_RCCL;
  }
  return SFM_OK;
}

int rccl_fail(int code, const char* what) {
  Rccl& r = rccl();
  set_error("%s failed: %s (ncclResult %d)", what, r.GetErrorString ? r.GetErrorString(code) : "?", code);
  return SFM_E_RCCL;
}

}  // namespace

int comm_attach(sfm_comm* comm, int delta) {
  if (comm == nullptr) return SFM_OK;
  if (comm->magic != kCommMagic) { set_error("invalid communicator handle"); return SFM_E_HANDLE; }
  comm->attached += delta;
  return SFM_OK;
}

int comm_all_reduce_f64(sfm_comm* comm, double* buf, size_t count, hipStream_t s) {
  if (comm == nullptr || comm->magic != kCommMagic) { set_error("invalid communicator handle"); return SFM_E_HANDLE; }
  const int rc = rccl().AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, comm->nccl, s);
  if (rc != 0) return rccl_fail(rc, "ncclAllReduce");
  return SFM_OK;
}

}  // namespace sfm

using namespace sfm;

extern "C" {

int sfm_comm_available(void) {
  SFM_TRY(ensure_init());
  return load_rccl();
}

int sfm_comm_unique_id(char id_out[128]) {
  SFM_TRY(ensure_init());
  if (id_out == nullptr) { set_error("sfm_comm_unique_id: id_out is null"); return SFM_E_SHAPE; }
  SFM_TRY(load_rccl());
  NcclUniqueId id;
  const int rc = rccl().GetUniqueId(&id);
  if (rc != 0) return rccl_fail(rc, "ncclGetUniqueId");
  std::memcpy(id_out, id.internal, 128);
  return SFM_OK;
}

int sfm_comm_create(int world_size, int rank, const char id[128], sfm_comm** out) {
  SFM_TRY(ensure_init());
  if (out == nullptr || id == nullptr) { set_error("sfm_comm_create: null argument"); return SFM_E_SHAPE; }
  *out = nullptr;
  if (world_size < 1 || rank < 0 || rank >= world_size) {
    set_error("sfm_comm_create: rank %d outside a world of %d", rank, world_size);
    return SFM_E_SHAPE;
  }
  SFM_TRY(load_rccl());
  SFM_HIP(hipSetDevice(ctx().device));
  NcclUniqueId uid;
  std::memcpy(uid.internal, id, 128);
  void* nccl = nullptr;
  const int rc = rccl().CommInitRank(&nccl, world_size, uid, rank);
  if (rc != 0) return rccl_fail(rc, "ncclCommInitRank");
  *out = new sfm_comm{kCommMagic, nccl, world_size, rank, 0};
  return SFM_OK;
}

int sfm_comm_destroy(sfm_comm* comm) {
  if (comm == nullptr) return SFM_OK;
  if (comm->magic != kCommMagic) { set_error("sfm_comm_destroy: invalid handle"); return SFM_E_HANDLE; }
  if (comm->attached > 0) {
    set_error("sfm_comm_destroy: %d problem(s) still hold this communicator (sfm_ba_set_comm(p, NULL) or sfm_ba_destroy first)", comm->attached);
    return SFM_E_HANDLE;
  }
  if (ctx().inited) (void)hipDeviceSynchronize();
  const int rc = rccl().CommDestroy(comm->nccl);
  comm->magic = 0;
  delete comm;
  if (rc != 0) return rccl_fail(rc, "ncclCommDestroy");
  return SFM_OK;
}

}  // extern "C"
