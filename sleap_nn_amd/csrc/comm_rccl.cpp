// Gradient exchange of data-parallel training over RCCL, behind the C ABI (`ph_allreduce`).
//
// What it replaces: Lightning's DDP strategy in the reference's trainer (sleap_nn/training/model_trainer.py:1751-1813; docs/guides/multi-gpu.md:67-81): one process per GPU,
// identical replicas, the gradients averaged over the ranks every step.  Here the gradients of a replica are ONE flat arena (ph_model_backward), so the exchange is an
// in-place sum of that arena -- two buckets, the tail first -- on a side stream; the 1 / world factor rides in ph_adam_step's grad_scale.
//
// librccl is opened at run time (dlopen), not linked: the library loads on hosts without it (the build container's CPU tests), and a process that never trains on more
// than one GPU never touches it.  The communicator is created from a 128-byte unique id that rank 0 draws (ph_comm_unique_id) and the host language hands to the other
// ranks over whatever channel it has (torch.distributed's store / broadcast in the Python host: sleap_nn_amd/parallel.py).
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>

#include "common.h"

namespace {

// the part of rccl.h this file uses (RCCL 2.x ABI: ncclUniqueId is 128 opaque bytes passed BY VALUE; ncclFloat = 7, ncclSum = 0)
struct UniqueId {
  char internal[128];
};
typedef void* Comm;
typedef int (*GetUniqueIdFn)(UniqueId*);
typedef int (*CommInitRankFn)(Comm*, int, UniqueId, int);
typedef int (*AllReduceFn)(const void*, void*, size_t, int, int, Comm, hipStream_t);
typedef int (*CommDestroyFn)(Comm);
typedef const char* (*GetErrorStringFn)(int);
typedef int (*CommCountFn)(Comm, int*);

struct Rccl {
  void* handle = nullptr;
  GetUniqueIdFn get_unique_id = nullptr;
  CommInitRankFn comm_init_rank = nullptr;
  AllReduceFn all_reduce = nullptr;
  CommDestroyFn comm_destroy = nullptr;
  GetErrorStringFn get_error_string = nullptr;
  CommCountFn comm_count = nullptr;
  bool tried = false;
};

Rccl& rccl() {
  static Rccl r;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  if (r.tried) return r;
  r.tried = true;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
    r.handle = dlopen(name, RTLD_NOW | RTLD_LOCAL);
    if (r.handle) break;
  }
  if (!r.handle) return r;
  r.get_unique_id = reinterpret_cast<GetUniqueIdFn>(dlsym(r.handle, "ncclGetUniqueId"));
  r.comm_init_rank = reinterpret_cast<CommInitRankFn>(dlsym(r.handle, "ncclCommInitRank"));
  r.all_reduce = reinterpret_cast<AllReduceFn>(dlsym(r.handle, "ncclAllReduce"));
  r.comm_destroy = reinterpret_cast<CommDestroyFn>(dlsym(r.handle, "ncclCommDestroy"));
  r.get_error_string = reinterpret_cast<GetErrorStringFn>(dlsym(r.handle, "ncclGetErrorString"));
  r.comm_count = reinterpret_cast<CommCountFn>(dlsym(r.handle, "ncclCommCount"));
  if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) {
    dlclose(r.handle);
    r.handle = nullptr;
  }
  return r;
}

const char* rccl_error(const Rccl& r, int code) { return r.get_error_string ? r.get_error_string(code) : "RCCL error"; }

}  // namespace

struct ph_comm {
  Comm comm = nullptr;
  int world = 1, rank = 0;
};

extern "C" {

int ph_comm_available(void) { return rccl().handle ? 1 : 0; }

int ph_comm_unique_id(void* out_id128) {
  PH_REQUIRE(out_id128, "ph_comm_unique_id: null argument");
  Rccl& r = rccl();
  PH_REQUIRE(r.handle, "ph_comm_unique_id: librccl could not be opened (%s)", dlerror() ? dlerror() : "not found");
  UniqueId id;
  const int rc = r.get_unique_id(&id);
  if (rc != 0) {
    ph::set_error("ncclGetUniqueId failed: %s", rccl_error(r, rc));
    return PH_E_HIP;
  }
  std::memcpy(out_id128, id.internal, sizeof(id.internal));
  return PH_OK;
}

ph_comm* ph_comm_create(const void* id128, int32_t world, int32_t rank) {
  if (!id128 || world < 1 || rank < 0 || rank >= world) {
    ph::set_error("ph_comm_create: bad arguments (world %d, rank %d)", world, rank);
    return nullptr;
  }
  Rccl& r = rccl();
  if (!r.handle) {
    ph::set_error("ph_comm_create: librccl could not be opened");
    return nullptr;
  }
  UniqueId id;
  std::memcpy(id.internal, id128, sizeof(id.internal));
  ph_comm* c = new ph_comm();
  c->world = world;
  c->rank = rank;
  const int rc = r.comm_init_rank(&c->comm, world, id, rank);  // collective: returns when every rank has joined (on the current HIP device)
  if (rc != 0) {
    ph::set_error("ncclCommInitRank(world %d, rank %d) failed: %s", world, rank, rccl_error(r, rc));
    delete c;
    return nullptr;
  }
  return c;
}

void ph_comm_destroy(ph_comm* c) {
  if (!c) return;
  Rccl& r = rccl();
  if (r.handle && c->comm) (void)r.comm_destroy(c->comm);
  delete c;
}

int32_t ph_comm_world(const ph_comm* c) { return c ? c->world : 0; }

int ph_allreduce(ph_comm* c, float* buf_dev, int64_t count, void* stream) {
  PH_REQUIRE(c && c->comm && buf_dev && count >= 0, "ph_allreduce: bad arguments");
  if (count == 0) return PH_OK;
  Rccl& r = rccl();
  PH_REQUIRE(r.handle, "ph_allreduce: librccl is not open");
  const int rc = r.all_reduce(buf_dev, buf_dev, (size_t)count, /*ncclFloat*/ 7, /*ncclSum*/ 0, c->comm, static_cast<hipStream_t>(stream));
  if (rc != 0) {
    ph::set_error("ncclAllReduce(%lld floats) failed: %s", (long long)count, rccl_error(r, rc));
    return PH_E_HIP;
  }
  return PH_OK;
}

}  // extern "C"
