// RCCL all-reduce of the shared block of an epoch-sharded joint fit, called from the library's own loop
// (include/lcmi.h, "RCCL group").
//
// SURVEY.md 8(e) / BASELINE.json north_star: "RCCL all-reduce over xGMI only for the shared-background gradient in the joint
// fit".  The reference keeps all epochs on one device (lightcurver/processes/roi_modelling.py:154-160,213) and has no
// counterpart.  lc_joint_run_sharded takes the all-reduce as a callback; with torch.distributed's nccl backend that callback
// used to be a Python function (ctypes -> torch.distributed.all_reduce under an ExternalStream) entered once per 65 - 90 us
// iteration.  Here the library owns a communicator itself: librccl is loaded at run time (dlopen - the library does not link
// against it, and a process that already carries a copy, e.g. torch's, shares that one), the communicator is built from a
// unique id the caller distributes over whatever channel it has (lc_rccl_unique_id on rank 0, then any host broadcast), and
// lc_rccl_allreduce - which has the callback's signature - enqueues ncclAllReduce in place on the library's stream.  No
// Python, no host synchronisation inside the loop.
#include <dlfcn.h>

#include <cstring>

#include "lc_common.h"

namespace {

constexpr int kUniqueIdBytes = 128;  // NCCL_UNIQUE_ID_BYTES
struct RcclUniqueId {
  char internal[kUniqueIdBytes];
};
typedef void *RcclComm;
typedef int (*get_unique_id_fn)(RcclUniqueId *);
typedef int (*comm_init_rank_fn)(RcclComm *, int, RcclUniqueId, int);
typedef int (*comm_destroy_fn)(RcclComm);
typedef int (*all_reduce_fn)(const void *, void *, size_t, int, int, RcclComm, hipStream_t);
typedef const char *(*get_error_string_fn)(int);
constexpr int kRcclFloat32 = 7, kRcclSum = 0;  // ncclFloat32, ncclSum (rccl.h)

struct RcclApi {
  void *dl = nullptr;
  get_unique_id_fn get_unique_id = nullptr;
  comm_init_rank_fn comm_init_rank = nullptr;
  comm_destroy_fn comm_destroy = nullptr;
  all_reduce_fn all_reduce = nullptr;
  get_error_string_fn get_error_string = nullptr;
  std::string err;
};

// one copy of the library per process: the one already loaded (torch carries its own) before a new one
RcclApi &rccl_api() {
  static RcclApi api;
  if (api.dl || !api.err.empty()) return api;
  const char *names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names)
    if ((api.dl = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_LOCAL))) break;
  if (!api.dl)
    for (const char *n : names)
      if ((api.dl = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!api.dl) {
    api.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : "dlopen failed");
    return api;
  }
  api.get_unique_id = (get_unique_id_fn)dlsym(api.dl, "ncclGetUniqueId");
  api.comm_init_rank = (comm_init_rank_fn)dlsym(api.dl, "ncclCommInitRank");
  api.comm_destroy = (comm_destroy_fn)dlsym(api.dl, "ncclCommDestroy");
  api.all_reduce = (all_reduce_fn)dlsym(api.dl, "ncclAllReduce");
  api.get_error_string = (get_error_string_fn)dlsym(api.dl, "ncclGetErrorString");
  if (!api.get_unique_id || !api.comm_init_rank || !api.comm_destroy || !api.all_reduce) {
    api.err = "librccl: a required entry point is missing";
    api.dl = nullptr;
  }
  return api;
}

std::string rccl_error(const RcclApi &api, const char *what, int code) {
  std::string s = std::string(what) + ": RCCL error " + std::to_string(code);
  if (api.get_error_string) s += std::string(" (") + api.get_error_string(code) + ")";
  return s;
}

}  // namespace

struct lc_rccl_group {
  lc_ctx *ctx = nullptr;
  RcclComm comm = nullptr;
  int rank = 0, world = 1;
  long long calls = 0;
};

extern "C" {

int lc_rccl_available(void) { return rccl_api().dl != nullptr ? 1 : 0; }

int lc_rccl_unique_id(void *id_out, int id_bytes) {
  if (!id_out || id_bytes < kUniqueIdBytes) return LC_ERR_INVALID;
  RcclApi &api = rccl_api();
  if (!api.dl) return LC_ERR_UNSUPPORTED;
  RcclUniqueId id;
  if (api.get_unique_id(&id) != 0) return LC_ERR_DEVICE;
  std::memset(id_out, 0, (size_t)id_bytes);
  std::memcpy(id_out, &id, sizeof(id));
  return LC_OK;
}

int lc_rccl_group_create(lc_ctx *ctx, const void *unique_id, int id_bytes, int rank, int world, lc_rccl_group **out) {
  if (!ctx || !out || !unique_id || id_bytes < kUniqueIdBytes || world < 1 || rank < 0 || rank >= world) {
    if (ctx) ctx->err = "lc_rccl_group_create: invalid argument";
    return LC_ERR_INVALID;
  }
  LC_ENTER(ctx);
  RcclApi &api = rccl_api();
  if (!api.dl) LC_FAIL(ctx, LC_ERR_UNSUPPORTED, api.err.c_str());
  RcclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  lc_rccl_group *g = new lc_rccl_group();
  g->ctx = ctx;
  g->rank = rank;
  g->world = world;
  const int rc = api.comm_init_rank(&g->comm, world, id, rank);
  if (rc != 0) {
    ctx->err = rccl_error(api, "lc_rccl_group_create (ncclCommInitRank)", rc);
    delete g;
    return LC_ERR_DEVICE;
  }
  *out = g;
  return LC_OK;
}

// matches lc_allreduce_fn (user = the group): the callback of lc_joint_run_sharded, or called directly
int lc_rccl_allreduce(void *user, void *dev_buf, int count, void *hip_stream) {
  lc_rccl_group *g = (lc_rccl_group *)user;
  if (!g || !dev_buf || count <= 0) return LC_ERR_INVALID;
  LC_ENTER(g->ctx);
  RcclApi &api = rccl_api();
  const int rc = api.all_reduce(dev_buf, dev_buf, (size_t)count, kRcclFloat32, kRcclSum, g->comm, (hipStream_t)hip_stream);
  if (rc != 0) {
    g->ctx->err = rccl_error(api, "lc_rccl_allreduce (ncclAllReduce)", rc);
    return LC_ERR_DEVICE;
  }
  g->calls += 1;
  return LC_OK;
}

int lc_rccl_group_info(lc_rccl_group *g, int *rank, int *world, long long *calls) {
  if (!g) return LC_ERR_INVALID;
  if (rank) *rank = g->rank;
  if (world) *world = g->world;
  if (calls) *calls = g->calls;
  return LC_OK;
}

void lc_rccl_group_destroy(lc_rccl_group *g) {
  if (!g) return;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->stream);
  RcclApi &api = rccl_api();
  if (api.dl && g->comm) (void)api.comm_destroy(g->comm);
  delete g;
}

}  // extern "C"
