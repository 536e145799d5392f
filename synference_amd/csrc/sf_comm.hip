// sf_comm.hip -- the data-parallel exchange step of training: ONE sum all-reduce of the flat fp32 gradient per optimiser
// step, over RCCL (xGMI inside a node), issued on the library's stream between the gradient gather and clip + Adam
// (the reference trains on CPU threads only -- examples/sbi/slurm/train_final_model.slurm:26 -- and has no
// counterpart: the call sits where custom_runner.py:585-618 has `loss.backward()` -> `optimizer.step()`).
//
// RCCL is bound at RUN time (dlopen + dlsym), never at link time: a process that already holds a RCCL -- PyTorch-ROCm ships
// its own librccl.so and torch.distributed's "nccl" backend lives in it -- must not get a second copy interposed on the same
// symbols, and a host without RCCL must still load this library for everything that does not shard.  <rccl/rccl.h> is used
// for types and prototypes only.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "sf_internal.h"
#include "synference_hip.h"

struct sf_comm {
  ncclComm_t comm = nullptr;
  int nranks = 0, rank = 0, device = 0;
};

namespace {
struct RcclApi {
  void* handle = nullptr;
  std::string path, error;
  decltype(&ncclGetUniqueId) get_unique_id = nullptr;
  decltype(&ncclCommInitRank) comm_init_rank = nullptr;
  decltype(&ncclCommDestroy) comm_destroy = nullptr;
  decltype(&ncclAllReduce) all_reduce = nullptr;
  decltype(&ncclGetErrorString) error_string = nullptr;
  decltype(&ncclGetVersion) get_version = nullptr;
};
std::mutex g_mu;
RcclApi g_api;
std::string g_user_path;

bool bind(RcclApi& a, void* h, const std::string& path) {
#define SF_SYM(field, name)                                                   \
  a.field = reinterpret_cast<decltype(a.field)>(dlsym(h, name));              \
  if (!a.field) { a.error = path + ": missing symbol " name; return false; }
  SF_SYM(get_unique_id, "ncclGetUniqueId")
  SF_SYM(comm_init_rank, "ncclCommInitRank")
  SF_SYM(comm_destroy, "ncclCommDestroy")
  SF_SYM(all_reduce, "ncclAllReduce")
  SF_SYM(error_string, "ncclGetErrorString")
  SF_SYM(get_version, "ncclGetVersion")
#undef SF_SYM
  a.handle = h;
  a.path = path;
  return true;
}

// Order: the caller's path (sf_comm_set_library / SF_RCCL_LIB) -> a RCCL this process has ALREADY loaded (RTLD_NOLOAD: the
// one torch.distributed uses) -> the loader's search path -> the ROCm default location.
const RcclApi* api() {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_api.handle) return &g_api;
  std::string tried;
  auto attempt = [&](const char* name, int flags) -> bool {
    void* h = dlopen(name, flags);
    if (!h) {
      if (!(flags & RTLD_NOLOAD)) { const char* e = dlerror(); tried += std::string(name) + " (" + (e ? e : "?") + "); "; }
      return false;
    }
    RcclApi a;
    if (bind(a, h, name)) { g_api = a; return true; }
    tried += a.error + "; ";
    dlclose(h);
    return false;
  };
  const char* env = std::getenv("SF_RCCL_LIB");
  const std::string user = !g_user_path.empty() ? g_user_path : std::string(env ? env : "");
  if (!user.empty() && attempt(user.c_str(), RTLD_NOW | RTLD_LOCAL)) return &g_api;
  for (const char* n : {"librccl.so.1", "librccl.so"})
    if (attempt(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD)) return &g_api;
  for (const char* n : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
    if (attempt(n, RTLD_NOW | RTLD_LOCAL)) return &g_api;
  g_api.error = "RCCL could not be loaded: " + tried;
  return nullptr;
}

int rccl_fail(const RcclApi* a, ncclResult_t r, const char* what) {
  sf_set_error(std::string(what) + ": " + (a && a->error_string ? a->error_string(r) : "RCCL error") + " (" + std::to_string((int)r) + ")");
  return SF_ERR_HIP;
}
int no_rccl() {
  sf_set_error(g_api.error.empty() ? std::string("RCCL could not be loaded") : g_api.error);
  return SF_ERR_STATE;
}
}  // namespace

int sf_comm_all_reduce_impl(sf_comm* c, float* buf, long n, hipStream_t st) {
  const RcclApi* a = api();
  if (!a) return no_rccl();
  ncclResult_t r = a->all_reduce(buf, buf, (size_t)n, ncclFloat, ncclSum, c->comm, st);
  if (r != ncclSuccess) return rccl_fail(a, r, "ncclAllReduce");
  return SF_OK;
}

extern "C" {

int sf_comm_set_library(const char* path) {
  std::lock_guard<std::mutex> lock(g_mu);
  if (g_api.handle) {
    sf_set_error("RCCL is already bound (" + g_api.path + ")");
    return SF_ERR_STATE;
  }
  g_user_path = path ? path : "";
  return SF_OK;
}

int sf_comm_library(char* out, int64_t cap, int* version) {
  const RcclApi* a = api();
  if (!a) return no_rccl();
  if (out && cap > 0) {
    std::strncpy(out, a->path.c_str(), (size_t)cap - 1);
    out[cap - 1] = 0;
  }
  if (version) {
    int v = 0;
    if (a->get_version(&v) != ncclSuccess) v = 0;
    *version = v;
  }
  return SF_OK;
}

int sf_comm_unique_id(void* id_out, int64_t bytes) {
  if (!id_out || bytes < (int64_t)sizeof(ncclUniqueId)) {
    sf_set_error("sf_comm_unique_id: the buffer must hold SF_COMM_ID_BYTES (128) bytes");
    return SF_ERR_INVALID;
  }
  const RcclApi* a = api();
  if (!a) return no_rccl();
  ncclUniqueId id;
  ncclResult_t r = a->get_unique_id(&id);
  if (r != ncclSuccess) return rccl_fail(a, r, "ncclGetUniqueId");
  std::memcpy(id_out, &id, sizeof(id));
  return SF_OK;
}

int sf_comm_create(const void* id, int64_t bytes, int nranks, int rank, sf_comm** out) {
  if (!id || !out || bytes < (int64_t)sizeof(ncclUniqueId) || nranks < 1 || rank < 0 || rank >= nranks) {
    sf_set_error("sf_comm_create: bad id, nranks or rank");
    return SF_ERR_INVALID;
  }
  const RcclApi* a = api();
  if (!a) return no_rccl();
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) {
    sf_set_error("sf_comm_create: no current HIP device");
    return SF_ERR_NO_DEVICE;
  }
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  sf_comm* c = new sf_comm();
  c->nranks = nranks; c->rank = rank; c->device = dev;
  ncclResult_t r = a->comm_init_rank(&c->comm, nranks, uid, rank);   // (collective: every rank of the group enters it)
  if (r != ncclSuccess) {
    delete c;
    return rccl_fail(a, r, "ncclCommInitRank");
  }
  *out = c;
  return SF_OK;
}

void sf_comm_destroy(sf_comm* c) {
  if (!c) return;
  const RcclApi* a = api();
  if (a && c->comm) (void)a->comm_destroy(c->comm);
  delete c;
}

int sf_comm_info(const sf_comm* c, int* nranks, int* rank) {
  if (!c) {
    sf_set_error("null communicator");
    return SF_ERR_INVALID;
  }
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->rank;
  return SF_OK;
}

int sf_comm_all_reduce_sum(sf_comm* c, float* buf, int64_t n, void* stream) {
  if (!c || (!buf && n > 0) || n < 0) {
    sf_set_error("sf_comm_all_reduce_sum: null argument");
    return SF_ERR_INVALID;
  }
  if (n == 0) return SF_OK;
  return sf_comm_all_reduce_impl(c, buf, (long)n, (hipStream_t)stream);
}

}  // extern "C"
