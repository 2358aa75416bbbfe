// finrom_comm_*: the one exchange step of the path ("an RCCL gather over xGMI at the end") behind the C ABI, for callers
// that do not bring torch.distributed -- a NumPy / ctypes caller in the reference's own style (deep_learning/generate_fin_dataset.py
// :62-111 run per rank).  RCCL is loaded lazily with dlopen at the first finrom_comm_* call: the library has no link-time dependency
// on librccl, single-GPU use never touches it, and a process that already holds an RCCL (PyTorch's) gets that same copy.
#include "finrom_core.h"

#include <cstring>
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace finrom {
namespace {

struct Rccl {
  void* so = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;

int load_rccl() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.so) return 0;
  void* so = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (so) break;
  }
  if (!so) { set_error(std::string("finrom_comm: librccl could not be loaded: ") + dlerror()); return FINROM_ERR_UNSUPPORTED; }
  Rccl r;
  r.so = so;
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(so, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(so, "ncclCommInitRank");
  r.AllGather = (decltype(r.AllGather))dlsym(so, "ncclAllGather");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(so, "ncclCommDestroy");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(so, "ncclGetErrorString");
  if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy || !r.GetErrorString) {
    set_error("finrom_comm: librccl lacks an expected symbol");
    return FINROM_ERR_UNSUPPORTED;
  }
  g_rccl = r;
  return 0;
}
int rccl_fail(ncclResult_t e, const char* what) {
  set_error(std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(e) : "RCCL error"));
  return FINROM_ERR_HIP;
}

}  // namespace
}  // namespace finrom

using namespace finrom;

struct finrom_comm_s { ncclComm_t comm = nullptr; int rank = 0, nranks = 1; };

extern "C" {

int finrom_comm_unique_id(void* id_out) {
  if (!id_out) { set_error("comm_unique_id: null argument"); return FINROM_ERR_ARG; }
  static_assert(sizeof(ncclUniqueId) == FINROM_COMM_ID_BYTES, "id size");
  if (int rc = load_rccl()) return rc;
  ncclUniqueId id;
  const ncclResult_t e = g_rccl.GetUniqueId(&id);
  if (e != ncclSuccess) return rccl_fail(e, "ncclGetUniqueId");
  std::memcpy(id_out, &id, sizeof(id));
  return 0;
}

int finrom_comm_init(finrom_comm_t* out, int32_t rank, int32_t nranks, const void* id) {
  if (!out || !id || nranks < 1 || rank < 0 || rank >= nranks) { set_error("comm_init: bad argument"); return FINROM_ERR_ARG; }
  *out = nullptr;
  if (int rc = load_rccl()) return rc;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  auto* c = new finrom_comm_s();
  c->rank = rank; c->nranks = nranks;
  const ncclResult_t e = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);      // the communicator binds to the CURRENT device
  if (e != ncclSuccess) { delete c; return rccl_fail(e, "ncclCommInitRank"); }
  *out = c;
  return 0;
}

int finrom_gather(finrom_comm_t c, const double* send, int64_t count, double* recv, void* stream) {
  if (!c || count < 0 || (count > 0 && (!send || !recv))) { set_error("gather: bad argument"); return FINROM_ERR_ARG; }
  if (count == 0) return 0;
  const ncclResult_t e = g_rccl.AllGather(send, recv, (size_t)count, ncclFloat64, c->comm, (hipStream_t)stream);
  if (e != ncclSuccess) return rccl_fail(e, "ncclAllGather");
  return 0;
}

int finrom_comm_destroy(finrom_comm_t c) {
  if (!c) return 0;
  ncclResult_t e = ncclSuccess;
  if (c->comm && g_rccl.CommDestroy) e = g_rccl.CommDestroy(c->comm);
  delete c;
  if (e != ncclSuccess) return rccl_fail(e, "ncclCommDestroy");
  return 0;
}

}  // extern "C"
