// comm.hip -- RCCL communicator wrapper for the row-sharded engines (one process per GPU,
// collectives over xGMI).  Replaces the reference's implicit PCT reductions
// (unwrappedadmm.m:118-122, 135-137; getProxOps.m:1255-1257, 1281-1284, 1321-1323).
//
// RCCL is bound at run time with dlopen so that (a) the library loads on machines without
// a GPU / RCCL, and (b) a host process that already loaded an RCCL (torch.distributed's
// "nccl" backend IS RCCL) shares that one instance instead of initialising a second copy.
#include <dlfcn.h>

#include <mutex>

#include "common.h"

using namespace admm;

namespace {

// minimal mirror of the NCCL/RCCL C API we use (stable since NCCL 2.x)
typedef struct ncclComm* ncclComm_t;
typedef struct {
  char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;  // 0 == ncclSuccess
enum { kNcclFloat64 = 8, kNcclSum = 0 };

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* nm : names) {
      api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);  // already in the process (torch)?
      if (api.handle) break;
    }
    for (const char* nm : names) {
      if (api.handle) break;
      api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!api.handle) {
      api.error = std::string("dlopen(librccl): ") + (dlerror() ? dlerror() : "not found");
      return;
    }
    auto sym = [&](const char* s) { return dlsym(api.handle, s); };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
    api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce)
      api.error = "librccl is missing ncclGetUniqueId/ncclCommInitRank/ncclCommDestroy/ncclAllReduce";
  });
  return api;
}

int rccl_fail(const char* what, ncclResult_t r) {
  RcclApi& a = rccl();
  return fail(ADMM_E_COMM, std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error") + " (" +
                               std::to_string(r) + ")");
}

}  // namespace

struct admm_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1, device = 0;
};

namespace admm {

int comm_nranks(admm_comm* c) { return c ? c->nranks : 1; }
int comm_rank(admm_comm* c) { return c ? c->rank : 0; }

// in-place sum all-reduce of `count` doubles on `stream` (no host sync)
int comm_allreduce_device(admm_comm* c, double* buf, size_t count, hipStream_t stream) {
  if (!c || c->nranks == 1) return ADMM_OK;
  ncclResult_t r = rccl().AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, c->comm, stream);
  if (r != 0) return rccl_fail("ncclAllReduce", r);
  return ADMM_OK;
}

}  // namespace admm

extern "C" {

int admm_comm_unique_id(char id[ADMM_COMM_ID_BYTES]) {
  if (!id) return fail(ADMM_E_INVALID, "id is NULL");
  RcclApi& a = rccl();
  if (!a.error.empty()) return fail(ADMM_E_COMM, a.error);
  static_assert(sizeof(ncclUniqueId) == ADMM_COMM_ID_BYTES, "unique id size");
  ncclUniqueId uid;
  ncclResult_t r = a.GetUniqueId(&uid);
  if (r != 0) return rccl_fail("ncclGetUniqueId", r);
  std::memcpy(id, uid.internal, ADMM_COMM_ID_BYTES);
  return ADMM_OK;
}

int admm_comm_init(const char id[ADMM_COMM_ID_BYTES], int rank, int nranks, int device, admm_comm** out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(ADMM_E_INVALID, "comm_init: bad argument");
  *out = nullptr;
  RcclApi& a = rccl();
  if (!a.error.empty()) return fail(ADMM_E_COMM, a.error);
  ADMM_HIP_TRY(hipSetDevice(device));
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, ADMM_COMM_ID_BYTES);
  admm_comm* c = new admm_comm();
  c->rank = rank;
  c->nranks = nranks;
  c->device = device;
  ncclResult_t r = a.CommInitRank(&c->comm, nranks, uid, rank);
  if (r != 0) {
    delete c;
    return rccl_fail("ncclCommInitRank", r);
  }
  *out = c;
  return ADMM_OK;
}

int admm_comm_allreduce_sum(admm_comm* comm, double* host_buf, size_t count) {
  if (!comm || !host_buf) return fail(ADMM_E_INVALID, "allreduce: NULL argument");
  ADMM_HIP_TRY(hipSetDevice(comm->device));
  double* d = nullptr;
  ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(double) * (count ? count : 1)));
  int rc = ADMM_OK;
  if (hipMemcpy(d, host_buf, sizeof(double) * count, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(ADMM_E_DEVICE, "hipMemcpy H2D");
  if (rc == ADMM_OK) rc = comm_allreduce_device(comm, d, count, nullptr);
  if (rc == ADMM_OK && hipDeviceSynchronize() != hipSuccess) rc = fail(ADMM_E_DEVICE, "hipDeviceSynchronize");
  if (rc == ADMM_OK && hipMemcpy(host_buf, d, sizeof(double) * count, hipMemcpyDeviceToHost) != hipSuccess)
    rc = fail(ADMM_E_DEVICE, "hipMemcpy D2H");
  (void)hipFree(d);
  return rc;
}

void admm_comm_destroy(admm_comm* comm) {
  if (!comm) return;
  if (comm->comm) (void)rccl().CommDestroy(comm->comm);
  delete comm;
}

}  // extern "C"
