// comm.hip -- communicator for the row-sharded engines (one process per GPU).  Replaces the
// reference's implicit PCT reductions (unwrappedadmm.m:118-122, 135-137; getProxOps.m:1255-1257,
// 1281-1284, 1321-1323) with an explicit sum all-reduce.
//
// Two transports behind one handle:
//   RCCL   collectives over xGMI, enqueued on the engine's stream (no host sync).  RCCL is bound
//          at run time with dlopen so that (a) the library loads on machines without a GPU or
//          RCCL, and (b) a host process that already loaded an RCCL (torch.distributed's "nccl"
//          backend IS RCCL) shares that instance instead of initialising a second copy.
//   SHM    host-staged all-reduce through a POSIX shared-memory segment: device -> pinned slot,
//          process barrier, rank-ordered sum, -> device.  Slow (one host sync per call) but it
//          needs nothing except a shared /dev/shm, so several ranks can share ONE GPU: it is how
//          the sharded engines are tested on a single-GPU box, and the fallback when RCCL
//          refuses the topology.
#include <dlfcn.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>

#include "common.h"

using namespace admm;

namespace {

// ---- RCCL (NCCL C API, stable since NCCL 2.x) ------------------------------------------
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
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;  // optional
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
      const char* de = dlerror();
      api.error = std::string("dlopen(librccl): ") + (de ? de : "not found");
      return;
    }
    auto sym = [&](const char* s) { return dlsym(api.handle, s); };
    api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
    api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
    api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
    api.CommAbort = reinterpret_cast<decltype(api.CommAbort)>(sym("ncclCommAbort"));
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

// ---- SHM transport ---------------------------------------------------------------------
constexpr size_t kShmSlotElems = 1u << 20;  // 8 MiB of doubles per rank and round

struct ShmHeader {
  std::atomic<uint64_t> arrived;  // monotone arrival counter (barrier k completes at k*nranks)
  std::atomic<uint64_t> attached;
  std::atomic<uint64_t> aborted;  // a rank of the group failed outside a collective: every barrier gives up (comm_abort)
  uint64_t pad[5];
};

struct ShmState {
  int fd = -1;
  void* base = nullptr;
  size_t bytes = 0;
  std::string name;
  uint64_t barriers = 0;
  std::vector<double> tmp;
};

std::string shm_name(const char* id) {
  uint64_t h = 1469598103934665603ull;  // FNV-1a over the 128 id bytes
  for (int i = 0; i < ADMM_COMM_ID_BYTES; ++i) {
    h ^= static_cast<unsigned char>(id[i]);
    h *= 1099511628211ull;
  }
  char buf[64];
  std::snprintf(buf, sizeof(buf), "/admm_hip_%016llx", static_cast<unsigned long long>(h));
  return buf;
}

}  // namespace

struct admm_comm {
  int transport = ADMM_COMM_RCCL;
  ncclComm_t comm = nullptr;
  ShmState shm;
  int rank = 0, nranks = 1, device = 0;
};

namespace {

int shm_barrier(admm_comm* c) {
  ShmHeader* h = static_cast<ShmHeader*>(c->shm.base);
  c->shm.barriers += 1;
  const uint64_t target = c->shm.barriers * static_cast<uint64_t>(c->nranks);
  h->arrived.fetch_add(1, std::memory_order_acq_rel);
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (h->arrived.load(std::memory_order_acquire) < target) {
    if (h->aborted.load(std::memory_order_acquire) != 0)
      return fail(ADMM_E_COMM, "a peer rank reported a failure: the collective was abandoned (the communicators of "
                               "this group are unusable now)");
    if (++spins > 2000) {
      std::this_thread::sleep_for(std::chrono::microseconds(50));
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
        return fail(ADMM_E_COMM, "shared-memory barrier timed out (a peer rank died?)");
    }
  }
  return ADMM_OK;
}

int shm_attach(admm_comm* c, const char* id) {
  ShmState& s = c->shm;
  s.name = shm_name(id);
  s.bytes = sizeof(ShmHeader) + static_cast<size_t>(c->nranks) * kShmSlotElems * sizeof(double);
  if (c->rank == 0) {
    shm_unlink(s.name.c_str());
    s.fd = shm_open(s.name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (s.fd < 0) return fail(ADMM_E_COMM, "shm_open(create) failed for " + s.name);
    if (ftruncate(s.fd, static_cast<off_t>(s.bytes)) != 0) return fail(ADMM_E_COMM, "ftruncate failed on " + s.name);
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      s.fd = shm_open(s.name.c_str(), O_RDWR, 0600);
      if (s.fd >= 0) {
        struct stat st;
        if (fstat(s.fd, &st) == 0 && static_cast<size_t>(st.st_size) >= s.bytes) break;
        close(s.fd);
        s.fd = -1;
      }
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60))
        return fail(ADMM_E_COMM, "timed out waiting for rank 0 to create " + s.name);
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  }
  s.base = mmap(nullptr, s.bytes, PROT_READ | PROT_WRITE, MAP_SHARED, s.fd, 0);
  if (s.base == MAP_FAILED) {
    s.base = nullptr;
    return fail(ADMM_E_COMM, "mmap failed on " + s.name);
  }
  // a fresh segment is zero-filled: header counters start at 0
  ShmHeader* h = static_cast<ShmHeader*>(s.base);
  h->attached.fetch_add(1, std::memory_order_acq_rel);
  ADMM_TRY(shm_barrier(c));
  if (c->rank == 0) shm_unlink(s.name.c_str());  // every rank holds a mapping; the name can go
  s.tmp.resize(kShmSlotElems);
  return ADMM_OK;
}

int shm_allreduce(admm_comm* c, double* dbuf, size_t count, hipStream_t stream) {
  ShmState& s = c->shm;
  double* slots = reinterpret_cast<double*>(static_cast<char*>(s.base) + sizeof(ShmHeader));
  for (size_t off = 0; off < count; off += kShmSlotElems) {
    const size_t k = (count - off < kShmSlotElems) ? count - off : kShmSlotElems;
    double* mine = slots + static_cast<size_t>(c->rank) * kShmSlotElems;
    ADMM_HIP_TRY(hipMemcpyAsync(mine, dbuf + off, k * sizeof(double), hipMemcpyDeviceToHost, stream));
    ADMM_HIP_TRY(hipStreamSynchronize(stream));
    ADMM_TRY(shm_barrier(c));
    for (size_t i = 0; i < k; ++i) {
      double acc = 0.0;
      for (int r = 0; r < c->nranks; ++r) acc += slots[static_cast<size_t>(r) * kShmSlotElems + i];  // rank order
      s.tmp[i] = acc;
    }
    ADMM_TRY(shm_barrier(c));  // all ranks finished reading before any slot is overwritten
    ADMM_HIP_TRY(hipMemcpyAsync(dbuf + off, s.tmp.data(), k * sizeof(double), hipMemcpyHostToDevice, stream));
    ADMM_HIP_TRY(hipStreamSynchronize(stream));
  }
  return ADMM_OK;
}

}  // namespace

namespace admm {

// A rank failed where its peers cannot see it (a bad descriptor, an out-of-memory, a rank-local Cholesky breakdown) while
// they sit in, or are about to enter, a collective that now never completes.  SHM: the group's shared header says so and
// every barrier returns ADMM_E_COMM.  RCCL: ncclCommAbort (callable from another thread) tears the communicator down and
// lets the kernel a peer is blocked in finish.  The group is unusable afterwards.
void comm_abort(admm_comm* c) {
  if (!c) return;
  if (c->transport == ADMM_COMM_SHM) {
    if (c->shm.base) static_cast<ShmHeader*>(c->shm.base)->aborted.store(1, std::memory_order_release);
    return;
  }
  if (c->comm && rccl().CommAbort) {
    (void)rccl().CommAbort(c->comm);
    c->comm = nullptr;
  }
}

int comm_nranks(admm_comm* c) { return c ? c->nranks : 1; }
int comm_rank(admm_comm* c) { return c ? c->rank : 0; }
bool comm_is_async(admm_comm* c) { return !c || c->nranks == 1 || c->transport == ADMM_COMM_RCCL; }

// in-place sum all-reduce of `count` doubles.  RCCL: enqueued on `stream`, returns immediately.
// SHM: blocks the host until the reduced values are back on the device.
int comm_allreduce_device(admm_comm* c, double* buf, size_t count, hipStream_t stream) {
  if (!c || c->nranks == 1 || count == 0) return ADMM_OK;
  if (c->transport == ADMM_COMM_SHM) return shm_allreduce(c, buf, count, stream);
  if (!c->comm) return fail(ADMM_E_COMM, "the communicator was aborted after a peer rank's failure");
  ncclResult_t r = rccl().AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, c->comm, stream);
  if (r != 0) return rccl_fail("ncclAllReduce", r);
  return ADMM_OK;
}

}  // namespace admm

extern "C" {

int admm_comm_unique_id(char id[ADMM_COMM_ID_BYTES]) {
  if (!id) return fail(ADMM_E_INVALID, "id is NULL");
  static_assert(sizeof(ncclUniqueId) == ADMM_COMM_ID_BYTES, "unique id size");
  RcclApi& a = rccl();
  int devs = 0;
  if (a.error.empty() && hipGetDeviceCount(&devs) == hipSuccess && devs > 0) {
    ncclUniqueId uid;
    ncclResult_t r = a.GetUniqueId(&uid);
    if (r == 0) {
      std::memcpy(id, uid.internal, ADMM_COMM_ID_BYTES);
      return ADMM_OK;
    }
  }
  // no usable RCCL: a random id is still good for the SHM transport
  FILE* f = std::fopen("/dev/urandom", "rb");
  if (!f || std::fread(id, 1, ADMM_COMM_ID_BYTES, f) != ADMM_COMM_ID_BYTES) {
    if (f) std::fclose(f);
    return fail(ADMM_E_COMM, "cannot create a unique id (no RCCL and no /dev/urandom)");
  }
  std::fclose(f);
  return ADMM_OK;
}

int admm_comm_init(const char id[ADMM_COMM_ID_BYTES], int rank, int nranks, int device, int transport,
                   admm_comm** out) {
  if (!id || !out || nranks < 1 || rank < 0 || rank >= nranks) return fail(ADMM_E_INVALID, "comm_init: bad argument");
  if (transport != ADMM_COMM_RCCL && transport != ADMM_COMM_SHM) return fail(ADMM_E_INVALID, "comm_init: bad transport");
  *out = nullptr;
  admm_comm* c = new admm_comm();
  c->rank = rank;
  c->nranks = nranks;
  c->device = device;
  c->transport = transport;
  if (transport == ADMM_COMM_SHM) {
    int rc = shm_attach(c, id);
    if (rc != ADMM_OK) {
      admm_comm_destroy(c);
      return rc;
    }
    *out = c;
    return ADMM_OK;
  }
  RcclApi& a = rccl();
  if (!a.error.empty()) {
    delete c;
    return fail(ADMM_E_COMM, a.error);
  }
  hipError_t he = hipSetDevice(device);
  if (he != hipSuccess) {
    delete c;
    return fail(ADMM_E_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(he));
  }
  ncclUniqueId uid;
  std::memcpy(uid.internal, id, ADMM_COMM_ID_BYTES);
  ncclResult_t r = a.CommInitRank(&c->comm, nranks, uid, rank);
  if (r != 0) {
    delete c;
    return rccl_fail("ncclCommInitRank", r);
  }
  *out = c;
  return ADMM_OK;
}

int admm_comm_info(admm_comm* comm, int* rank, int* nranks, int* transport) {
  if (!comm) return fail(ADMM_E_INVALID, "comm is NULL");
  if (rank) *rank = comm->rank;
  if (nranks) *nranks = comm->nranks;
  if (transport) *transport = comm->transport;
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
  if (comm->shm.base) munmap(comm->shm.base, comm->shm.bytes);
  if (comm->shm.fd >= 0) close(comm->shm.fd);
  delete comm;
}

}  // extern "C"
