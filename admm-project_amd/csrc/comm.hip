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

// ---- P2P transport: per-rank device buffers, addresses exchanged through the shm segment ------------------------------
constexpr size_t kP2pCap = size_t{1} << 15;  // doubles per peer slot (256 KB); larger payloads go in rounds
constexpr int kP2pMaxRanks = 16;
struct P2pExchange {  // one per rank, in the shm segment behind the all-reduce slots
  int64_t pid;
  int64_t device;
  uint64_t raw_recv, raw_flags;  // device addresses (valid inside that process)
  hipIpcMemHandle_t h_recv, h_flags;
};
struct P2pState {
  double* recv = nullptr;               // [2][nranks][kP2pCap] on this rank's device: parity of the round, sender, element
  unsigned long long* flags = nullptr;  // [nranks]: round number of the last complete contribution of every sender
  int32_t* state = nullptr;             // [0] workgroup arrival counter, [1] error word
  double** peer_recv = nullptr;         // device array [nranks]: every rank's recv (this process's mapping)
  unsigned long long** peer_flags = nullptr;
  std::vector<void*> opened;            // IPC mappings to close
  unsigned long long round = 0;
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
  P2pState p2p;
  int rank = 0, nranks = 1, device = 0;
  int queue_slot = 0, queue_sharers = 1;  // P2P, ranks that are threads of one process on ONE device (comm_stream_create)
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
  s.bytes = sizeof(ShmHeader) + static_cast<size_t>(c->nranks) * kShmSlotElems * sizeof(double) +
            sizeof(P2pExchange) * kP2pMaxRanks;  // (the P2P transport's address exchange sits behind the slots)
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

// ---- the one-shot all-reduce ------------------------------------------------------------------------------------------
// buf[i] <- sum over ranks of buf[i], i < count <= kP2pCap.  Every rank runs this kernel (same count, same round) on its
// own stream; the kernels must be able to run at the same time (one GPU per rank; ranks sharing a GPU need a hardware
// queue each).  System-scope stores / loads for everything another device reads or writes; the slot set alternates with
// the round's parity: a rank can be one round ahead of a peer (it needs that peer's flag to finish a round), never two.
constexpr int kP2pSpinMax = 1 << 22;  // ~5 s of polling: a missing peer raises the error word instead of hanging

__global__ __launch_bounds__(kBlock) void p2p_allreduce_kernel(double* __restrict__ buf, uint32_t count, int rank,
                                                               int nranks, double* const* __restrict__ peer_recv,
                                                               unsigned long long* const* __restrict__ peer_flags,
                                                               double* __restrict__ my_recv,
                                                               unsigned long long* __restrict__ my_flags,
                                                               unsigned long long round, int32_t* __restrict__ state) {
  __shared__ int32_t sh_last;
  const uint32_t stride = gridDim.x * kBlock, t0 = blockIdx.x * kBlock + threadIdx.x;
  const size_t set = static_cast<size_t>(round & 1ull) * static_cast<size_t>(nranks) * kP2pCap;
  // 1. push this rank's contribution into its slot on every rank (its own included)
  for (uint32_t i = t0; i < count; i += stride) {
    const double v = buf[i];
    for (int p = 0; p < nranks; ++p)
      __hip_atomic_store(peer_recv[p] + set + static_cast<size_t>(rank) * kP2pCap + i, v, __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_SYSTEM);
  }
  // 2. every storing wave drains its stores; the last workgroup of this rank raises the flag on every rank
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");  // system scope
    const int32_t old = __hip_atomic_fetch_add(state, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    sh_last = (old == static_cast<int32_t>(gridDim.x) - 1) ? 1 : 0;
    if (sh_last) {
      __hip_atomic_store(state, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int p = 0; p < nranks; ++p)
        __hip_atomic_store(peer_flags[p] + rank, round, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // 3. wait until every rank's contribution of this round has landed here
  if (threadIdx.x == 0) {
    for (int q = 0; q < nranks; ++q) {
      int spin = 0;
      while (__hip_atomic_load(my_flags + q, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < round) {
        if (++spin > kP2pSpinMax || __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
          __hip_atomic_store(state + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(32);
      }
    }
  }
  __syncthreads();
  const bool failed = __hip_atomic_load(state + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
  // 4. rank-ordered sum: bitwise the same on every rank
  for (uint32_t i = t0; i < count; i += stride) {
    double acc = 0.0;
    for (int q = 0; q < nranks; ++q)
      acc += __hip_atomic_load(my_recv + set + static_cast<size_t>(q) * kP2pCap + i, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_SYSTEM);
    buf[i] = failed ? __builtin_nan("") : acc;
  }
}

int p2p_allreduce(admm_comm* c, double* dbuf, size_t count, hipStream_t stream) {
  P2pState& s = c->p2p;
  for (size_t off = 0; off < count; off += kP2pCap) {
    const uint32_t k = static_cast<uint32_t>((count - off < kP2pCap) ? count - off : kP2pCap);
    s.round += 1;
    unsigned blocks = (k + kBlock - 1) / kBlock;
    if (blocks > 32) blocks = 32;
    hipLaunchKernelGGL(p2p_allreduce_kernel, dim3(blocks), dim3(kBlock), 0, stream, dbuf + off, k, c->rank, c->nranks,
                       s.peer_recv, s.peer_flags, s.recv, s.flags, s.round, s.state);
  }
  return ADMM_OK;
}

void p2p_release(admm_comm* c) {
  P2pState& s = c->p2p;
  for (void* m : s.opened) (void)hipIpcCloseMemHandle(m);
  s.opened.clear();
  for (void* d : {static_cast<void*>(s.recv), static_cast<void*>(s.flags), static_cast<void*>(s.state),
                  static_cast<void*>(s.peer_recv), static_cast<void*>(s.peer_flags)})
    if (d) (void)hipFree(d);
  s = P2pState{};
}

// after shm_attach: allocate, publish addresses / IPC handles, map every peer's buffers
int p2p_attach(admm_comm* c) {
  if (c->nranks > kP2pMaxRanks) return fail(ADMM_E_UNSUPPORTED, "the P2P transport serves up to 16 ranks of one node");
  P2pState& s = c->p2p;
  ADMM_HIP_TRY(hipSetDevice(c->device));
  const size_t recv_bytes = sizeof(double) * 2 * static_cast<size_t>(c->nranks) * kP2pCap;
  // fine-grained device memory where the runtime offers it (remote stores visible without a kernel boundary)
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&s.recv), recv_bytes, hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.recv), recv_bytes));
  }
  if (hipExtMallocWithFlags(reinterpret_cast<void**>(&s.flags), sizeof(unsigned long long) * kP2pMaxRanks,
                            hipDeviceMallocFinegrained) != hipSuccess) {
    (void)hipGetLastError();
    ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.flags), sizeof(unsigned long long) * kP2pMaxRanks));
  }
  ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&s.state), sizeof(int32_t) * 4));
  ADMM_HIP_TRY(hipMemset(s.recv, 0, recv_bytes));
  ADMM_HIP_TRY(hipMemset(s.flags, 0, sizeof(unsigned long long) * kP2pMaxRanks));
  ADMM_HIP_TRY(hipMemset(s.state, 0, sizeof(int32_t) * 4));
  ADMM_HIP_TRY(hipDeviceSynchronize());
  P2pExchange* ex = reinterpret_cast<P2pExchange*>(static_cast<char*>(c->shm.base) + sizeof(ShmHeader) +
                                                   static_cast<size_t>(c->nranks) * kShmSlotElems * sizeof(double));
  P2pExchange mine{};
  mine.pid = static_cast<int64_t>(getpid());
  mine.device = c->device;
  mine.raw_recv = reinterpret_cast<uint64_t>(s.recv);
  mine.raw_flags = reinterpret_cast<uint64_t>(s.flags);
  if (hipIpcGetMemHandle(&mine.h_recv, s.recv) != hipSuccess || hipIpcGetMemHandle(&mine.h_flags, s.flags) != hipSuccess) {
    (void)hipGetLastError();  // same-process peers do not need the handles; other processes will fail to open them
    std::memset(&mine.h_recv, 0, sizeof(mine.h_recv));
    std::memset(&mine.h_flags, 0, sizeof(mine.h_flags));
  }
  ex[c->rank] = mine;
  ADMM_TRY(shm_barrier(c));
  std::vector<double*> pr(static_cast<size_t>(c->nranks));
  std::vector<unsigned long long*> pf(static_cast<size_t>(c->nranks));
  int rc = ADMM_OK;
  c->queue_slot = 0;
  c->queue_sharers = 0;
  for (int p = 0; p < c->nranks; ++p)
    if (ex[p].pid == mine.pid && ex[p].device == mine.device) {
      c->queue_sharers += 1;
      if (p < c->rank) c->queue_slot += 1;
    }
  for (int p = 0; p < c->nranks && rc == ADMM_OK; ++p) {
    const P2pExchange& e = ex[p];
    if (e.pid == mine.pid) {  // a thread of this process: its addresses are ours
      pr[p] = reinterpret_cast<double*>(e.raw_recv);
      pf[p] = reinterpret_cast<unsigned long long*>(e.raw_flags);
      continue;
    }
    void *mr = nullptr, *mf = nullptr;
    if (hipIpcOpenMemHandle(&mr, e.h_recv, hipIpcMemLazyEnablePeerAccess) != hipSuccess ||
        hipIpcOpenMemHandle(&mf, e.h_flags, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
      rc = fail(ADMM_E_COMM, std::string("P2P transport: hipIpcOpenMemHandle failed for rank ") + std::to_string(p) +
                                 " (" + hipGetErrorString(hipGetLastError()) + ")");
      break;
    }
    s.opened.push_back(mr);
    s.opened.push_back(mf);
    pr[p] = static_cast<double*>(mr);
    pf[p] = static_cast<unsigned long long*>(mf);
  }
  if (rc == ADMM_OK) {
    int ndev = 0;
    (void)hipGetDeviceCount(&ndev);
    for (int d = 0; d < ndev; ++d)  // peers of this process on other devices (one process driving N GPUs)
      if (d != c->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, c->device, d) == hipSuccess && can) {
          if (hipDeviceEnablePeerAccess(d, 0) != hipSuccess) (void)hipGetLastError();  // (already enabled: fine)
        }
      }
    if (hipMalloc(reinterpret_cast<void**>(&s.peer_recv), sizeof(double*) * kP2pMaxRanks) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&s.peer_flags), sizeof(unsigned long long*) * kP2pMaxRanks) != hipSuccess ||
        hipMemcpy(s.peer_recv, pr.data(), sizeof(double*) * c->nranks, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(s.peer_flags, pf.data(), sizeof(unsigned long long*) * c->nranks, hipMemcpyHostToDevice) != hipSuccess)
      rc = fail(ADMM_E_DEVICE, "P2P transport: device pointer tables");
  }
  const int rb = shm_barrier(c);  // nobody proceeds (or tears down) before every rank has mapped every peer
  return rc != ADMM_OK ? rc : rb;
}

}  // namespace

namespace admm {

// the P2P transport's error word (a peer never arrived): synchronises the stream
int comm_check_error(admm_comm* c, hipStream_t stream) {
  if (!c || c->transport != ADMM_COMM_P2P || !c->p2p.state) return ADMM_OK;
  int32_t err = 0;
  ADMM_HIP_TRY(hipMemcpyAsync(&err, c->p2p.state + 1, sizeof(int32_t), hipMemcpyDeviceToHost, stream));
  ADMM_HIP_TRY(hipStreamSynchronize(stream));
  if (err != 0)
    return fail(ADMM_E_COMM, "P2P all-reduce: a peer rank's contribution did not arrive within the polling limit "
                             "(the ranks' kernels must be able to run at the same time)");
  return ADMM_OK;
}

// A rank failed where its peers cannot see it (a bad descriptor, an out-of-memory, a rank-local Cholesky breakdown) while
// they sit in, or are about to enter, a collective that now never completes.  SHM: the group's shared header says so and
// every barrier returns ADMM_E_COMM.  RCCL: ncclCommAbort (callable from another thread) tears the communicator down and
// lets the kernel a peer is blocked in finish.  The group is unusable afterwards.
void comm_abort(admm_comm* c) {
  if (!c) return;
  if (c->transport == ADMM_COMM_SHM || c->transport == ADMM_COMM_P2P) {
    if (c->shm.base) static_cast<ShmHeader*>(c->shm.base)->aborted.store(1, std::memory_order_release);
    // (P2P: a peer's kernel that is already polling gives up at its own limit and poisons its result)
    return;
  }
  if (c->comm && rccl().CommAbort) {
    (void)rccl().CommAbort(c->comm);
    c->comm = nullptr;
  }
}

int comm_nranks(admm_comm* c) { return c ? c->nranks : 1; }
int comm_rank(admm_comm* c) { return c ? c->rank : 0; }
bool comm_is_async(admm_comm* c) { return !c || c->nranks == 1 || c->transport != ADMM_COMM_SHM; }

// in-place sum all-reduce of `count` doubles.  RCCL: enqueued on `stream`, returns immediately.
// SHM: blocks the host until the reduced values are back on the device.
int comm_allreduce_device(admm_comm* c, double* buf, size_t count, hipStream_t stream) {
  if (!c || c->nranks == 1 || count == 0) return ADMM_OK;
  if (c->transport == ADMM_COMM_SHM) return shm_allreduce(c, buf, count, stream);
  if (c->transport == ADMM_COMM_P2P) return p2p_allreduce(c, buf, count, stream);
  if (!c->comm) return fail(ADMM_E_COMM, "the communicator was aborted after a peer rank's failure");
  ncclResult_t r = rccl().AllReduce(buf, buf, count, kNcclFloat64, kNcclSum, c->comm, stream);
  if (r != 0) return rccl_fail("ncclAllReduce", r);
  return ADMM_OK;
}

// A stream for work that contains this communicator's collectives.  The P2P kernel of a rank waits for the kernels of its
// peers, so ranks that are threads of ONE process on ONE device (the one-GPU rehearsal of the MEX-gateway deployment)
// need hardware queues of their own: the runtime deals a process's streams onto a small pool of queues PER PRIORITY
// (which queue a new stream gets depends on every stream the process made before: two ranks that landed on one queue
// trip the polling limit), so such ranks take different priorities -- a different pool each.  With more sharers than
// priority levels (three here) the pools are shared again.  One rank per device: a plain stream.
int comm_stream_create(const admm_comm* comm, hipStream_t* out) {
  if (comm && comm->transport == ADMM_COMM_P2P && comm->queue_sharers > 1) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && least > greatest) {
      const int levels = least - greatest + 1;
      if (hipStreamCreateWithPriority(out, hipStreamNonBlocking, greatest + comm->queue_slot % levels) == hipSuccess)
        return ADMM_OK;
    }
    (void)hipGetLastError();
  }
  return hipStreamCreateWithFlags(out, hipStreamNonBlocking) == hipSuccess ? ADMM_OK : fail(ADMM_E_DEVICE, "hipStreamCreate");
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
  if (transport != ADMM_COMM_RCCL && transport != ADMM_COMM_SHM && transport != ADMM_COMM_P2P)
    return fail(ADMM_E_INVALID, "comm_init: bad transport");
  *out = nullptr;
  admm_comm* c = new admm_comm();
  c->rank = rank;
  c->nranks = nranks;
  c->device = device;
  c->transport = transport;
  if (transport == ADMM_COMM_SHM || transport == ADMM_COMM_P2P) {
    int rc = shm_attach(c, id);
    if (rc == ADMM_OK && transport == ADMM_COMM_P2P) rc = p2p_attach(c);
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
  // a stream of its own: ranks that are threads of one process must not queue behind each other on the null stream
  // (the P2P kernel of one rank waits for the kernels of the others)
  hipStream_t st = nullptr;
  if (comm_stream_create(comm, &st) != ADMM_OK) {
    (void)hipFree(d);
    return ADMM_E_DEVICE;
  }
  if (hipMemcpy(d, host_buf, sizeof(double) * count, hipMemcpyHostToDevice) != hipSuccess)
    rc = fail(ADMM_E_DEVICE, "hipMemcpy H2D");
  if (rc == ADMM_OK) rc = comm_allreduce_device(comm, d, count, st);
  if (rc == ADMM_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(ADMM_E_DEVICE, "hipStreamSynchronize");
  if (rc == ADMM_OK) rc = comm_check_error(comm, st);
  (void)hipStreamDestroy(st);
  if (rc == ADMM_OK && hipMemcpy(host_buf, d, sizeof(double) * count, hipMemcpyDeviceToHost) != hipSuccess)
    rc = fail(ADMM_E_DEVICE, "hipMemcpy D2H");
  (void)hipFree(d);
  return rc;
}

int admm_comm_measure_latency(admm_comm* comm, size_t count, int reps, double* microseconds) {
  if (!comm || !microseconds || reps < 1 || count == 0) return fail(ADMM_E_INVALID, "measure_latency: bad argument");
  ADMM_HIP_TRY(hipSetDevice(comm->device));
  double* d = nullptr;
  ADMM_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&d), sizeof(double) * count));
  hipStream_t st = nullptr;
  if (comm_stream_create(comm, &st) != ADMM_OK) {
    (void)hipFree(d);
    return ADMM_E_DEVICE;
  }
  int rc = ADMM_OK;
  if (hipMemsetAsync(d, 0, sizeof(double) * count, st) != hipSuccess) rc = fail(ADMM_E_DEVICE, "hipMemsetAsync");
  for (int k = 0; k < 3 && rc == ADMM_OK; ++k) rc = comm_allreduce_device(comm, d, count, st);
  if (rc == ADMM_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(ADMM_E_DEVICE, "hipStreamSynchronize");
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < reps && rc == ADMM_OK; ++k) rc = comm_allreduce_device(comm, d, count, st);
  if (rc == ADMM_OK && hipStreamSynchronize(st) != hipSuccess) rc = fail(ADMM_E_DEVICE, "hipStreamSynchronize");
  *microseconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / reps;
  if (rc == ADMM_OK) rc = comm_check_error(comm, st);
  (void)hipStreamDestroy(st);
  (void)hipFree(d);
  return rc;
}

void admm_comm_destroy(admm_comm* comm) {
  if (!comm) return;
  if (comm->comm) (void)rccl().CommDestroy(comm->comm);
  if (comm->transport == ADMM_COMM_P2P) p2p_release(comm);
  if (comm->shm.base) munmap(comm->shm.base, comm->shm.bytes);
  if (comm->shm.fd >= 0) close(comm->shm.fd);
  delete comm;
}

}  // extern "C"
