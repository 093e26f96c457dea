// multi.hip -- ONE host process driving several GPUs: the reference opens its worker pool from one MATLAB session
// (admm.m:347-356, unwrappedadmm.m:47) and a MEX gateway lives in that one process too.  Every rank's engine is
// the same object as in the one-process-per-GPU deployment; a rank's create / run blocks inside its collectives, so
// the *_all entry points give each rank its own host thread for the duration of the call and join them.
#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "common.h"

using namespace admm;

struct admm_comm;
struct admm_engine;
namespace admm {
void comm_abort(admm_comm* c);
admm_comm* engine_comm(admm_engine* e);  // engine_run.hip
}  // namespace admm

namespace {

// Runs fn(r) for r in [0, n) on n threads; the first failure's code and message become the caller's.  A rank that fails
// outside a collective (bad descriptor, out of memory, a rank-local factorisation breaking down) would leave its peers
// blocked in the next one -- for 120 s on the shm transport, for ever inside RCCL, with the one host process (a MATLAB
// session, a Python caller) holding every GPU.  So the failing rank's thread abandons the whole group at once
// (comm_abort on every communicator: the shm header's flag, ncclCommAbort) and the join below returns promptly; the
// group is unusable afterwards and the caller gets the FIRST failure, not the ADMM_E_COMM of the peers it released.
template <class F>
int on_all_ranks(int n, admm_comm* const* comms, F fn) {
  std::vector<int> rc(static_cast<size_t>(n), ADMM_OK);
  std::vector<std::string> msg(static_cast<size_t>(n));
  std::vector<std::thread> th;
  std::atomic<int> first{-1};
  th.reserve(static_cast<size_t>(n));
  for (int r = 0; r < n; ++r)
    th.emplace_back([&, r] {
      rc[r] = fn(r);
      if (rc[r] != ADMM_OK) {
        msg[r] = admm_last_error();  // thread-local: copy it out before the thread ends
        int none = -1;
        if (first.compare_exchange_strong(none, r) && comms)
          for (int q = 0; q < n; ++q) comm_abort(comms[q]);
      }
    });
  for (auto& t : th) t.join();
  const int f = first.load();
  if (f >= 0) return fail(rc[f], "rank " + std::to_string(f) + ": " + msg[f]);
  return ADMM_OK;
}

}  // namespace

extern "C" {

int admm_comm_init_all(int nranks, const int* devices, int transport, admm_comm** comms) {
  if (nranks < 1 || !devices || !comms) return fail(ADMM_E_INVALID, "comm_init_all: bad argument");
  char id[ADMM_COMM_ID_BYTES];
  ADMM_TRY(admm_comm_unique_id(id));
  for (int r = 0; r < nranks; ++r) comms[r] = nullptr;
  // every rank joins the rendezvous from its own thread (ncclCommInitRank and the shm attach both block until all
  // ranks have arrived)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
  for (int r = 0; r < nranks; ++r)  // (checked here, on the calling thread: a rank that fails alone strands the others in the rendezvous)
    if (devices[r] < 0 || devices[r] >= ndev)
      return fail(ADMM_E_INVALID, "comm_init_all: rank " + std::to_string(r) + " names device " + std::to_string(devices[r]) +
                                      " of " + std::to_string(ndev));
  const int rc = on_all_ranks(nranks, nullptr, [&](int r) { return admm_comm_init(id, r, nranks, devices[r], transport, &comms[r]); });
  if (rc != ADMM_OK)
    for (int r = 0; r < nranks; ++r) {
      admm_comm_destroy(comms[r]);
      comms[r] = nullptr;
    }
  return rc;
}

int admm_engine_create_all(int nranks, const admm_problem_desc* descs, admm_engine** engines) {
  if (nranks < 1 || !descs || !engines) return fail(ADMM_E_INVALID, "create_all: bad argument");
  for (int r = 0; r < nranks; ++r) engines[r] = nullptr;
  std::vector<admm_comm*> comms(static_cast<size_t>(nranks));
  for (int r = 0; r < nranks; ++r) {  // what can be refused without touching a device is refused before any rank starts
    comms[r] = descs[r].comm;
    if ((descs[r].comm == nullptr) != (descs[0].comm == nullptr))
      return fail(ADMM_E_INVALID, "create_all: either every rank has a communicator or none");
    if (descs[r].problem != descs[0].problem)
      return fail(ADMM_E_INVALID, "create_all: rank " + std::to_string(r) + " describes another problem than rank 0");
    if (descs[r].m < 0 || descs[r].n < 0 || (descs[r].D && descs[r].m == 0))
      return fail(ADMM_E_INVALID, "create_all: rank " + std::to_string(r) + " has no rows");
  }
  const int rc = on_all_ranks(nranks, comms.data(), [&](int r) { return admm_engine_create(&descs[r], &engines[r]); });
  if (rc != ADMM_OK)
    for (int r = 0; r < nranks; ++r) {
      admm_engine_destroy(engines[r]);
      engines[r] = nullptr;
    }
  return rc;
}

int admm_engine_run_all(int nranks, admm_engine* const* engines, const admm_options* opts, int opts_per_rank,
                        admm_run_summary* summaries) {
  if (nranks < 1 || !engines || !opts) return fail(ADMM_E_INVALID, "run_all: bad argument");
  std::vector<admm_comm*> comms(static_cast<size_t>(nranks));
  for (int r = 0; r < nranks; ++r) {
    if (!engines[r]) return fail(ADMM_E_INVALID, "run_all: engine of rank " + std::to_string(r) + " is NULL");
    comms[r] = engine_comm(engines[r]);
  }
  return on_all_ranks(nranks, comms.data(), [&](int r) {
    return admm_engine_run(engines[r], opts_per_rank ? &opts[r] : &opts[0], summaries ? &summaries[r] : nullptr);
  });
}

}  // extern "C"
