// multi.hip -- ONE host process driving several GPUs: the reference opens its worker pool from one MATLAB session
// (admm.m:347-356, unwrappedadmm.m:47) and a MEX gateway lives in that one process too.  Every rank's engine is
// the same object as in the one-process-per-GPU deployment; a rank's create / run blocks inside its collectives, so
// the *_all entry points give each rank its own host thread for the duration of the call and join them.
#include <string>
#include <thread>
#include <vector>

#include "common.h"

using namespace admm;

namespace {

// runs fn(r) for r in [0, n) on n threads; the first failure's code and message become the caller's
template <class F>
int on_all_ranks(int n, F fn) {
  std::vector<int> rc(static_cast<size_t>(n), ADMM_OK);
  std::vector<std::string> msg(static_cast<size_t>(n));
  std::vector<std::thread> th;
  th.reserve(static_cast<size_t>(n));
  for (int r = 0; r < n; ++r)
    th.emplace_back([&, r] {
      rc[r] = fn(r);
      if (rc[r] != ADMM_OK) msg[r] = admm_last_error();  // thread-local: copy it out before the thread ends
    });
  for (auto& t : th) t.join();
  for (int r = 0; r < n; ++r)
    if (rc[r] != ADMM_OK) return fail(rc[r], "rank " + std::to_string(r) + ": " + msg[r]);
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
  const int rc = on_all_ranks(nranks, [&](int r) { return admm_comm_init(id, r, nranks, devices[r], transport, &comms[r]); });
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
  const int rc = on_all_ranks(nranks, [&](int r) { return admm_engine_create(&descs[r], &engines[r]); });
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
  return on_all_ranks(nranks, [&](int r) {
    return admm_engine_run(engines[r], opts_per_rank ? &opts[r] : &opts[0], summaries ? &summaries[r] : nullptr);
  });
}

}  // extern "C"
