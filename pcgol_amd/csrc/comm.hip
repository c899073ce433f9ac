// comm.hip -- the one exchange of the sharded ICP path behind the C ABI: every GPU
// holds a replica of the base tree and one spatial tile of the target; per iteration the 10 (plane:
// 30) float64 partial sums are all-reduced (sum) over the ranks, then every rank runs the evaluate
// tail + pose update redundantly.  Nothing else on the path communicates.
//
// RCCL is bound at run time (dlopen), not linked: a host that already carries an RCCL (a PyTorch
// process: torch/lib/librccl.so) must not get a second copy with a second HIP runtime, and a host
// that never shards (the reference is a single process) needs none.  One process per GPU, the
// communicator is created from an ncclUniqueId the host distributes (any out-of-band channel).
// pcgx_comm_init_callback is the same exchange through a host function (tests on one GPU, hosts with
// their own transport): the sums make a round trip through host memory.
#include <dlfcn.h>
#include <string.h>

#include "pcgx_internal.h"

namespace {

typedef int (*fn_get_unique_id)(void *id);
typedef int (*fn_comm_init_rank)(void **comm, int nranks, pcgx_comm_id id, int rank);  // ncclUniqueId by value: 128 bytes
typedef int (*fn_all_reduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t st);
typedef int (*fn_comm_destroy)(void *comm);
typedef const char *(*fn_error_string)(int rc);

struct Rccl {
  void *lib = nullptr;
  fn_get_unique_id get_unique_id = nullptr;
  fn_comm_init_rank comm_init_rank = nullptr;
  fn_all_reduce all_reduce = nullptr;
  fn_comm_destroy comm_destroy = nullptr;
  fn_error_string error_string = nullptr;
  std::string err;
};

Rccl &rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    if (getenv("PCGX_RCCL_DISABLE")) {  // tests: a host without RCCL (every pcgx_comm_unique_id / _init fails)
      r.err = "librccl.so not found: disabled by PCGX_RCCL_DISABLE";
      return;
    }
    // a library the process already holds first (RTLD_NOLOAD), then the default search path
    const char *names[] = {"librccl.so", "librccl.so.1"};
    for (const char *n : names)
      if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    if (!r.lib)
      if (const char *e = getenv("PCGX_RCCL_LIB")) r.lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
    for (const char *n : names)
      if (!r.lib) r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) r.lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!r.lib) {
      r.err = std::string("librccl.so not found: ") + (dlerror() ? dlerror() : "");
      return;
    }
    r.get_unique_id = (fn_get_unique_id)dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (fn_comm_init_rank)dlsym(r.lib, "ncclCommInitRank");
    r.all_reduce = (fn_all_reduce)dlsym(r.lib, "ncclAllReduce");
    r.comm_destroy = (fn_comm_destroy)dlsym(r.lib, "ncclCommDestroy");
    r.error_string = (fn_error_string)dlsym(r.lib, "ncclGetErrorString");
    if (!r.get_unique_id || !r.comm_init_rank || !r.all_reduce || !r.comm_destroy) r.err = "librccl.so lacks the nccl* entry points";
  });
  return r;
}

pcgx_status rccl_fail(const char *what, int rc) {
  Rccl &r = rccl();
  return pcgx::fail(PCGX_E_RCCL, "%s failed: %s", what, r.error_string ? r.error_string(rc) : "unknown RCCL error");
}

bool comm_force_collective() {
  const char *e = getenv("PCGX_COMM_FORCE_COLLECTIVE");
  return e && atoi(e) != 0;
}

constexpr int kNcclFloat64 = 8, kNcclSum = 0;  // rccl.h: ncclDataType_t, ncclRedOp_t

}  // namespace

struct pcgx_comm {
  int32_t rank = 0, world = 1;
  void *nccl = nullptr;
  pcgx_allreduce_fn fn = nullptr;
  void *user = nullptr;
};

using namespace pcgx;

extern "C" pcgx_status pcgx_comm_unique_id(pcgx_comm_id *id) {
  if (!id) return fail(PCGX_E_INVALID, "pcgx_comm_unique_id: NULL argument");
  Rccl &r = rccl();
  if (!r.err.empty()) return fail(PCGX_E_RCCL, "%s", r.err.c_str());
  PCGX_TRY(ensure_init());
  const int rc = r.get_unique_id(id);
  if (rc != 0) return rccl_fail("ncclGetUniqueId", rc);
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_comm_init(int32_t rank, int32_t world, const pcgx_comm_id *id, pcgx_comm **out) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world) return fail(PCGX_E_INVALID, "pcgx_comm_init: bad argument");
  *out = nullptr;
  Rccl &r = rccl();
  if (!r.err.empty()) return fail(PCGX_E_RCCL, "%s", r.err.c_str());
  PCGX_TRY(ensure_init());
  PCGX_HIP_TRY(hipSetDevice(ctx().device));
  pcgx_comm *c = new pcgx_comm();
  c->rank = rank;
  c->world = world;
  const int rc = r.comm_init_rank(&c->nccl, world, *id, rank);
  if (rc != 0) {
    delete c;
    return rccl_fail("ncclCommInitRank", rc);
  }
  *out = c;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_comm_init_callback(int32_t rank, int32_t world, pcgx_allreduce_fn fn, void *user,
                                               pcgx_comm **out) {
  if (!out || !fn || world < 1 || rank < 0 || rank >= world) return fail(PCGX_E_INVALID, "pcgx_comm_init_callback: bad argument");
  pcgx_comm *c = new pcgx_comm();
  c->rank = rank;
  c->world = world;
  c->fn = fn;
  c->user = user;
  *out = c;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_comm_free(pcgx_comm *c) {
  if (!c) return PCGX_OK;
  if (c->nccl) (void)rccl().comm_destroy(c->nccl);
  delete c;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_comm_rank(const pcgx_comm *c, int32_t *rank, int32_t *world) {
  if (!c || !rank || !world) return fail(PCGX_E_INVALID, "pcgx_comm_rank: NULL argument");
  *rank = c->rank;
  *world = c->world;
  return PCGX_OK;
}

// the same for a few doubles in HOST memory (a rank that could not get device memory still has to answer the others)
extern "C" pcgx_status pcgx_comm_allreduce_host_f64(pcgx_comm *c, double *h_buf, int32_t count) {
  if (!c || !h_buf || count < 1) return fail(PCGX_E_INVALID, "pcgx_comm_allreduce_host_f64: bad argument");
  if (c->world == 1 && !comm_force_collective()) return PCGX_OK;
  if (!c->nccl) {
    const int32_t rc = c->fn(h_buf, count, c->user);
    if (rc != 0) return fail(PCGX_E_RCCL, "the host's all-reduce callback failed (%d)", rc);
    return PCGX_OK;
  }
  PCGX_TRY(ensure_init());
  double *d = nullptr;
  PCGX_HIP_TRY(hipMalloc((void **)&d, (size_t)count * sizeof(double)));
  hipStream_t st = ctx().stream;
  pcgx_status rc = PCGX_OK;
  if (hipMemcpyAsync(d, h_buf, (size_t)count * sizeof(double), hipMemcpyHostToDevice, st) != hipSuccess) rc = PCGX_E_HIP;
  if (rc == PCGX_OK) rc = pcgx_comm_allreduce_f64(c, d, count, st);
  if (rc == PCGX_OK && (hipMemcpyAsync(h_buf, d, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, st) != hipSuccess ||
                        hipStreamSynchronize(st) != hipSuccess))
    rc = PCGX_E_HIP;
  (void)hipFree(d);
  return rc;
}

// sum over the ranks of `count` float64 in device memory, in place, in stream order
extern "C" pcgx_status pcgx_comm_allreduce_f64(pcgx_comm *c, double *d_buf, int32_t count, void *stream) {
  if (!c || !d_buf || count < 1) return fail(PCGX_E_INVALID, "pcgx_comm_allreduce_f64: bad argument");
  // (one rank: the sum is the buffer itself.  PCGX_COMM_FORCE_COLLECTIVE=1 runs the collective all the same -- tests
  // on a one-GPU box: ncclAllReduce, or the callback's round trip through host memory, executes at least once there)
  if (c->world == 1 && !comm_force_collective()) return PCGX_OK;
  PCGX_TRY(ensure_init());
  hipStream_t st = pick_stream(stream);
  if (c->nccl) {
    const int rc = rccl().all_reduce(d_buf, d_buf, (size_t)count, kNcclFloat64, kNcclSum, c->nccl, st);
    if (rc != 0) return rccl_fail("ncclAllReduce", rc);
    return PCGX_OK;
  }
  std::vector<double> h((size_t)count);
  PCGX_HIP_TRY(hipMemcpyAsync(h.data(), d_buf, (size_t)count * sizeof(double), hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  const int32_t rc = c->fn(h.data(), count, c->user);
  if (rc != 0) return fail(PCGX_E_RCCL, "the host's all-reduce callback failed (%d)", rc);
  PCGX_HIP_TRY(hipMemcpyAsync(d_buf, h.data(), (size_t)count * sizeof(double), hipMemcpyHostToDevice, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  return PCGX_OK;
}
