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
#include <fcntl.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include "pcgx_internal.h"
#include "strict_terms.h"

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

std::atomic<long long> g_shard_stats[4];  // pcgx_debug_shard_stats
std::atomic<long long> g_ring_kinds[2];   // pcgx_debug_ring_kinds

}  // namespace

struct pcgx_comm {
  int32_t rank = 0, world = 1;
  void *nccl = nullptr;
  pcgx_allreduce_fn fn = nullptr;
  void *user = nullptr;
  // The ring of the reference-sums steps (strict.hip, strict_enqueue_ring): every rank's inbox in ONE block of
  // host-coherent memory that all GPUs of the node write and poll -- pinned host memory of the one process
  // (pcgx_icp_fit_multi hands it in), or a POSIX shared-memory segment that every process of the node maps and
  // registers with HIP (made here, collectively, on first use: the communicator's own all-reduce carries the segment's
  // name and whether every rank could map it).  Ranks that cannot share memory (several nodes) keep the collectives.
  bool ring_tried = false;
  unsigned long long *ring_host = nullptr, *ring_dev = nullptr;
  int32_t ring_words = 0;
  void *ring_map = nullptr;  // a mapping this communicator owns (shared memory); nullptr: somebody else's block
  size_t ring_bytes = 0;
  uint32_t ring_fit = 0;  // Fits begun on this communicator (comm_ring_new_fit; every rank counts alike): the upper bits of a word's tag
  // The inboxes' DATA words in DEVICE memory (ring_setup_device, collective, behind the host block): every rank's
  // inbox lives in its own GPU's memory and is mapped by the others -- hipIpcOpenMemHandle between processes, the
  // plain pointer (+ hipDeviceEnablePeerAccess) between the device slots of one process.  A store into a peer's inbox
  // crosses xGMI once; the owner polls its own HBM (0.55 us per hop against 2.2 through pinned host memory,
  // tools/micro/ipc_hop.cpp).  Where that cannot be had on EVERY rank the data words stay in the host block.
  bool dev_tried = false;
  unsigned long long *inbox = nullptr;               // mine
  std::vector<unsigned long long *> peers;            // [world] every rank's inbox as this process addresses it
  std::vector<void *> ipc_open;                       // mappings to close
  unsigned long long **ring_tab = nullptr;            // the same table in this rank's device memory (StrictWork::ring_tab)
  int ring_kind = 0;                                  // 0: no ring (collectives); 1: host-coherent memory; 2: device memory
  bool ranks_share_device = true;                     // two ranks on one physical GPU (or unknown): the short wait bounds
};

namespace {

void ring_unmap(pcgx_comm *c) {
  for (void *p : c->ipc_open) (void)hipIpcCloseMemHandle(p);
  c->ipc_open.clear();
  if (c->ring_tab) (void)hipFree(c->ring_tab);
  if (c->inbox) (void)hipFree(c->inbox);
  c->ring_tab = nullptr;
  c->inbox = nullptr;
  c->peers.clear();
  if (c->ring_map) {
    (void)hipHostUnregister(c->ring_map);
    (void)munmap(c->ring_map, c->ring_bytes);
  }
  c->ring_map = nullptr;
  c->ring_host = c->ring_dev = nullptr;
}

// collective: every rank of `c` calls it at the same point (the first sharded step / Fit with the reference's sums)
void ring_setup(pcgx_comm *c) {
  c->ring_tried = true;
  if (c->world < 2 || c->world > 64 || c->ring_dev) return;
  const pcgx::RingLayout RL{c->world};
  const size_t bytes = ((size_t)c->world * RL.words() * sizeof(unsigned long long) + 4095) & ~(size_t)4095;
  const char *off = getenv("PCGX_SHARD_RING");
  const bool disabled = off && atoi(off) == 0;
  static std::atomic<unsigned> counter{0};
  double v[3] = {0.0, 0.0, disabled ? 1.0 : 0.0};  // segment name: rank 0's pid and a number of its own; ranks that cannot
  char name[64];
  int fd = -1;
  if (c->rank == 0 && !disabled) {
    const unsigned nonce = (counter++ * 2654435761u + (unsigned)time(nullptr)) & 0x3fffffffu;
    snprintf(name, sizeof name, "/pcgx_ring_%d_%u", (int)getpid(), nonce);
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) v[2] += 1.0;  // (a fresh segment reads as zeros: epoch 0, nobody's)
    v[0] = (double)getpid();
    v[1] = (double)nonce;
  }
  bool ok = pcgx_comm_allreduce_host_f64(c, v, 3) == PCGX_OK && v[2] == 0.0;
  if (ok && c->rank != 0) {
    snprintf(name, sizeof name, "/pcgx_ring_%d_%u", (int)v[0], (unsigned)v[1]);
    fd = shm_open(name, O_RDWR, 0600);  // (a rank on another node finds no such segment)
    struct stat sb;
    if (fd < 0 || fstat(fd, &sb) != 0 || (size_t)sb.st_size != bytes) ok = false;
  }
  void *map = MAP_FAILED;
  void *dev = nullptr;
  bool mine = ok;
  if (mine) {
    map = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    mine = map != MAP_FAILED;
  }
  if (fd >= 0) close(fd);
  if (mine && pcgx::ensure_init() != PCGX_OK) mine = false;
  bool registered = false;
  if (mine) {
    registered = hipHostRegister(map, bytes, hipHostRegisterMapped | hipHostRegisterPortable) == hipSuccess;
    mine = registered && hipHostGetDevicePointer(&dev, map, 0) == hipSuccess && dev != nullptr;
    if (!mine) (void)hipGetLastError();
  }
  double bad = mine ? 0.0 : 1.0;
  const bool agreed = pcgx_comm_allreduce_host_f64(c, &bad, 1) == PCGX_OK && bad == 0.0;
  if (c->rank == 0 && !disabled && v[0] != 0.0) (void)shm_unlink(name);  // (every rank that could has it mapped by now)
  if (!agreed) {
    if (registered) (void)hipHostUnregister(map);
    if (map != MAP_FAILED) (void)munmap(map, bytes);
    g_shard_stats[3]++;
    return;
  }
  g_shard_stats[2]++;
  c->ring_map = map;
  c->ring_bytes = bytes;
  c->ring_host = (unsigned long long *)map;
  c->ring_dev = (unsigned long long *)dev;
  c->ring_words = RL.words();
}

// The mapped inboxes tried out before a Fit depends on them: every rank stores a tagged word into the test slot it owns
// in every peer's inbox (behind the layout's words) and waits -- bounded -- for every peer's word in its own.  What it
// proves is what the ring needs: a store through the mapping is seen by the owner's poll, in both directions, between
// these very devices.  (No box of this pipeline has two GPUs: the first node that does must not find out in the walkers'
// ten-second wait.)
__global__ __launch_bounds__(64) void ring_selftest_kernel(unsigned long long *const *tab, unsigned long long *mine, int world, int rank,
                                                          int off, unsigned long long tag, long long max_ticks, int *ok) {
  const int k = (int)threadIdx.x;
  if (k < world && k != rank)
    __hip_atomic_store(tab[k] + off + rank, tag | (unsigned long long)rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  bool seen = !(k < world && k != rank);
  long long t0 = 0;
  for (int spins = 0; !seen; spins++) {
    seen = __hip_atomic_load(mine + off + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == (tag | (unsigned long long)k);
    if (!seen && (spins & 15) == 15) {
      const long long now = (long long)wall_clock64();
      if (t0 == 0) t0 = now;
      if (now - t0 > max_ticks) break;  // (an exit every lane reaches)
    }
    if (!seen) __builtin_amdgcn_s_sleep(8);
  }
  const unsigned long long all = __ballot(seen);
  if (threadIdx.x == 0) *ok = all == ~0ull ? 1 : 0;
}

// The data words' inboxes in device memory.  Collective: every rank of `c` calls it at the same point, behind the host
// block (which keeps the abort words -- hosts write those -- and is where the data words stay if any rank fails here).
// What the ranks tell each other rides on the communicator's own host all-reduce, a rank's record in its slot of a
// vector of zeros, every number an integer below 2^32 (exact in a float64 sum): pid, HIP device, the inbox's address
// (for the slots of one process) and its 64-byte IPC handle (for everybody else).
void ring_setup_device(pcgx_comm *c) {
  c->dev_tried = true;
  if (!c->ring_dev || c->world < 2) return;
  c->ring_kind = 1;
  const pcgx::RingLayout RL{c->world};
  // the table for the host-memory form (what is used if anything below fails on any rank)
  auto host_table = [&]() {
    c->peers.assign((size_t)c->world, nullptr);
    for (int k = 0; k < c->world; k++) c->peers[(size_t)k] = c->ring_dev + (size_t)k * c->ring_words;
  };
  const char *knob = getenv("PCGX_RING_MEM");  // host: the data words stay in host memory (the round-5 form; measurement)
  const bool want_dev = !(knob && (knob[0] == 'h' || knob[0] == '0'));
  constexpr int kRec = 22;  // pid, device, address (2), ok, handle (16), PCI bus id
  std::vector<double> v((size_t)c->world * kRec, 0.0);
  const size_t bytes = (((size_t)RL.words() + 128) * sizeof(unsigned long long) + 4095) & ~(size_t)4095;  // (+ the trial's slots and its verdict)
  bool mine = want_dev && pcgx::ensure_init() == PCGX_OK;
  int device = -1;
  hipIpcMemHandle_t handle;
  memset(&handle, 0, sizeof handle);
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "the handle travels as sixteen 32-bit words");
  if (mine) {
    mine = hipGetDevice(&device) == hipSuccess;
    // uncached device memory: a peer's store must be seen by the owner's next poll whatever its L2 holds
    if (mine && hipExtMallocWithFlags((void **)&c->inbox, bytes, hipDeviceMallocUncached) != hipSuccess) {
      (void)hipGetLastError();
      c->inbox = nullptr;
      if (hipExtMallocWithFlags((void **)&c->inbox, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        c->inbox = nullptr;
        mine = false;
      }
    }
    if (mine) mine = hipMemset(c->inbox, 0, bytes) == hipSuccess && hipDeviceSynchronize() == hipSuccess;  // (tag 0: nobody's)
    // (the table of the inboxes too, now: nothing is allocated or freed between the ranks' agreement and the trial below --
    // an allocation call in one thread may wait for the device, on which another slot's trial kernel waits for this one's)
    if (mine) mine = hipMalloc((void **)&c->ring_tab, (size_t)c->world * sizeof(void *)) == hipSuccess;
    if (mine) mine = hipIpcGetMemHandle(&handle, c->inbox) == hipSuccess;
    if (!mine) (void)hipGetLastError();
  }
  {
    double *r = v.data() + (size_t)c->rank * kRec;
    const unsigned long long a = (unsigned long long)(uintptr_t)c->inbox;
    r[0] = (double)getpid();
    r[1] = (double)device;
    r[2] = (double)(uint32_t)a;
    r[3] = (double)(uint32_t)(a >> 32);
    r[4] = mine ? 1.0 : 0.0;
    uint32_t w[16];
    memcpy(w, &handle, sizeof w);
    for (int k = 0; k < 16; k++) r[5 + k] = (double)w[k];
    // which physical GPU this rank works on (two processes number their devices as they please): domain:bus:device.function
    char bus[64] = {0};
    int cur = -1;
    unsigned dom = 0, b = 0, d = 0, f = 0;
    if (pcgx::ensure_init() == PCGX_OK && hipGetDevice(&cur) == hipSuccess && hipDeviceGetPCIBusId(bus, (int)sizeof bus, cur) == hipSuccess &&
        sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) == 4)
      r[21] = (double)(1u + ((dom & 0xfffu) << 16 | (b & 0xffu) << 8 | (d & 0x1fu) << 3 | (f & 7u)));
    else
      (void)hipGetLastError();
  }
  bool all = pcgx_comm_allreduce_host_f64(c, v.data(), c->world * kRec) == PCGX_OK;
  c->ranks_share_device = !all;
  for (int k = 0; k < c->world && all; k++) {
    if (v[(size_t)k * kRec + 21] == 0.0) c->ranks_share_device = true;  // (unknown: as if shared)
    for (int j = 0; j < k; j++)
      if (v[(size_t)k * kRec + 21] == v[(size_t)j * kRec + 21]) c->ranks_share_device = true;
  }
  for (int k = 0; k < c->world && all; k++) all = v[(size_t)k * kRec + 4] == 1.0;
  bool ok = all;
  if (ok) {
    c->peers.assign((size_t)c->world, nullptr);
    for (int k = 0; k < c->world && ok; k++) {
      const double *r = v.data() + (size_t)k * kRec;
      if (k == c->rank) {
        c->peers[(size_t)k] = c->inbox;
        continue;
      }
      if ((pid_t)r[0] == getpid()) {  // a device slot of this process: its pointer is mine too
        const int peer_dev = (int)r[1];
        if (peer_dev != device) {
          int can = 0;
          if (hipDeviceCanAccessPeer(&can, device, peer_dev) != hipSuccess || !can) ok = false;
          if (ok) {
            const hipError_t e = hipDeviceEnablePeerAccess(peer_dev, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
            (void)hipGetLastError();
          }
        }
        c->peers[(size_t)k] = (unsigned long long *)(uintptr_t)((unsigned long long)(uint32_t)r[2] | (unsigned long long)(uint32_t)r[3] << 32);
      } else {
        hipIpcMemHandle_t h;
        uint32_t w[16];
        for (int j = 0; j < 16; j++) w[j] = (uint32_t)r[5 + j];
        memcpy(&h, w, sizeof h);
        void *p = nullptr;
        if (hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess) != hipSuccess || !p) {
          (void)hipGetLastError();
          ok = false;
        } else {
          c->ipc_open.push_back(p);
          c->peers[(size_t)k] = (unsigned long long *)p;
        }
      }
    }
  }
  // every rank with every peer's inbox mapped, or nobody: a rank writing into host memory that its peer does not poll
  // would be a Fit that waits out its bound
  double bad = ok ? 0.0 : 1.0;
  bool agreed = pcgx_comm_allreduce_host_f64(c, &bad, 1) == PCGX_OK && bad == 0.0;
  auto drop_device_inboxes = [&]() {
    for (void *p : c->ipc_open) (void)hipIpcCloseMemHandle(p);
    c->ipc_open.clear();
    if (c->inbox) (void)hipFree(c->inbox);
    c->inbox = nullptr;
    (void)hipGetLastError();
    host_table();
  };
  // (on the library's own stream, by name: a copy on the NULL stream waits for every blocking stream of the device -- the
  // slots' streams with a hardware queue of their own are such -- and on one of those another slot's trial kernel may be
  // waiting for THIS rank's word: eight slots of one process then stood in each other's way until a trial ran out of time)
  auto upload_table = [&]() -> bool {
    if (!c->ring_tab && (pcgx::ensure_init() != PCGX_OK || hipMalloc((void **)&c->ring_tab, (size_t)c->world * sizeof(void *)) != hipSuccess)) return false;
    hipStream_t st = pcgx::ctx().stream;
    return hipMemcpyAsync(c->ring_tab, c->peers.data(), (size_t)c->world * sizeof(void *), hipMemcpyHostToDevice, st) == hipSuccess &&
           hipStreamSynchronize(st) == hipSuccess;
  };
  bool tab_ok = true;
  if (!agreed) {
    drop_device_inboxes();
    tab_ok = upload_table();
  } else {
    // the mappings tried out (ring_selftest_kernel): every rank or nobody again
    tab_ok = upload_table();
    int *d_ok = reinterpret_cast<int *>(c->inbox + RL.words() + 64);  // (a word of the inbox's own block: zero since it was made)
    int h_ok = 0;
    bool tried = tab_ok && c->world <= 64;
    if (tried) {
      // (one value on every rank, whatever process or thread it lives in; the inboxes are fresh and zeroed: nothing stale to mistake for it)
      const unsigned long long tag = 0x5e1f7e5700000100ull;
      const bool forced_fail = getenv("PCGX_TEST_RING_SELFTEST_FAIL") != nullptr;  // (tests: the fall-back to host memory)
      hipLaunchKernelGGL(ring_selftest_kernel, dim3(1), dim3(64), 0, pcgx::ctx().stream, (unsigned long long *const *)c->ring_tab, c->inbox, c->world,
                         c->rank, (int)RL.words(), forced_fail && c->rank == c->world - 1 ? tag ^ 1ull << 40 : tag, (long long)200000000 /* 2 s */, d_ok);
      tried = hipMemcpyAsync(&h_ok, d_ok, sizeof(int), hipMemcpyDeviceToHost, pcgx::ctx().stream) == hipSuccess &&
              hipStreamSynchronize(pcgx::ctx().stream) == hipSuccess;
    }
    if (!tried) (void)hipGetLastError();
    double failed = (tried && h_ok == 1) ? 0.0 : 1.0;
    if (getenv("PCGX_RING_TRACE")) fprintf(stderr, "pcgx ring set-up, rank %d: the inboxes' trial %s (launched and read back: %d)\n", c->rank, h_ok == 1 ? "passed" : "FAILED", tried ? 1 : 0);
    const bool all_ok = pcgx_comm_allreduce_host_f64(c, &failed, 1) == PCGX_OK && failed == 0.0;
    if (all_ok) {
      c->ring_kind = 2;
    } else {
      drop_device_inboxes();
      tab_ok = upload_table();
    }
  }
  if (!tab_ok) {  // (no table, no ring: the collectives)
    (void)hipGetLastError();
    c->ring_kind = 0;
  }
  if (getenv("PCGX_RING_TRACE"))
    fprintf(stderr, "pcgx ring set-up, rank %d of %d: inbox %s, every rank's %s, mapped %s, agreed %s -> kind %d\n", c->rank, c->world,
            mine ? "made" : "not made", all ? "made" : "not all made", ok ? "yes" : "no", agreed ? "yes" : "no", c->ring_kind);
  if (c->rank == 0 && c->ring_kind > 0) g_ring_kinds[c->ring_kind == 2 ? 0 : 1]++;
}

}  // namespace

extern "C" pcgx_status pcgx_debug_ring_kinds(int64_t out[2], int32_t reset) {
  if (!out) return pcgx::fail(PCGX_E_INVALID, "pcgx_debug_ring_kinds: NULL argument");
  for (int k = 0; k < 2; k++) {
    out[k] = (int64_t)g_ring_kinds[k].load();
    if (reset) g_ring_kinds[k].store(0);
  }
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_debug_shard_stats(int64_t out[4], int32_t reset) {
  if (!out) return pcgx::fail(PCGX_E_INVALID, "pcgx_debug_shard_stats: NULL argument");
  for (int k = 0; k < 4; k++) {
    out[k] = (int64_t)g_shard_stats[k].load();
    if (reset) g_shard_stats[k].store(0);
  }
  return PCGX_OK;
}

namespace pcgx {

void shard_count(int what) { g_shard_stats[what & 3]++; }

void comm_attach_local_ring(pcgx_comm *c, unsigned long long *block, int32_t words_per_rank) {
  c->ring_tried = true;
  c->ring_host = c->ring_dev = block;
  c->ring_words = words_per_rank;
  if (c->rank == 0) shard_count(2);
}

static void ring_ensure(pcgx_comm *c) {  // collective on first use
  if (!c->ring_tried) ring_setup(c);
  if (!c->dev_tried) ring_setup_device(c);
}

// A Fit begins on the communicator: every rank calls this ONCE per Fit, at the same point of its call sequence (the
// first sharded step of a session since it was made or reset; pcgx_icp_fit_sharded, also on a rank that can go no
// further).  A word's tag is {Fit number, step + 1} (ring_tag): what an earlier Fit left in the inboxes -- data or
// abort words, or what a laggard of that Fit still writes -- carries another Fit's number and is nobody's business;
// and a rank that STOPS stepping in the middle of a Fit (its caller saw an error) is in step with the others again at
// the next Fit, which a count of steps taken would not be.  The owner wipes its own abort word if an earlier Fit's.
// (kRingTagStepBits, strict_terms.h: steps 0 .. 4094 of a Fit ride the ring; beyond: the collective form)
static uint32_t ring_tag(uint32_t fit, int32_t step) { return (fit & 0xfffffu) << kRingTagStepBits | (uint32_t)(step + 1); }
void comm_ring_new_fit(pcgx_comm *c) {
  ring_ensure(c);
  if (++c->ring_fit == 0u || (c->ring_fit & 0xfffffu) == 0u) c->ring_fit = 1u;
  if (!c->ring_host) return;
  const RingLayout RL{c->world};
  unsigned long long *w = c->ring_host + (size_t)c->rank * c->ring_words + RL.abort();
  unsigned long long old = __atomic_load_n(w, __ATOMIC_SEQ_CST);
  while (old != 0ull && (uint32_t)(old >> 32) >> kRingTagStepBits != (c->ring_fit & 0xfffffu))
    if (__atomic_compare_exchange_n(w, &old, 0ull, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) break;
}

int comm_ring_kind(pcgx_comm *c) { return c->ring_kind; }

bool comm_ring_step(pcgx_comm *c, int32_t step, RingView *out) {
  ring_ensure(c);
  if (c->ring_kind == 0 || step < 0 || step + 1 >= (1 << kRingTagStepBits) || c->ring_fit == 0u) {
    shard_count(1);
    return false;
  }
  shard_count(0);
  out->words = c->ring_dev;
  out->host = c->ring_host;
  out->tab = c->ring_tab;
  out->mine = c->peers[(size_t)c->rank];
  out->words_per_rank = c->ring_words;
  out->rank = c->rank;
  out->world = c->world;
  out->epoch = ring_tag(c->ring_fit, step);
  out->kind = c->ring_kind;
  {
    static const long long knob = getenv("PCGX_RING_GUESS_WAIT_US") ? atoll(getenv("PCGX_RING_GUESS_WAIT_US")) * 100 : 0;  // (tests)
    out->guess_ticks = knob > 0 ? knob : (c->ranks_share_device ? kRingGuessTicks : kRingGuessTicksApart);
  }
  return true;
}

// the abort word of every inbox: {reason, tag}, the EARLIEST step of this Fit wins (the host runs ahead of the device:
// a rank that fails while enqueuing step 12 must not wipe out what told the others about step 7); another Fit's word
// is overwritten like an empty one
void ring_abort_from_host(const RingView &ring, uint32_t reason) {
  const RingLayout RL{ring.world};
  const unsigned long long mine = (unsigned long long)ring.epoch << 32 | reason;
  for (int k = 0; k < ring.world; k++) {
    unsigned long long *w = ring.host + (size_t)k * ring.words_per_rank + RL.abort();
    unsigned long long old = __atomic_load_n(w, __ATOMIC_SEQ_CST);
    while (true) {
      const uint32_t tag = (uint32_t)(old >> 32);
      const bool earlier_of_this_fit = tag != 0u && tag >> kRingTagStepBits == ring.epoch >> kRingTagStepBits && tag <= ring.epoch;
      if (earlier_of_this_fit) break;
      if (__atomic_compare_exchange_n(w, &old, mine, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) break;
    }
  }
}

}  // namespace pcgx

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
  ring_unmap(c);
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
