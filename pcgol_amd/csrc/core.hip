// core.hip -- context, error reporting, workspace arena, device-memory helpers
// and the host-only math exports of the pcgx C ABI.
#include <stdarg.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <vector>

#include "pcgx_internal.h"

namespace pcgx {

static thread_local std::string g_last_error;

const char *last_error_text() { return g_last_error.c_str(); }

pcgx_status fail(pcgx_status code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return code;
}

namespace {
struct CachedBlock {
  void *p;
  size_t cap;
  int device;  // the HIP device the block lives on (one process may drive several: pcgx_init_devices)
};
std::vector<CachedBlock> &cache_free_list() {
  static std::vector<CachedBlock> v;
  return v;
}
std::vector<CachedBlock> &cache_live_list() {
  static std::vector<CachedBlock> v;
  return v;
}
constexpr size_t kCacheLimitBytes = (size_t)2 << 30;
std::mutex g_cache_mu;
void cache_release_all_locked() {
  for (auto &b : cache_free_list()) (void)hipFree(b.p);
  cache_free_list().clear();
}
}  // namespace

void dev_cache_quiesce() { (void)hipDeviceSynchronize(); }

hipError_t dev_cache_alloc(void **ptr, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  bytes = bytes ? bytes : 1;
  int device = 0;
  (void)hipGetDevice(&device);
  auto &fl = cache_free_list();
  size_t best = fl.size();
  for (size_t i = 0; i < fl.size(); i++)
    if (fl[i].device == device && fl[i].cap >= bytes && fl[i].cap <= 2 * bytes + 4096 &&
        (best == fl.size() || fl[i].cap < fl[best].cap))
      best = i;
  if (best != fl.size()) {
    *ptr = fl[best].p;
    cache_live_list().push_back(fl[best]);
    fl.erase(fl.begin() + (long)best);
    return hipSuccess;
  }
  const size_t cap = (bytes + 255) & ~(size_t)255;
  hipError_t e = hipMalloc(ptr, cap);
  if (e != hipSuccess) {  // make room and try once more
    cache_release_all_locked();
    e = hipMalloc(ptr, cap);
  }
  if (e == hipSuccess) cache_live_list().push_back(CachedBlock{*ptr, cap, device});
  return e;
}

void dev_cache_free(void *ptr) {
  if (!ptr) return;
  std::lock_guard<std::mutex> lk(g_cache_mu);
  auto &ll = cache_live_list();
  for (size_t i = 0; i < ll.size(); i++)
    if (ll[i].p == ptr) {
      auto &fl = cache_free_list();
      size_t held = 0;
      for (auto &b : fl) held += b.cap;
      if (held + ll[i].cap <= kCacheLimitBytes) fl.push_back(ll[i]);
      else (void)hipFree(ptr);
      ll.erase(ll.begin() + (long)i);
      return;
    }
  (void)hipFree(ptr);  // not one of ours
}

void dev_cache_release_all() {
  std::lock_guard<std::mutex> lk(g_cache_mu);
  cache_release_all_locked();
}

// ---- call contexts -----------------------------------------------------------
namespace {
constexpr int kPoolSlots = 4;
struct Global {
  Context slots[1 + kPoolSlots];  // [0]: the library's context; [1..]: the pool
  bool busy[1 + kPoolSlots] = {};
  std::mutex mu;                 // busy[]
  std::condition_variable cv;
  std::recursive_mutex lib_mu;   // context 0, one caller at a time
  std::mutex init_mu;
  int in_pool = 0, peak_in_pool = 0;
  long long pooled_calls = 0;
};
// One Global per DEVICE SLOT.  Slot 0 is what a process that drives one GPU uses and never thinks about
// (pcgx_init).  A process that drives several (pcgx_init_devices) gives every slot its HIP device -- for tests, the
// same device several times -- and every host thread says which slot its calls are for (pcgx_set_device, as with
// HIP's own current device); handles live on the slot they were made on.
constexpr int kMaxSlots = 16;
Global g_slots[kMaxSlots];
int g_slot_device[kMaxSlots] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};  // -1: pcgx_init's choice
int g_num_slots = 1;
bool g_private_queues = false;  // pcgx_init_devices found several slots on one device
thread_local int tl_slot = 0;
Global &glob() { return g_slots[tl_slot]; }
thread_local Context *tl_ctx = nullptr;
thread_local int tl_depth = 0;
}  // namespace

int current_slot() { return tl_slot; }

Context &ctx() { return tl_ctx ? *tl_ctx : glob().slots[0]; }

CallScope::CallScope(bool pooled) {
  Global &g = glob();
  if (tl_depth++ > 0) return;  // nested: the outer call's context
  if (pooled) {
    std::unique_lock<std::mutex> lk(g.mu);
    for (;;) {
      for (int k = 1; k <= kPoolSlots && slot_ < 0; k++)
        if (!g.busy[k]) slot_ = k;
      if (slot_ >= 0) break;
      g.cv.wait(lk);
    }
    g.busy[slot_] = true;
    g.pooled_calls++;
    if (++g.in_pool > g.peak_in_pool) g.peak_in_pool = g.in_pool;
  } else {
    g.lib_mu.lock();
    slot_ = 0;
  }
  tl_ctx = &g.slots[slot_];
  if (g.slots[0].ready) (void)hipSetDevice(g.slots[0].device);
}

CallScope::~CallScope() {
  if (--tl_depth > 0) return;
  Global &g = glob();
  tl_ctx = nullptr;
  if (slot_ == 0) {
    g.lib_mu.unlock();
  } else if (slot_ > 0) {
    {
      std::lock_guard<std::mutex> lk(g.mu);
      g.busy[slot_] = false;
      g.in_pool--;
    }
    g.cv.notify_one();
  }
}

static pcgx_status init_device(int device) {
  // (HIP hands its streams to a few hardware queues, four by default: the library's own stream, its four pooled call
  // contexts and the host's streams share them, and kernels of independent calls queue up behind one another -- four
  // host-pointer Fits in flight: 4.6 ms with four queues, 4.0 with GPU_MAX_HW_QUEUES=8, tools/conc4_probe.py.  The host's
  // to set, not the library's: with several PROCESSES on one GPU -- the sharded tests and rehearsals on the one-GPU box --
  // more queues than the hardware has are time-sliced by the driver, and a kernel that waits for another process's
  // kernel then waits for a time slice: a skewed three-rank Fit took 97 ms instead of 10 when this library asked for 8.)
  Global &g = glob();
  std::lock_guard<std::mutex> lk(g.init_mu);
  Context &c = g.slots[0];
  if (c.ready) {
    if (device >= 0 && device != c.device)
      return fail(PCGX_E_INVALID, "pcgx already initialised on device %d (one process per GPU)", c.device);
    return PCGX_OK;
  }
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(PCGX_E_HIP, "no HIP device available (%s): libpcgx has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  if (device < 0) device = g_slot_device[tl_slot] >= 0 ? g_slot_device[tl_slot] : 0;  // (pcgx_init_devices' choice for this slot)
  if (device >= count) return fail(PCGX_E_INVALID, "device %d out of range (%d devices)", device, count);
  PCGX_HIP_TRY(hipSetDevice(device));
  hipDeviceProp_t prop;
  PCGX_HIP_TRY(hipGetDeviceProperties(&prop, device));
  for (int k = 0; k <= kPoolSlots; k++) {
    Context &s = g.slots[k];
    s.num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // (PCGX_POOL_QUEUES=1, measurement: the pooled contexts' streams with a hardware queue each too)
    static const bool pool_queues = getenv("PCGX_POOL_QUEUES") && atoi(getenv("PCGX_POOL_QUEUES")) != 0;
    if ((k == 0 && g_private_queues) || (k > 0 && pool_queues)) {
      // Slots that share ONE device (pcgx_init_devices on a test box): the library's own stream of every slot gets a
      // hardware queue to itself.  HIP hands its few hardware queues to streams as they have work, so two slots'
      // streams may land in one queue -- and the ring form of the sharded sums has slot A's kernel wait for slot B's:
      // queued behind it, B's never starts (measured: a Fit over eight slots stood still until A's wait ran out of
      // time).  A stream made with a CU mask owns its queue; the mask names every CU.
      uint32_t mask[32];
      for (auto &m : mask) m = 0xffffffffu;
      const uint32_t words = (uint32_t)((s.num_cu + 31) / 32);
      if (hipExtStreamCreateWithCUMask(&s.stream, words < 32u ? words : 32u, mask) != hipSuccess) {
        (void)hipGetLastError();
        s.stream = nullptr;
      }
    }
    if (!s.stream) PCGX_HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
    void *mb = nullptr;
    if (hipHostMalloc(&mb, kMailboxBytes, hipHostMallocDefault) == hipSuccess) {  // (not required: without it results are copied and waited for)
      memset(mb, 0, kMailboxBytes);
      s.mailbox = (volatile uint32_t *)mb;
    } else {
      (void)hipGetLastError();
    }
    s.mailbox_seq = 0;
    if (hipHostMalloc(&s.up, kSmallUploadBytes, hipHostMallocDefault) != hipSuccess ||
        hipEventCreateWithFlags(&s.up_read, hipEventDisableTiming) != hipSuccess) {  // (not required)
      (void)hipGetLastError();
      if (s.up) (void)hipHostFree(s.up);
      s.up = nullptr;
    }
    s.up_pending = false;
    PCGX_HIP_TRY(hipMalloc((void **)&s.tickets, kTicketBytes));
    PCGX_HIP_TRY(hipMemset(s.tickets, 0, kTicketBytes));
    s.device = device;
  }
  for (int k = kPoolSlots; k >= 0; k--) g.slots[k].ready = true;
  return PCGX_OK;
}

pcgx_status ensure_init() {
  if (glob().slots[0].ready) {
    (void)hipSetDevice(glob().slots[0].device);  // this thread may not have been bound yet (first call of its scope)
    return PCGX_OK;
  }
  return init_device(-1);
}

// ---- kernel timing -----------------------------------------------------------
namespace {
struct ProfRec { hipEvent_t a, b; int kind; };
bool g_prof_on = false;
int g_prof_stride = 1;                 // time every n-th scope of a kind
int64_t g_prof_seen[PCGX_PROF_KINDS];  // scopes opened per kind
std::vector<ProfRec> g_prof_pending;
std::vector<hipEvent_t> g_prof_pool;
double g_prof_ms[PCGX_PROF_KINDS];
double g_prof_max_ms[PCGX_PROF_KINDS];
int64_t g_prof_n[PCGX_PROF_KINDS];
std::mutex g_prof_mu;

hipEvent_t prof_event() {
  if (!g_prof_pool.empty()) {
    hipEvent_t e = g_prof_pool.back();
    g_prof_pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
void prof_resolve() {
  for (auto &r : g_prof_pending) {
    float ms = 0.0f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      g_prof_ms[r.kind] += ms;
      g_prof_n[r.kind] += 1;
      if ((double)ms > g_prof_max_ms[r.kind]) g_prof_max_ms[r.kind] = (double)ms;
    }
    g_prof_pool.push_back(r.a);
    g_prof_pool.push_back(r.b);
  }
  g_prof_pending.clear();
}
}  // namespace

ProfScope::ProfScope(int kind, hipStream_t st) : kind_(kind), st_(st) {
  if (!g_prof_on || kind < 0) return;  // kind < 0: not timed
  std::lock_guard<std::mutex> lk(g_prof_mu);
  if (g_prof_seen[kind]++ % g_prof_stride != 0) return;
  a_ = prof_event();
  b_ = prof_event();
  if (a_) (void)hipEventRecord(a_, st_);
}
ProfScope::~ProfScope() {
  if (!a_ || !b_) return;
  std::lock_guard<std::mutex> lk(g_prof_mu);
  (void)hipEventRecord(b_, st_);
  g_prof_pending.push_back(ProfRec{a_, b_, kind_});
}

// ---- arena -----------------------------------------------------------------
static size_t round_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

pcgx_status Arena::begin(hipStream_t st) {
  if (has_last_ && last_stream_ != st) {
    // a different stream may still be using the previous call's temporaries
    PCGX_HIP_TRY(hipStreamSynchronize(last_stream_));
  }
  last_stream_ = st;
  has_last_ = true;
  if (blocks_.size() > 1) {
    // One block from the next call on.  The blocks go back to the block cache, not to hipFree: hipFree waits for the
    // WHOLE device -- a context that grows its arena then stands still until every other context's Fit has drained
    // (four host-pointer Fits in flight: 13 ms in a session's set-up, tools/conc4_probe.py).  What may still read the
    // old blocks is this context's own earlier work: waited for here, on this stream alone.
    PCGX_HIP_TRY(hipStreamSynchronize(st));
    size_t total = 0;
    for (auto &b : blocks_) total += b.cap;
    release_all();
    total = round_up(total + total / 4, 1 << 20);
    uint8_t *p = nullptr;
    hipError_t e = dev_cache_alloc((void **)&p, total);
    if (e != hipSuccess) return fail(PCGX_E_OOM, "arena hipMalloc(%zu) failed: %s", total, hipGetErrorString(e));
    blocks_.push_back(Block{p, total, 0});
  }
  for (auto &b : blocks_) b.used = 0;
  return PCGX_OK;
}

pcgx_status Arena::alloc(size_t bytes, void **out) {
  bytes = round_up(bytes ? bytes : 1, 256);
  if (!blocks_.empty()) {
    Block &b = blocks_.back();
    if (b.used + bytes <= b.cap) {
      *out = b.p + b.used;
      b.used += bytes;
      return PCGX_OK;
    }
  }
  size_t cap = round_up(bytes, 1 << 20);
  uint8_t *p = nullptr;
  hipError_t e = dev_cache_alloc((void **)&p, cap);
  if (e != hipSuccess) return fail(PCGX_E_OOM, "arena hipMalloc(%zu) failed: %s", cap, hipGetErrorString(e));
  blocks_.push_back(Block{p, cap, bytes});
  *out = p;
  return PCGX_OK;
}

void Arena::release_all() {
  for (auto &b : blocks_) dev_cache_free(b.p);
  blocks_.clear();
}

pcgx_status Arena::zeroed_words(size_t count, uint32_t **out) {
  if (!words_) {
    constexpr size_t kWords = 8192;
    hipError_t e = hipMalloc((void **)&words_, kWords * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemset(words_, 0, kWords * sizeof(uint32_t));  // (once; synchronous)
    if (e != hipSuccess) {
      (void)hipFree(words_);
      words_ = nullptr;
      return fail(PCGX_E_OOM, "arena: no memory for the persistent counters: %s", hipGetErrorString(e));
    }
    n_words_ = kWords;
  }
  if (count > n_words_) return fail(PCGX_E_INVALID, "arena: %zu persistent words asked for, %zu there", count, n_words_);
  *out = words_;
  return PCGX_OK;
}

void Arena::release_words() {
  (void)hipFree(words_);
  words_ = nullptr;
  n_words_ = 0;
  turn_ = 0;
}

}  // namespace pcgx

using namespace pcgx;

extern "C" pcgx_status pcgx_init(int32_t device) { return init_device(device); }

// One process, several GPUs: slot k of the library works on HIP device device_ids[k] (NULL: device k).  The same
// device may be named more than once (several independent sets of streams and workspaces on one GPU: how the
// several-GPU paths are tested on a one-GPU box).  Call before any other entry point, or with the assignment already
// in force.
extern "C" pcgx_status pcgx_init_devices(int32_t n, const int32_t *device_ids) {
  if (n < 1 || n > kMaxSlots) return fail(PCGX_E_INVALID, "pcgx_init_devices: 1 .. %d device slots", kMaxSlots);
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count == 0)
    return fail(PCGX_E_HIP, "no HIP device available (%s): libpcgx has no CPU fallback",
                e != hipSuccess ? hipGetErrorString(e) : "device count 0");
  for (int k = 0; k < n; k++) {
    const int d = device_ids ? device_ids[k] : k;
    if (d < 0 || d >= count) return fail(PCGX_E_INVALID, "pcgx_init_devices: device %d out of range (%d devices)", d, count);
    if (g_slots[k].slots[0].ready && g_slots[k].slots[0].device != d)
      return fail(PCGX_E_INVALID, "pcgx_init_devices: slot %d already works on device %d", k, g_slots[k].slots[0].device);
  }
  for (int a = 0; a < n; a++)
    for (int b = a + 1; b < n; b++)
      if ((device_ids ? device_ids[a] : a) == (device_ids ? device_ids[b] : b)) g_private_queues = true;
  const int keep = tl_slot;
  pcgx_status rc = PCGX_OK;
  for (int k = 0; k < n && rc == PCGX_OK; k++) {
    g_slot_device[k] = device_ids ? device_ids[k] : k;
    tl_slot = k;
    rc = init_device(g_slot_device[k]);
  }
  tl_slot = keep;
  if (rc == PCGX_OK && n > g_num_slots) g_num_slots = n;
  if (g_slots[tl_slot].slots[0].ready) (void)hipSetDevice(g_slots[tl_slot].slots[0].device);
  return rc;
}

// The calling thread's device slot for the calls that follow (default 0), as hipSetDevice is for HIP.
extern "C" pcgx_status pcgx_set_device(int32_t slot) {
  if (slot < 0 || slot >= kMaxSlots) return fail(PCGX_E_INVALID, "pcgx_set_device: slot %d out of range", slot);
  if (tl_depth > 0) return fail(PCGX_E_INVALID, "pcgx_set_device: inside a library call");
  if (slot >= g_num_slots && !g_slots[slot].slots[0].ready)
    return fail(PCGX_E_INVALID, "pcgx_set_device: slot %d was not set up (pcgx_init_devices)", slot);
  tl_slot = slot;
  if (g_slots[slot].slots[0].ready) (void)hipSetDevice(g_slots[slot].slots[0].device);
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_get_device(int32_t *slot, int32_t *hip_device) {
  if (slot) *slot = tl_slot;
  if (hip_device) *hip_device = g_slots[tl_slot].slots[0].ready ? g_slots[tl_slot].slots[0].device : -1;
  return PCGX_OK;
}

static void shutdown_slot() {
  PCGX_API_LOCK();
  Global &g = glob();
  std::lock_guard<std::mutex> lk(g.init_mu);
  if (!g.slots[0].ready) return;
  (void)hipSetDevice(g.slots[0].device);
  (void)hipDeviceSynchronize();
  for (int k = 0; k <= kPoolSlots; k++) {
    Context &c = g.slots[k];
    c.arena.release_all();
    c.arena.release_words();
    c.host_arena.release_all();
    (void)hipStreamDestroy(c.stream);
    c.stream = nullptr;
    if (c.mailbox) (void)hipHostFree((void *)c.mailbox);
    c.mailbox = nullptr;
    if (c.up) (void)hipHostFree(c.up);
    c.up = nullptr;
    if (c.up_read) (void)hipEventDestroy(c.up_read);
    c.up_read = nullptr;
    c.up_pending = false;
    if (c.tickets) (void)hipFree(c.tickets);
    c.tickets = nullptr;
    c.ready = false;
    c.device = -1;
  }
}

extern "C" pcgx_status pcgx_shutdown(void) {
  if (tl_depth > 0) return fail(PCGX_E_INVALID, "pcgx_shutdown: inside a library call");
  const int keep = tl_slot;
  for (int k = 0; k < kMaxSlots; k++) {
    tl_slot = k;
    shutdown_slot();
    g_slot_device[k] = -1;
  }
  tl_slot = keep;
  g_num_slots = 1;
  dev_cache_release_all();
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_debug_call_stats(int64_t out[2], int32_t reset) {
  if (!out) return fail(PCGX_E_INVALID, "pcgx_debug_call_stats: NULL argument");
  Global &g = glob();
  std::lock_guard<std::mutex> lk(g.mu);
  out[0] = g.peak_in_pool;
  out[1] = g.pooled_calls;
  if (reset) {
    g.peak_in_pool = g.in_pool;
    g.pooled_calls = 0;
  }
  return PCGX_OK;
}

extern "C" int32_t pcgx_last_error(char *buf, size_t cap) {
  const std::string &s = g_last_error;
  if (buf && cap > 0) {
    size_t n = s.size() < cap - 1 ? s.size() : cap - 1;
    memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return (int32_t)s.size();
}

extern "C" const char *pcgx_version(void) { return "pcgx 0.1 (gfx950)"; }
extern "C" int32_t pcgx_abi_version(void) { return PCGX_ABI_VERSION; }
extern "C" pcgx_status pcgx_icp_params_init(pcgx_icp_params *p, size_t sizeof_params) {
  if (!p) return fail(PCGX_E_INVALID, "pcgx_icp_params_init: NULL argument");
  if (sizeof_params != sizeof(pcgx_icp_params))
    return fail(PCGX_E_INVALID, "pcgx_icp_params_init: the caller's pcgx_icp_params has %zu bytes, the library's %zu (ABI version %d)",
                sizeof_params, sizeof(pcgx_icp_params), PCGX_ABI_VERSION);
  memset(p, 0, sizeof *p);
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_sync(void *stream) {
  PCGX_API_LOCK();
  PCGX_TRY(ensure_init());
  PCGX_HIP_TRY(hipStreamSynchronize(pick_stream(stream)));
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_prof_enable(int32_t on) {
  PCGX_API_LOCK();
  PCGX_TRY(ensure_init());
  std::lock_guard<std::mutex> lk(g_prof_mu);
  g_prof_on = on != 0;
  g_prof_stride = on > 1 ? on : 1;
  for (int k = 0; k < PCGX_PROF_KINDS; k++) g_prof_seen[k] = 0;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_prof_read(int32_t kind, double *total_ms, int64_t *launches) {
  PCGX_API_LOCK();
  if (kind < 0 || kind >= PCGX_PROF_KINDS || !total_ms || !launches)
    return fail(PCGX_E_INVALID, "pcgx_prof_read: bad argument");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_resolve();
  *total_ms = g_prof_ms[kind];
  *launches = g_prof_n[kind];
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_prof_reset(void) {
  PCGX_API_LOCK();
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_resolve();
  for (int k = 0; k < PCGX_PROF_KINDS; k++) { g_prof_ms[k] = 0.0; g_prof_max_ms[k] = 0.0; g_prof_n[k] = 0; }
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_prof_read_max(int32_t kind, double *max_ms) {
  PCGX_API_LOCK();
  if (kind < 0 || kind >= PCGX_PROF_KINDS || !max_ms) return fail(PCGX_E_INVALID, "pcgx_prof_read_max: bad argument");
  std::lock_guard<std::mutex> lk(g_prof_mu);
  prof_resolve();
  *max_ms = g_prof_max_ms[kind];
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_dev_alloc(size_t bytes, void **dptr) {
  PCGX_API_LOCK();
  if (!dptr) return fail(PCGX_E_INVALID, "pcgx_dev_alloc: dptr is NULL");
  PCGX_TRY(ensure_init());
  hipError_t e = hipMalloc(dptr, bytes ? bytes : 1);
  if (e != hipSuccess) return fail(PCGX_E_OOM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_dev_free(void *dptr) {
  PCGX_API_LOCK();
  if (dptr) PCGX_HIP_TRY(hipFree(dptr));
  return PCGX_OK;
}

// ---- host <-> device copies of the host-pointer seams ----------------------------------------------
// The slices a Go caller holds are pageable memory.  Measured on the test box (tools/staging_probe.cpp,
// tools/stage_probe.py): the runtime's own pageable path moves 120 MB at 56 GB/s once a process has made its
// first large copy (19 ms for the very first one), fresh or reused buffers alike; a pinned ring of three
// 16 MB slots filled by six copy threads reached 45-47 GB/s -- built, measured, removed.  What made these
// seams slow in round 2 was host-side: ids widened to Go's 64-bit int in a host loop, zero-filled
// temporaries, and (in the measurement itself) output arrays allocated per call, whose first touch is a
// page fault per 4 KB.  The ids are widened on the device now; these two are the copies, in one place.
namespace pcgx {
// (Large copies one at a time: the runtime pins the caller's pages for the transfer, and four threads doing that at once --
// four host-pointer Fits in flight, 12 MB each -- stood in each other's way for up to 7.5 ms where one copy alone
// takes 0.4: tools/conc4_probe.py with PCGX_FIT_TRACE.  The copies overlap with the other contexts' kernels all the same.)
pcgx_status staged_upload(void *d_dst, const void *h_src, size_t bytes, hipStream_t st) {
  static std::mutex big_copy;
  if (bytes >= ((size_t)1 << 20)) {
    std::lock_guard<std::mutex> lk(big_copy);
    PCGX_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
  } else if (bytes) {
    PCGX_HIP_TRY(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, st));
  }
  return PCGX_OK;
}

const void *small_upload(const void *h_src, size_t bytes) {
  Context &c = ctx();
  if (!c.up || bytes > kSmallUploadBytes) return nullptr;
  if (c.up_pending) {
    if (hipEventQuery(c.up_read) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    c.up_pending = false;
  }
  memcpy(c.up, h_src, bytes);
  return c.up;
}
void small_upload_read(hipStream_t st) {
  Context &c = ctx();
  if (hipEventRecord(c.up_read, st) == hipSuccess) {
    c.up_pending = true;
  } else {  // (cannot tell when it is read: wait for it now)
    (void)hipGetLastError();
    (void)hipStreamSynchronize(st);
  }
}

pcgx_status staged_download(void *h_dst, const void *d_src, size_t bytes, hipStream_t st) {
  if (bytes) PCGX_HIP_TRY(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, st));
  PCGX_HIP_TRY(hipStreamSynchronize(st));
  return PCGX_OK;
}
}  // namespace pcgx

extern "C" pcgx_status pcgx_dev_upload(void *dptr, const void *host, size_t bytes) {
  PCGX_API_LOCK();
  PCGX_TRY(ensure_init());
  if (bytes) PCGX_HIP_TRY(hipMemcpy(dptr, host, bytes, hipMemcpyHostToDevice));
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_dev_download(void *host, const void *dptr, size_t bytes) {
  PCGX_API_LOCK();
  PCGX_TRY(ensure_init());
  if (bytes) PCGX_HIP_TRY(hipMemcpy(host, dptr, bytes, hipMemcpyDeviceToHost));
  return PCGX_OK;
}

// ---- host-only math exports (no GPU needed) ---------------------------------

extern "C" pcgx_status pcgx_rodrigues(const float v[3], float out16[16]) {
  PCGX_API_LOCK();
  if (!v || !out16) return fail(PCGX_E_INVALID, "pcgx_rodrigues: NULL argument");
  Mat4 r = rodrigues_to_rotation(v[0], v[1], v[2]);
  memcpy(out16, r.m, sizeof r.m);
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_mat4_mul(const float m[16], const float a[16], float out16[16]) {
  PCGX_API_LOCK();
  if (!m || !a || !out16) return fail(PCGX_E_INVALID, "pcgx_mat4_mul: NULL argument");
  Mat4 x, y;
  memcpy(x.m, m, sizeof x.m);
  memcpy(y.m, a, sizeof y.m);
  Mat4 r = mat4_mul(x, y);
  memcpy(out16, r.m, sizeof r.m);
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_mat4_transform(const float m[16], const float *xyz, int64_t n,
                                           float *out_xyz) {
  PCGX_API_LOCK();
  if (!m || n < 0 || (n > 0 && (!xyz || !out_xyz))) return fail(PCGX_E_INVALID, "pcgx_mat4_transform: bad argument");
  for (int64_t i = 0; i < n; i++) {
    float x, y, z;
    mat4_transform(m, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], x, y, z);
    out_xyz[3 * i] = x;
    out_xyz[3 * i + 1] = y;
    out_xyz[3 * i + 2] = z;
  }
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_finish_evaluate(const double sums10[10], int32_t min_pairs,
                                                pcgx_icp_evaluated *out) {
  PCGX_API_LOCK();
  if (!sums10 || !out) return fail(PCGX_E_INVALID, "pcgx_icp_finish_evaluate: NULL argument");
  if (min_pairs == 0) min_pairs = 6;  // evaluator.go:92-95
  const int64_t npairs = (int64_t)sums10[S_PAIRS];
  out->num_pairs = npairs;
  if (npairs < min_pairs)
    return fail(PCGX_E_NOT_ENOUGH_PAIRS, "not enough correspondence pairs (%lld < %d)", (long long)npairs, min_pairs);
  Evaluated ev;
  finish_evaluate(sums10, ev);
  out->value = ev.value;
  memcpy(out->gradient, ev.gradient, sizeof ev.gradient);
  out->dist_rms = ev.dist_rms;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_plane_finish_evaluate(const double sums30[30], int32_t min_pairs,
                                                      pcgx_icp_evaluated *out, float hessian36[36]) {
  PCGX_API_LOCK();
  if (!sums30 || !out || !hessian36) return fail(PCGX_E_INVALID, "pcgx_icp_plane_finish_evaluate: NULL argument");
  if (min_pairs == 0) min_pairs = 6;
  const int64_t npairs = (int64_t)sums30[P_PAIRS];
  out->num_pairs = npairs;
  if (npairs < min_pairs)
    return fail(PCGX_E_NOT_ENOUGH_PAIRS, "not enough correspondence pairs (%lld < %d)", (long long)npairs, min_pairs);
  EvaluatedPlane ev;
  finish_evaluate_plane(sums30, ev);
  out->value = ev.value;
  memcpy(out->gradient, ev.gradient, sizeof ev.gradient);
  out->dist_rms = 0.0f;
  memcpy(hessian36, ev.hessian, sizeof ev.hessian);
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_gauss_newton_update(const pcgx_icp_params *p, float damping, int32_t *iter,
                                                    const float gradient[6], const float hessian36[36],
                                                    float trans16[16], int32_t *converged) {
  PCGX_API_LOCK();
  if (!p || !iter || !gradient || !hessian36 || !trans16 || !converged)
    return fail(PCGX_E_INVALID, "pcgx_icp_gauss_newton_update: NULL argument");
  GaussNewtonParams u = resolve_gauss_newton(p->threshold, damping, p->max_iteration);
  EvaluatedPlane ev;
  memset(&ev, 0, sizeof ev);
  memcpy(ev.gradient, gradient, sizeof ev.gradient);
  memcpy(ev.hessian, hessian36, sizeof ev.hessian);
  Mat4 t;
  memcpy(t.m, trans16, sizeof t.m);
  const int rc = gauss_newton_update(u, *iter, ev, t);
  if (rc < 0) return fail(PCGX_E_SINGULAR, "normal equations are not positive definite");
  memcpy(trans16, t.m, sizeof t.m);
  *converged = rc;
  return PCGX_OK;
}

extern "C" pcgx_status pcgx_icp_update(const pcgx_icp_params *p, int32_t *iter, const float gradient[6],
                                       float trans16[16], int32_t *converged) {
  PCGX_API_LOCK();
  if (!p || !iter || !gradient || !trans16 || !converged)
    return fail(PCGX_E_INVALID, "pcgx_icp_update: NULL argument");
  UpdaterParams u = resolve_updater(p->weight, p->threshold, p->max_iteration);
  Mat4 t;
  memcpy(t.m, trans16, sizeof t.m);
  bool c = gradient_descent_update(u, *iter, gradient, t);
  memcpy(trans16, t.m, sizeof t.m);
  *converged = c ? 1 : 0;
  return PCGX_OK;
}
