// pcgx_internal.h -- library-internal declarations (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <atomic>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/pcgx.h"
#include "pcgx_math.h"

namespace pcgx {

// std::vector whose resize() leaves new elements uninitialised (buffers that are overwritten at
// once: zero-filling 12 MB first costs as much as the copy that follows)
template <class T>
struct DefaultInitAlloc : std::allocator<T> {
  template <class U>
  struct rebind {
    using other = DefaultInitAlloc<U>;
  };
  template <class U>
  void construct(U *p) noexcept {
    ::new ((void *)p) U;
  }
  template <class U, class... A>
  void construct(U *p, A &&...a) {
    ::new ((void *)p) U(std::forward<A>(a)...);
  }
};
template <class T>
using RawVector = std::vector<T, DefaultInitAlloc<T>>;

// ---- error handling --------------------------------------------------------
pcgx_status fail(pcgx_status code, const char *fmt, ...);
const char *last_error_text();  // the calling thread's last error message (what pcgx_last_error copies)

#define PCGX_HIP_TRY(expr)                                                             \
  do {                                                                                 \
    hipError_t e__ = (expr);                                                           \
    if (e__ != hipSuccess)                                                             \
      return ::pcgx::fail(e__ == hipErrorOutOfMemory ? PCGX_E_OOM : PCGX_E_HIP,        \
                          "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),      \
                          __FILE__, __LINE__);                                         \
  } while (0)

#define PCGX_TRY(expr)                 \
  do {                                 \
    pcgx_status s__ = (expr);          \
    if (s__ != PCGX_OK) return s__;    \
  } while (0)

// ---- workspace arena -------------------------------------------------------
// Call-scoped device temporaries are bump-allocated from a grow-only arena so
// that steady-state calls perform no hipMalloc/hipFree.  begin() is called on
// entry of every API function that needs temporaries; work enqueued by the
// previous call on the same stream is ordered before the new work, so reuse
// is safe.  If a block overflows, a new one is added and the blocks are
// consolidated at the next begin() (hipFree synchronises the device).
class Arena {
 public:
  pcgx_status begin(hipStream_t st);
  pcgx_status alloc(size_t bytes, void **out);
  template <class T>
  pcgx_status alloc_n(size_t count, T **out) {
    return alloc(count * sizeof(T), (void **)out);
  }
  void release_all();
  // `count` zeroed words that LIVE ACROSS calls (same ordering rules as the temporaries: touched only behind begin()):
  // counters a call's kernels leave at zero again for the next call -- a hipMemsetAsync per call is a launch of its
  // own (4.5 us at the head of a 0.11 ms Nearest batch).  One region per arena; `count` may not grow.
  pcgx_status zeroed_words(size_t count, uint32_t **out);
  void release_words();
  // 0, 1, 0, ...: a word that must outlive its call (a count the call's last kernel still reads) has a twin; a call
  // uses one and leaves the other at zero for the next call
  uint32_t take_turn() { return turn_++ & 1u; }

 private:
  uint32_t *words_ = nullptr;
  size_t n_words_ = 0;
  uint32_t turn_ = 0;
  struct Block {
    uint8_t *p;
    size_t cap, used;
  };
  std::vector<Block> blocks_;
  hipStream_t last_stream_ = nullptr;
  bool has_last_ = false;
};

// ---- cached device blocks -----------------------------------------------------
// Short-lived objects with device state (ICP sessions: the reference's own Fit loop calls
// Evaluate once per iteration, i.e. one session per iteration through the Go shim) take their
// buffers from a small cache of blocks instead of hipMalloc / hipFree, which cost more than the
// whole evaluation at 1M points.  A freed block is kept for reuse (first fit among blocks at most
// twice the requested size); the cache holds at most 2 GiB, anything beyond is hipFree'd.
hipError_t dev_cache_alloc(void **ptr, size_t bytes);
void dev_cache_free(void *ptr);
void dev_cache_release_all();

// ---- kernel timing (pcgx_prof_*) ---------------------------------------------
struct ProfScope {
  ProfScope(int kind, hipStream_t st);
  ~ProfScope();
  int kind_;
  hipStream_t st_;
  hipEvent_t a_ = nullptr, b_ = nullptr;
};

// ---- context ---------------------------------------------------------------
// One process drives one GPU.  Work is enqueued from CALL CONTEXTS: a stream plus the two workspace
// arenas.  Context 0 is the library's own: every entry point that takes a `stream` argument (the
// device-resident `_dev` calls, ICP sessions) runs in it, one caller at a time, so that work given
// to the NULL stream stays ordered whatever OS thread a goroutine happens to be on.  The blocking
// host-pointer entry points (Nearest / Range batches, VoxelGrid.Filter, MinMax, tree build, Fit,
// Evaluate, Pairs -- the seams the reference's own interfaces map to, which it allows to be called
// from several goroutines at once, kdtree.go:44-50,72-81) take a context from a small pool instead:
// handles are immutable after build, so these calls overlap on the GPU, each on its own stream with
// its own workspace, and synchronise only their own stream before they return.
struct Context {
  bool ready = false;
  int device = -1;
  hipStream_t stream = nullptr;
  int num_cu = 256;
  Arena arena;       // temporaries of the device-resident (_dev) entry points
  Arena host_arena;  // device staging of the host-pointer entry points (they call the _dev ones,
                     // which restart `arena`; separate so the staging survives that)
  // 64 words of pinned host memory the device can write: small results the host waits for in the middle of a call
  // (the voxel filter's min / max) land here directly, behind a sequence word the host spins on -- a copy to pageable
  // memory plus hipStreamSynchronize cost 30 us of idle GPU per call
  volatile uint32_t *mailbox = nullptr;
  uint32_t mailbox_seq = 0;
  // zeroed device words (kTicketBytes) that kernels use as "last workgroup" tickets and put back to zero themselves
  unsigned int *tickets = nullptr;
  // pinned host memory the device READS: a small host-pointer input (a one-launch Fit's target, icp_small.hip) is copied
  // here by the host and read over the bus by the kernel that wants it -- hipMemcpyAsync from pageable memory is a copy
  // into the runtime's staging buffer, a copy command and its start-up: 25 us in front of a 160 us Fit.  One user at a
  // time: an event behind the reading kernel says when the next may write (small_upload / small_upload_read, core.hip).
  void *up = nullptr;
  hipEvent_t up_read = nullptr;
  bool up_pending = false;
};
constexpr size_t kSmallUploadBytes = 256 * 1024;
// the calling context's pinned copy of a small host input, readable by kernels enqueued on st -- or nullptr (too large, no
// such memory, the last one not read yet): the caller copies as ever.  small_upload_read: the kernels that read it are enqueued.
const void *small_upload(const void *h_src, size_t bytes);
void small_upload_read(hipStream_t st);
Context &ctx();  // the calling thread's current context (the library's outside any call)
int current_slot();  // the calling thread's device slot (pcgx_set_device; 0 unless a process drives several GPUs)
pcgx_status ensure_init();
// Scope of one ABI call: binds the thread to a context (pooled: any free one of the pool, waiting for
// one if all are busy; else the library's, exclusively) and to the library's device -- HIP's current
// device is per thread, a caller may come from any OS thread.  Nested calls keep the outer context.
struct CallScope {
  explicit CallScope(bool pooled);
  ~CallScope();
  int slot_ = -1;
};
#define PCGX_API_LOCK() ::pcgx::CallScope pcgx_call_scope__(false)
#define PCGX_API_CALL() ::pcgx::CallScope pcgx_call_scope__(true)
inline hipStream_t pick_stream(void *s) { return s ? (hipStream_t)s : ctx().stream; }
// host (pageable) -> device on `st` (the copies of the host-pointer seams, core.hip)
pcgx_status staged_upload(void *d_dst, const void *h_src, size_t bytes, hipStream_t st);
// device -> host (pageable): work enqueued on `st` before is waited for; returns when h_dst holds the data
pcgx_status staged_download(void *h_dst, const void *d_src, size_t bytes, hipStream_t st);
// Buffers go back to the block cache and may be handed out again at once: whatever any stream still
// has in flight on them must be done first (hipFree used to synchronise implicitly).
void dev_cache_quiesce();

// ---- KD-tree ---------------------------------------------------------------
// The reference's recursively sorted indice slice (kdtree.go:348-370), read left
// to right, is the in-order traversal of its tree and defines it completely: the
// node of range [lo,hi) is element lo + (hi-lo)/2, its children are [lo,mid) and
// [mid+1,hi), dim = depth%3.  Because the split is always at len/2 the SHAPE
// depends on N only, so no child pointers are stored.  Device layout (DESIGN.md
// "KD-tree"): BFS / Eytzinger order, nodes[b] = {x, y, z, bits(id)} with the root
// at b = 1 and the children of b at 2b and 2b+1 (slots of absent nodes are
// unused); depth(b) = floor(log2 b).  Ancestors of a node are plain shifts of
// its index, which is what lets a whole root-to-leaf path be fetched in
// parallel (speculative descent, knn_walk.h).
constexpr int64_t kMaxTreePoints = (int64_t)1 << 26;  // BFS index < 2^27 (frame encoding)

struct TreeView {
  const float4 *nodes;  // [2^depth] BFS slots, index 0 unused
  int32_t n;
  int32_t depth;  // node.maxDepth(0) = floor(log2 n) + 1
  // Leaf directory: a 2^g x 2^g x 2^g grid over the base cloud's bounding box; dir[cell] is
  // the BFS index of the leaf the descent of the cell's centre reaches.  Used ONLY as a
  // prediction of a query's descent path (verified against the real comparisons).
  const uint32_t *dir;
  int32_t dir_bits;  // g
  float dir_lo[3];
  float dir_scale[3];  // cells per metre (0 for a degenerate axis)
  int32_t refill_threshold;  // walk kernels: emit + refill once this many lanes of a wave wait
  int32_t tight_levels;      // walk kernels: levels a chunk preparation descends below a wrong prediction
  int32_t chunks_per_refill; // walk kernels: chunks a wave may prepare in one refill section
};

// knn_grid.h: uniform grid over the base cloud (certified fast path of Nearest)
struct GridView {
  const float4 *pts;      // [n] {x, y, z, bits(id)} in cell order
  const uint32_t *start;  // [cells + 1] first point of each cell, cell = (z * ny + y) * nx + x
  const float *cert;      // [n] by point id: a query whose DistSq to the point is below this has it as its nearest
                          // (knn_grid.hip, grid_cert_kernel); nullptr: not made (labelled trees)
  float lo[3];
  float h, inv_h;  // cell edge
  int32_t nx, ny, nz;
};

// XCD-contiguous placement of the tiles of a launch whose neighbouring tiles read neighbouring data:
// workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with an L2 of its own, so
// workgroup b takes tile (b % 8) * per + b / 8 and an XCD works through one contiguous eighth of the
// tiles -- its L2 then holds an eighth of what the launch reads instead of all of it.  Launch
// xcd_grid(ntiles) workgroups; tiles >= ntiles have nothing to do.
#if defined(__HIPCC__)
__device__ __forceinline__ uint32_t xcd_tile(uint32_t block, uint32_t ntiles) {
  const uint32_t per = (ntiles + 7u) / 8u;
  return (block & 7u) * per + (block >> 3);
}
#endif
inline unsigned xcd_grid(unsigned ntiles) { return 8u * ((ntiles + 7u) / 8u); }

// kdtree_build.cpp
void build_inorder(const float *xyz, int64_t n, int32_t *inorder_ids);
inline int32_t tree_depth(int64_t n) {
  int32_t d = 0;
  while (n > 0) { d++; n >>= 1; }
  return d;
}

// kdtree_build_gpu.hip: in-order ids (d_order) and BFS slots (d_nodes) from packed device xyz
pcgx_status build_tree_device(const float *d_xyz, int64_t n, int32_t depth, uint32_t *d_order, float4 *d_nodes,
                              const int32_t *d_labels, hipStream_t st);

}  // namespace pcgx
// knn.hip: the tree queries run on: `t` itself, or the tree over the points left after DeletePoint
// (rebuilt here if deletions happened since).  *empty: every point was deleted (root == nil).
pcgx_status resolve_tree(const pcgx_kdtree *t, const pcgx_kdtree **active, bool *empty);
// knn_explicit.hip: the reference's patched tree of a handle that has seen DeletePoint
void xtree_delete(pcgx_kdtree *t, int64_t pid);  // caller holds t->mu
void xtree_delete_batch(pcgx_kdtree *t, const int64_t *ids, int64_t m);  // ... in call order, on the host's threads where the order allows
void xtree_free(pcgx_kdtree *t);
int xtree_max_depth(const pcgx_kdtree *t);  // caller holds t->mu
// ... and the same walk on the host, for batches of a few points (knn_explicit.hip, at the end)
int64_t xtree_host_walk_max();
long long xtree_host_walks(bool reset);
void xtree_host_nearest(const pcgx_kdtree *t, const float *q, int64_t nq, float max_range, float min_dist_sq, int64_t *ids,
                        float *dist_sq);
bool xtree_host_range(const pcgx_kdtree *t, const float *q, int64_t nq, float max_range, int64_t *counts, const int64_t *offsets,
                      int64_t *ids, float *dist_sq);
pcgx_status xtree_launch_nearest(const pcgx_kdtree *t, const float *d_q, const int32_t *d_perm, int64_t nq,
                                 float max_range_sq, float min_dist_sq, int32_t *d_ids, float *d_dsq, hipStream_t st);
pcgx_status xtree_launch_range(const pcgx_kdtree *t, bool fill, const float *d_q, const int32_t *d_perm, int64_t nq,
                               float bound, int64_t *d_counts, const int64_t *d_offsets, int64_t total, int32_t *d_id,
                               uint32_t *d_key, uint32_t *d_query, hipStream_t st);
namespace pcgx {

// knn.hip
constexpr int kKnnBlock = 512;  // 8 waves (2 workgroups per CU; 256 x 4 and 1024 x 1 measured slower)
constexpr int kWalkQueueBytesPerWave = 7 * 128 * 4;  // knn_walk.h kQueueWords x kQueueSlots
// Dynamic LDS of a walk kernel block: frame stacks [(depth-1)][block] x 4 B (19 KB at 1M
// points), one prepared-query queue per wave (3.5 KB each), the top split values (256 B).
inline size_t walk_stack_bytes(const TreeView &tv, int block) {
  int levels = tv.depth > 1 ? tv.depth - 1 : 1;
  return (size_t)levels * block * sizeof(uint32_t);
}
#ifndef PCGX_WALK_TOP_LEVELS
#define PCGX_WALK_TOP_LEVELS 6
#endif
constexpr int kWalkTopLevels = PCGX_WALK_TOP_LEVELS;  // levels whose split values a walk block keeps in LDS
constexpr int kWalkTopBytes = (1 << kWalkTopLevels) * 4;
inline size_t walk_lds_bytes(const TreeView &tv, int block) {
  return walk_stack_bytes(tv, block) + (size_t)(block / 64) * kWalkQueueBytesPerWave + kWalkTopBytes;
}
int walk_blocks_per_cu(const TreeView &tv);
int walk_refill_threshold();
int walk_tight_levels();
int walk_chunks_per_refill();
int walk_oversubscribe();
pcgx_status launch_nearest(const TreeView &tv, const float *d_q, const int32_t *d_perm, int64_t nq,
                           float max_range_sq, float min_dist_sq, int32_t *d_ids, float *d_dsq,
                           hipStream_t st);
// exact-mode walk of the queries d_list[0 .. *d_count) (a device-side count <= nq_max)
pcgx_status launch_nearest_listed(const TreeView &tv, const float *d_q, const int32_t *d_list,
                                  const uint32_t *d_count, int64_t nq_max, float max_range_sq, int32_t *d_ids,
                                  float *d_dsq, hipStream_t st);

// sort.hip
size_t radix_sort_workspace_bytes(int64_t n);
// iota_vals: the values are the positions 0 .. n-1 and vals[0] need not be filled (the first pass
// takes a position for its value: a 4n-byte write and read less)
// first_hist_done: the caller's own kernel has filled the first pass's tile histograms while it made the keys
// (radix_first_hist below says where and for which tiles)
pcgx_status radix_sort_pairs(uint32_t *keys[2], uint32_t *vals[2], int64_t n, int key_bits,
                             void *workspace, int *result, hipStream_t st, bool iota_vals = false,
                             bool first_hist_done = false);
// The first pass's histograms: hist[digit * nblocks + tile], digit = key & 255, tile = 256 * items consecutive keys
struct RadixFirstHist {
  uint32_t *hist;
  int nblocks, items;
};
RadixFirstHist radix_first_hist(int64_t n, void *workspace);
// sticky_first: a NaN coordinate of the FIRST point stays (min, max := Vec3At(0), minmax.go:13-23); false for a
// later slice of a cloud whose min / max are folded over ranks
pcgx_status launch_minmax(const void *d_data, int64_t n, int32_t stride, int32_t off, float *d_out6,
                          hipStream_t st, bool sticky_first = true);
// a few result words of a call on the host: through the context's mailbox where there is one (a one-wave kernel behind
// the call's kernels, the host polls; the stream is not synchronised), else copy + wait.  bytes: a multiple of 4
constexpr size_t kTicketBytes = 8192;
constexpr size_t kMailboxBytes = 2048;
pcgx_status read_back_small(const void *d_src, size_t bytes, void *host_dst, hipStream_t st);
// ... for a kernel that stores its result words into the mailbox itself (words from mailbox[2] on, then `seq` into
// mailbox[0], system scope): the number to give it, and the host's wait for it
uint32_t mailbox_next_seq();
pcgx_status mailbox_wait(uint32_t seq, size_t bytes, void *host_dst, hipStream_t st);
// ... for a kernel that stores every result word together with `seq` as ONE 64-bit word {word, seq} from byte 8 of the
// mailbox on (no order among the stores needed, nothing to wait for on the device)
pcgx_status mailbox_wait_tagged(uint32_t seq, int words, uint32_t *host_dst, hipStream_t st);
// the same, and the six floats on the host (through the context's mailbox; the stream is not synchronised)
pcgx_status minmax_to_host(const void *d_data, int64_t n, int32_t stride, int32_t off, float *d_out6, float out6[6],
                           hipStream_t st, bool sticky_first = true);
// perm[pos] = index of the point visited at position pos (coarse Morton order over the box
// [lo, hi]).  Uses the arena.
pcgx_status morton_order(const float *d_q, int64_t n, const float lo[3], const float hi[3], int32_t *d_perm,
                         hipStream_t st);


// icp.hip / strict.hip
// Loop state kept in device memory so that a whole Fit can be enqueued without
// a host round trip per iteration (and captured in a hipGraph).
struct IcpState {
  float trans[16];        // accumulated transform (icp.go:47)
  int32_t iter;           // gradientDescentUpdater.i (updater.go:41)
  int32_t num_iteration;  // Stat.NumIteration (icp.go:50)
  int32_t done;           // converged, or failed
  int32_t status;         // PCGX_OK / PCGX_E_NOT_ENOUGH_PAIRS
  Evaluated ev;           // Stat.Evaluated (icp.go:54)
  float hessian[36];      // plane sessions: Evaluated.Hessian (evaluator.go:28), else unused
};


struct IcpKernelParams {
  float max_dist_sq;
  float min_dist_sq;
  int32_t min_pairs;
  UpdaterParams upd;
  GaussNewtonParams gn;  // plane sessions
  int32_t weight_fn;     // PCGX_WEIGHT_* (evaluator.go:130)
  float weight_a;
};


#if defined(__HIPCC__)
// "Am I the launch's last workgroup?" -- one thread per workgroup asks, when the workgroup's results are out (stores
// that the last workgroup reads: write-through and waited for, the caller's business).  Returning atomics on ONE word
// are served one after the other, 10 ns each (and so are atomics on different words of one 128-byte line): 1024
// workgroups that come within 3 us of each other waited up to 10 us for their turn.  Two steps instead: 32 ticket words
// (a line each) taken by the workgroups of the same index mod 32, and the last of a word's takers takes one of the top
// word.  `tickets`: the context's zeroed words (kTicketBytes); the words are zero again when the last workgroup has
// its answer -- one launch at a time per context may use them.
constexpr unsigned int kTicketGroups = 32u, kTicketStride = 32u;  // (words)
static_assert((1u + kTicketGroups) * kTicketStride * 4u <= kTicketBytes, "the context's ticket words");
__device__ __forceinline__ bool last_workgroup_ticket(unsigned int *tickets) {
  const unsigned int g = blockIdx.x & (kTicketGroups - 1u), groups = min(gridDim.x, kTicketGroups);
  const unsigned int members = (gridDim.x - g + kTicketGroups - 1u) / kTicketGroups;
  unsigned int *mine = tickets + kTicketStride * (1u + g);
  if (atomicAdd(mine, 1u) != members - 1u) return false;
  __hip_atomic_store(mine, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (atomicAdd(tickets, 1u) != groups - 1u) return false;
  __hip_atomic_store(tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}

// Inclusive sum over the lanes of a wave by DPP (row_shr:n inside a row of 16 lanes, row_bcast:15 / :31 from a row's last
// lane to the rows behind it): a partner's value is a register move, where __shfl_up is a trip through the LDS crossbar
// and a wait.  Call with all lanes of the wave active (a DPP read of a lane that is switched off returns the reader's own
// operand).
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
#define PCGX_DPP_ADD(CTRL, MASK) v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, MASK, 0xf, (MASK) == 0xf)
  PCGX_DPP_ADD(0x111, 0xf);
  PCGX_DPP_ADD(0x112, 0xf);
  PCGX_DPP_ADD(0x114, 0xf);
  PCGX_DPP_ADD(0x118, 0xf);
  PCGX_DPP_ADD(0x142, 0xa);
  PCGX_DPP_ADD(0x143, 0xc);
#undef PCGX_DPP_ADD
  return v;
}
// Evaluate tail (evaluator.go:92-105,156-186) + Update (updater.go:44-71) + the loop
// bookkeeping of Fit (icp.go:49-60); one thread.
__device__ __forceinline__ void icp_update_step(IcpState *__restrict__ state, const double *__restrict__ sums10,
                                                const IcpKernelParams &kp) {
  state->num_iteration += 1;
  const int64_t npairs = (int64_t)sums10[S_PAIRS];
  if (npairs < (int64_t)kp.min_pairs) {
    state->ev.num_pairs = npairs;
    state->status = PCGX_E_NOT_ENOUGH_PAIRS;
    state->done = 1;
    return;
  }
  Evaluated ev;
  finish_evaluate(sums10, ev);
  state->ev = ev;
  Mat4 t;
  for (int i = 0; i < 16; i++) t.m[i] = state->trans[i];
  int32_t it = state->iter;
  const bool converged = gradient_descent_update(kp.upd, it, ev.gradient, t);
  for (int i = 0; i < 16; i++) state->trans[i] = t.m[i];
  state->iter = it;
  if (converged) state->done = 1;
}

#endif

// strict.hip: the evaluator's sequential float32 sums (evaluator.go:122-145), bit for bit, in parallel
struct StrictBuffers;
struct StrictWork;
pcgx_status strict_create(int64_t nt, const float *tx, const float *ty, const float *tz, const uint32_t *pos_of,
                          StrictBuffers **out, hipStream_t st);
void strict_destroy(StrictBuffers *b);
// the work descriptor as this iteration's kernels take it (strict_terms.h): the correspondence kernel forms the
// tile sums on its way out when it is handed one
const StrictWork *strict_work(StrictBuffers *b, const IcpKernelParams &kp);
// have_tile_sums: the correspondence kernel formed them (else strict_tilesum_kernel runs first)
// first_iter: (as far as the host can tell) the first Evaluate of a Fit -- the repair pass runs (strict.hip)
pcgx_status strict_enqueue(StrictBuffers *b, const float4 *match, const uint32_t *pos_of, IcpState *state,
                           double *sums10, const IcpKernelParams &kp, bool fuse_update, bool have_tile_sums, bool first_iter,
                           hipStream_t st);
pcgx_status strict_read_debug(StrictBuffers *b, unsigned long long out[64], hipStream_t st);
// strict_check.hip: the same sums by one wave, term after term (the on-device cross-check; set_strict 2)
pcgx_status strict_check_enqueue(const float *d_xyz, int64_t nt, int64_t nt_pad, const float4 *match, const uint32_t *pos_of,
                                 IcpState *state, const IcpKernelParams &kp, float *d_terms, unsigned long long *d_valid,
                                 double *d_sums, bool fuse_update, hipStream_t st);
// the same sums over a target spread over the ranks of `c` (the sequential order: the ranks' tiles one after the
// other); local_failed: this rank launches nothing but still takes part in every collective, with its flag up
pcgx_status strict_enqueue_sharded(StrictBuffers *b, const float4 *match, const uint32_t *pos_of, IcpState *state,
                                   double *sums10, const IcpKernelParams &kp, pcgx_comm *c, int rank, int world,
                                   bool local_failed, hipStream_t st);
// ... the RING form of the same (strict.hip, strict_enqueue_ring; comm.hip makes the ring): every rank's inbox in
// host-coherent memory that all GPUs of the node write and poll, no collective per step
struct RingView {
  unsigned long long *words = nullptr;  // device-visible address of the host-coherent block (rank k's part at words + k * words_per_rank): the abort words
  unsigned long long *host = nullptr;   // the same memory as this process's host sees it
  unsigned long long *const *tab = nullptr;  // device memory: [world] every rank's inbox (data words) as this device addresses it
  unsigned long long *mine = nullptr;        // tab[rank]
  int32_t words_per_rank = 0, rank = 0, world = 1;
  uint32_t epoch = 0;                   // this step's tag: {the communicator's Fit number, step + 1} (comm.hip, ring_tag; every rank counts alike)
  int32_t kind = 0;                     // 1: the data words in host-coherent memory; 2: in the ranks' device memory
  long long guess_ticks = 0;            // bound of the waits only guesses depend on (strict_terms.h, kRingGuessTicks*)
};
pcgx_status strict_enqueue_ring(StrictBuffers *b, const float4 *match, const uint32_t *pos_of, IcpState *state, double *sums10,
                                const IcpKernelParams &kp, const RingView &ring, bool local_failed, bool first_iter, hipStream_t st);
void ring_abort_from_host(const RingView &ring, uint32_t reason);
pcgx_status strict_reset(StrictBuffers *b, hipStream_t st);
// icp_small.hip: a Fit's iterations in ONE persistent launch (small clouds: the tree's inner levels in LDS)
bool small_fit_eligible(const TreeView &tv, int64_t nt, bool many_ties);
size_t small_fit_sync_bytes();
size_t small_fit_terms_bytes(int64_t nt);
int small_fit_max_iters();
bool small_fit_wants_order(int64_t nt);
// d_perm: [nt] the session's order of the targets (position -> caller's index), made here -- or nullptr: the caller's order
pcgx_status small_fit_prepare(const float *d_target_aos, int64_t nt, const float box_lo[3], const float box_hi[3], int32_t *d_perm, float *d_xyz,
                               uint32_t *d_pos_of, IcpState *state, void *terms, void *sync, hipStream_t st);
pcgx_status small_fit_enqueue(const TreeView &tv, const float *tx, const float *ty, const float *tz, int64_t nt, IcpState *state,
                              const IcpKernelParams &kp, void *terms, unsigned long long *valid, double *sums10, void *sync,
                              uint32_t launch_no, int iters, const int32_t *perm, hipStream_t st, volatile uint32_t *mailbox = nullptr,
                              uint32_t mailbox_seq = 0u);
// comm.hip: the communicator's ring (made on first use, collectively; nullptr: this communicator exchanges through
// collectives only -- ranks on several nodes, no shared memory, PCGX_SHARD_RING=0), and a step's view of it
bool comm_ring_step(pcgx_comm *c, int32_t step, RingView *out);
void comm_ring_new_fit(pcgx_comm *c);
int comm_ring_kind(pcgx_comm *c);
void comm_attach_local_ring(pcgx_comm *c, unsigned long long *block, int32_t words_per_rank);
}  // namespace pcgx

struct pcgx_kdtree {
  int64_t n = 0;
  int32_t depth = 0;
  float4 *d_nodes = nullptr;       // [2^depth] BFS-ordered nodes
  uint32_t *d_dir = nullptr;       // [8^dir_bits] leaf directory
  uint32_t *d_inv = nullptr;       // [n] point id -> BFS index of its node (range.hip, made on first use, guarded by mu)
  int32_t dir_bits = 0;
  float dir_lo[3] = {0, 0, 0}, dir_scale[3] = {0, 0, 0};
  float bbox_lo[3] = {0, 0, 0}, bbox_hi[3] = {0, 0, 0};  // of the base cloud
  bool has_nan = false;            // a NaN coordinate among the points (host build; no one-launch Fit: icp_small.hip)
  bool many_ties = false;          // along some axis most coordinates come several times (a scan of a plane, a lattice): the
                                   // reference's plane test (kdtree.go:111-115) rules out little there -- icp_small.hip
  pcgx::RawVector<int32_t> inorder;  // host copy of the in-order ids
  pcgx::RawVector<float> points;   // host copy of xyz (accessor order), for Vec3At
  // KDTree.DeletePoint (kdtree.go:322-332).  The implicit layout cannot express the reference's
  // patched tree, so deletions are recorded here and the next query rebuilds a tree over the
  // remaining points whose nodes keep the ORIGINAL ids (`live`; nullptr while nothing is left).
  // A tree replaced by a later rebuild is freed at once unless ICP sessions still run on it
  // (`sessions`); those are retired and freed by a later rebuild or with the handle.
  // ... and, for Nearest / Range, the reference's own patched tree (knn_explicit.hip): host mirror
  // {id, child0, child1, dim} with node index = in-order position, patched by deleteNodeImpl's
  // rules, plus its explicit device copy.
  struct XNode {
    int32_t id, c0, c1, dim;
  };
  pcgx::RawVector<XNode> xnodes;   // (no zero fill: every node is written by the build)
  int32_t xroot = -1;
  bool x_init = false, x_dirty = false;
  float4 *d_xpts = nullptr;
  void *d_xlinks = nullptr;
  float *d_xsrc = nullptr;         // the cloud's points by id on the device (the patched tree's copy is put together there)
  std::vector<uint8_t> xsub;       // id -> the depth-6 subtree its ORIGINAL node lies in, 255: above the cut (batched deletions)
  // uniform grid of the certified-nearest fast path (knn_grid.h); grid_ok false: tree walk only
  float4 *d_gpts = nullptr;
  uint32_t *d_gstart = nullptr;
  float *d_gcert = nullptr;
  pcgx::GridView grid;
  bool grid_ok = false;
  double grid_crowding = 0.0;  // mean number of other points in a point's cell
  std::vector<uint8_t> deleted;    // [n] once the first point was deleted
  int64_t n_deleted = 0;
  bool dirty = false;              // deletions since `live` was built
  pcgx_kdtree *live = nullptr;
  std::vector<pcgx_kdtree *> retired;
  std::atomic<int> sessions{0};    // open pcgx_icp_session objects that walk THIS tree object
  std::mutex mu;
  pcgx::TreeView view() const {
    pcgx::TreeView v;
    v.nodes = d_nodes;
    v.n = (int32_t)n;
    v.depth = depth;
    v.dir = d_dir;
    v.dir_bits = dir_bits;
    for (int k = 0; k < 3; k++) { v.dir_lo[k] = dir_lo[k]; v.dir_scale[k] = dir_scale[k]; }
    v.refill_threshold = pcgx::walk_refill_threshold();
    v.tight_levels = pcgx::walk_tight_levels();
    v.chunks_per_refill = pcgx::walk_chunks_per_refill();
    return v;
  }
};
