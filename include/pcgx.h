/*
 * pcgx.h -- C ABI of libpcgx.so, the MI355X (gfx950) point-cloud hot path that
 * plugs in behind seqsense/pcgol's Go interfaces.
 *
 * The reference has NO FFI of its own (pure Go); its seams are Go interfaces.
 * Every entry point below names the reference interface / function it
 * replaces (paths relative to the reference repository root).  A cgo shim
 * (go/) implements those Go interfaces by calling these
 * symbols.  Plain pointers and sizes only; no C++/torch types.
 *
 * Conventions
 *  - every function returns a pcgx_status (0 = OK); pcgx_last_error() gives
 *    the message of the last failure on the calling thread.
 *  - "host" pointers are ordinary process memory owned by the caller (Go
 *    slices); the library copies in/out before returning and never retains
 *    them (cgo pointer rules).
 *  - "_dev" entry points take DEVICE pointers (hipMalloc / pcgx_dev_alloc /
 *    torch tensor .data_ptr()) and a HIP stream passed as void* (NULL = the
 *    library's own stream); they only enqueue work.
 *  - point clouds are the reference's AoS little-endian records
 *    (pc/pointcloud.go:64-78): record i = data + i*stride, xyz = three
 *    consecutive float32 at byte offset xyz_off (pc/pointcloud.go:130-163).
 *  - Mat4 is column-major float[16] (mat/mat4.go:8-10).
 *  - ids are indices into the accessor the tree was built from
 *    (Vec3At(id), pc/storage/search.go:8-11); -1 = not found.
 */
#ifndef PCGX_H
#define PCGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCGX_API __attribute__((visibility("default")))

typedef int32_t pcgx_status;
enum {
  PCGX_OK = 0,
  PCGX_E_NO_POINT = 1,         /* pc/minmax.go:10-12 errors.New("no point") */
  PCGX_E_NOT_ENOUGH_PAIRS = 2, /* icp/evaluator.go:16 ErrNotEnoughPairs */
  PCGX_E_BAD_FIELD = 3,        /* pc/pointcloud.go:115 "invalid field name" / bad stride, offset */
  PCGX_E_HIP = 4,              /* HIP runtime failure (no GPU, launch error, ...) */
  PCGX_E_OOM = 5,              /* host or device allocation failed */
  PCGX_E_INVALID = 6,          /* invalid argument (NULL handle, negative count, ...) */
  PCGX_E_OUT_OF_RANGE = 7,     /* voxel index outside the dense grid: the Go code panics here
                                  (pc/filter/voxelgrid/voxelgrid.go:151); we return an error */
  PCGX_E_TOO_LARGE = 8,        /* tree larger than 2^26 points (traversal frame encoding) */
  PCGX_E_NEED_GRADIENT = 9,    /* icp/icp.go:15 ErrNeedGradient (kept for the Go shim's mapping) */
  PCGX_E_SINGULAR = 10,        /* point-to-plane extension: 6x6 normal equations not positive definite */
  PCGX_E_SYNTAX = 11,          /* PCD: strconv.ErrSyntax / ErrRange from a header or ascii token (pc/io.go) */
  PCGX_E_EOF = 12,             /* PCD: io.EOF / io.ErrUnexpectedEOF */
  PCGX_E_CORRUPT = 13,         /* PCD: lzf.ErrDataCorruption / lzf.ErrInsufficientBuffer */
  PCGX_E_BAD_HEADER = 14,      /* PCD: the errors.New cases of pc/io.go:55,119,125-133,202 */
  PCGX_E_RCCL = 15             /* the exchange of the sharded ICP path failed (RCCL missing / error, callback error) */
};

/* ------------------------------------------------------------ lifecycle */

/* Selects HIP device `device` for this process (one process per GPU) and
 * creates the library stream.  Idempotent. */
PCGX_API pcgx_status pcgx_init(int32_t device);
PCGX_API pcgx_status pcgx_shutdown(void);
/* One process, several GPUs (threading row: "one process drives all 8 GPUs" -- the reference is one
 * process, icp.go:23).  The library keeps one set of streams, workspaces and call contexts per DEVICE SLOT; slot k
 * works on HIP device device_ids[k] (NULL: device k; the same device may be named several times, which is how the
 * several-GPU paths run on a one-GPU test box).  A host thread names the slot its calls are for with
 * pcgx_set_device (default 0, per thread, like HIP's current device); a handle (tree, session, communicator)
 * belongs to the slot it was made on and is used from threads that have that slot current.
 * pcgx_icp_fit_multi below needs nothing else from the caller: it starts a thread per slot itself. */
PCGX_API pcgx_status pcgx_init_devices(int32_t n, const int32_t *device_ids);
PCGX_API pcgx_status pcgx_set_device(int32_t slot);
PCGX_API pcgx_status pcgx_get_device(int32_t *slot, int32_t *hip_device);
/* Copies the last error message of the calling thread; returns its length. */
PCGX_API int32_t pcgx_last_error(char *buf, size_t cap);
PCGX_API const char *pcgx_version(void);
/* The layout version of this header's structs and fixed-size output arrays (pcgx_icp_params gained sums_mode in 3;
 * pcgx_debug_icp_strict_stats writes 64 words since 3; 4: device slots, pcgx_icp_fit_multi, pcgx_debug_voxel_stats;
 * 5: pcgx_debug_shard_stats, pcgx_prof_read_max, words 48 .. 63 of pcgx_debug_icp_strict_stats re-assigned;
 * 6: pcgx_debug_ring_kinds, pcgx_debug_host_walks, pcgx_debug_icp_one_launch).
 * A binding built against another version of the header must not call into the library: the mirrors (go/pcgx,
 * host/pcgx.hpp, pcgol_amd/_lib.py) compare PCGX_ABI_VERSION with pcgx_abi_version() when they load it.
 * pcgx_icp_params_init zeroes a parameter block of THIS version (all defaults); sizeof_params is the caller's
 * sizeof(pcgx_icp_params): a mismatch is PCGX_E_INVALID instead of a read past a shorter struct. */
#define PCGX_ABI_VERSION 6
PCGX_API int32_t pcgx_abi_version(void);
/* Block until all work enqueued on `stream` (NULL = library stream) is done. */
PCGX_API pcgx_status pcgx_sync(void *stream);
/* Threading.  Handles are immutable after build (DeletePoint excepted, as in the reference,
 * kdtree.go:322-332).  The blocking host-pointer entry points -- pcgx_kdtree_build, _nearest_batch,
 * _range_count / _range_fill, pcgx_voxel_filter, pcgx_minmax, pcgx_icp_fit / _evaluate / _pairs --
 * may be called from any number of threads at once: each call works on a stream and workspace of
 * its own (a pool of 4; a fifth caller waits) and they overlap on the GPU.  Entry points with a
 * `stream` argument (the `_dev` calls, ICP sessions) and everything else run one caller at a time
 * in the library's context, so that work given to the NULL stream stays ordered.
 * Measurement aid: out = {largest number of pooled calls in flight at once, pooled calls} since the
 * last reset. */
PCGX_API pcgx_status pcgx_debug_call_stats(int64_t out[2], int32_t reset);
/* Measurement / test aid: queries answered on the HOST since the last reset.  pcgx_kdtree_nearest_batch, _range_count
 * and _range_fill with at most 32 queries (PCGX_HOST_WALK_MAX; 0: never) -- what storage.Search.Nearest / Range for one
 * point come to (pc/storage/search.go:13-17) -- are answered by the reference-order walk (kdtree.go:83-222) on the
 * handle's host mirror of the tree, the one DeletePoint patches: no launch, no PCIe round trip; ids, DistSq bits, tie
 * winners and MinDistSq > 0 answers are those of the device path (csrc/knn_explicit.hip). */
PCGX_API pcgx_status pcgx_debug_host_walks(int64_t *queries, int32_t reset);
/* Measurement / test aid: launches of the ONE-LAUNCH Fit since the last reset (csrc/icp_small.hip: PointToPointICPGradient.Fit,
 * icp.go:23-67, for small clouds -- every iteration of the loop inside one launch).  out = {launches, of them with
 * the tree's chunks looked at only below chunks that could not be ruled out (a queue, or band by band), of them with the
 * targets grouped by place}. */
PCGX_API pcgx_status pcgx_debug_icp_one_launch(int64_t out[3], int32_t reset);
/* Measurement / test aid: which path the VoxelGrid filter calls took since the last reset.  The filter
 * (voxelgrid.go:136-187) has two device paths with identical output: the bucket path (the coordinates travel
 * with the sort keys, a workgroup per bucket of cells; csrc/voxel_bucket.hip) and the radix path (stable sort of
 * (key, index) pairs + gather; csrc/voxel.hip), which also takes what the bucket path gives up on.
 * out = {calls the bucket path answered, bucket attempts given up (a bucket or a cell too crowded),
 *        flags of the last attempt given up (1 bucket, 2 cell, 4 exchange), low key bits of the last plan}. */
PCGX_API pcgx_status pcgx_debug_voxel_stats(int64_t out[4], int32_t reset);
/* Measurement / test aid: how the sharded steps with the reference's sums were exchanged since the last reset
 * (see "The sharded ICP path" below).  out = {steps enqueued in the ring form, steps in the collective form,
 * rings made (shared memory of the node's processes, or the one process's pinned block), ring set-ups that
 * ended with the collective form (no shared memory between the ranks, PCGX_SHARD_RING=0)}. */
PCGX_API pcgx_status pcgx_debug_shard_stats(int64_t out[4], int32_t reset);
/* ... and where the rings made since the last reset keep their inboxes' data words: out = {rings whose inboxes live
 * in the ranks' DEVICE memory, mapped by their peers (hipIpcOpenMemHandle between processes, peer access between the
 * device slots of one process: a hop is one store over xGMI and a poll of local HBM), rings whose data words stay in
 * host-coherent memory (a rank could not export / map an inbox, PCGX_RING_MEM=host: a PCIe round trip per poll)}.
 * Counted once per communicator, by its rank 0 (pcgx_icp_fit_multi: by slot 0). */
PCGX_API pcgx_status pcgx_debug_ring_kinds(int64_t out[2], int32_t reset);

/* Optional in-library kernel timing (HIP events on the launch stream around
 * the named kernel class).  Used by bench.py for the live roofline figure. */
enum {
  PCGX_PROF_ICP_WALK = 0,   /* icp_corr_kernel (tree walk of the targets the grid pass left + reduce;
                               every target when the base tree has no grid) */
  PCGX_PROF_KNN_WALK = 1,   /* nearest_kernel */
  PCGX_PROF_VOXEL_ALL = 2,  /* whole voxel-filter pipeline of one call */
  PCGX_PROF_SORT_SCATTER = 3, /* rs_scatter_kernel (radix sort passes) */
  PCGX_PROF_ICP_GRID = 4,   /* icp_grid_kernel (re-projection + certified nearest + sums) */
  PCGX_PROF_KNN_GRID = 5,   /* grid_nearest_kernel */
  PCGX_PROF_STRICT_TERMS = 6, /* strict_tilesum_kernel (sessions whose correspondence kernel does not form the tile sums) */
  PCGX_PROF_STRICT_SUM = 7,   /* strict_sum_kernel (terms into LDS, parity summaries, one record per (sum, tile)) */
  PCGX_PROF_STRICT_CHAIN = 8, /* strict_chain_kernel (runs applied in order + pose update) */
  PCGX_PROF_STRICT_JOB = 9,   /* strict_job_kernel (tiles that cross a level / have no window) */
  PCGX_PROF_ICP_LEFTOVER = 10, /* icp_corr_kernel behind the grid pass (leftover walk; strict: + tile sums) */
  PCGX_PROF_KINDS = 11
};
/* on: 0 = off, 1 = every launch, n > 1 = every n-th launch of each kind (a pair of events around a
 * 30 us kernel costs several us of stream time: sampling keeps the timed run close to the untimed one). */
PCGX_API pcgx_status pcgx_prof_enable(int32_t on);
/* Resolves pending events; returns accumulated milliseconds and launch count
 * of `kind` since the last pcgx_prof_reset(). */
PCGX_API pcgx_status pcgx_prof_read(int32_t kind, double *total_ms, int64_t *launches);
/* ... and the longest single launch of `kind` since the last reset (the iteration a latency-bound kernel took longest in) */
PCGX_API pcgx_status pcgx_prof_read_max(int32_t kind, double *max_ms);
PCGX_API pcgx_status pcgx_prof_reset(void);

/* Profiling aid (not part of the drop-in surface): counters of the instrumented exact-mode walk
 * over device-resident queries: {loop iterations, active lanes summed, node fetches, emit/refill
 * sections, chunks prepared, queries verified to the leaf, first-descent levels kept, queries,
 * fetches while descending, explicit pops, passing, first-descent pops, passing, leaves,
 * iterations after the last hand-out, most iterations of a wave, 100 MHz ticks summed over waves
 * before / after the last hand-out, longest wave in ticks, waves, kernel ns, queries finished
 * inside the preparation, lane-steps / iterations of its descent loop, ticks per phase (6), -, -}. */
typedef struct pcgx_kdtree pcgx_kdtree;
PCGX_API pcgx_status pcgx_debug_walk_stats(const pcgx_kdtree *t, const float *d_q, int64_t nq, float max_range,
                                           int32_t presort, const float *d_hint_xyz, uint32_t *d_leaf_io,
                                           uint64_t stats32[32]);

/* Tuning aid (not part of the drop-in surface): out = {queries of the batch the grid pass leaves to
 * the tree walk, grid cells, 1000 x mean number of other points in a point's cell, grid in use,
 * lane-slots the scan loops ran (64 per round of 4 records per wave, idle lanes included),
 * queries per reason 1..7 (csrc/knn_grid.h), point records read, cell-bound words read}. */
PCGX_API pcgx_status pcgx_debug_grid_stats(const pcgx_kdtree *t, const float *d_q, int64_t nq, float max_range,
                                           int64_t out[14]);

/* Test aid: the points' certificates (csrc/knn_grid.hip, grid_cert_kernel), by point id, n floats into host memory.
 * A query whose DistSq to point i is below cert[i] has i as its one nearest base point; 0: no certificate (a point
 * with a twin).  PCGX_E_INVALID when the tree has none (no grid, labelled points). */
PCGX_API pcgx_status pcgx_debug_grid_cert(const pcgx_kdtree *t, float *cert, int64_t n);

/* Measurement aid: what the grid pass of the session's NEXT iteration would read, without changing
 * the session: out = {targets, targets left to the walk, point records read, cell-bound words read,
 * lane-slots its scan loops run (64 per round of 4 records per wave, idle lanes included), targets that keep last
 * iteration's partner on its certificate and are not searched for at all (the counts before it are those of searching
 * for every target)}. */
typedef struct pcgx_icp_session pcgx_icp_session;
PCGX_API pcgx_status pcgx_debug_icp_grid_stats(pcgx_icp_session *s, void *stream, int64_t out[6]);

/* Measurement aid: counters of the strict sums (set_strict 1) since the last call: out = {runs
 * applied, runs whose record did not cover the state, tiles recomputed exactly, leaves of those added
 * term by term, tile records that did not cover the state, -...}; words [48 .. 57]: states that missed their
 * tile's candidate table by log2 of the distance ([57]: other sign), [58]: plain tiles the repair pass of a Fit's first
 * Evaluate turned into jobs, [60] / [61]: ticks later chunks' walkers waited for their start state (all / the rows' last
 * chunks), [62]: walkers that gave up that wait and walked the earlier chunks alone, [63]: workgroups of the summary
 * kernel that gave up its exchange (both 0 in a healthy run: the tests require it). */
PCGX_API pcgx_status pcgx_debug_icp_strict_stats(pcgx_icp_session *s, void *stream, int64_t out[64]);

/* The strict-sum pipeline in plain host loops (no GPU): *out = the sequential float32 sum
 * 0 + t0 + t1 + ... computed the way the strict kernels compute it; stats as in csrc/strict_sum.h
 * (ss_host_model).  For tests of the arithmetic.  mode bit 0: general leaf form only; bit 1: no
 * error-prefix refinement. */
PCGX_API pcgx_status pcgx_debug_strict_sum_host(const float *terms, int64_t n, int32_t mode, float *out,
                                                int64_t stats[8]);

/* The same on the device: the strict_sum / strict_job / strict_chain kernels (csrc/strict.hip) run on terms given
 * as they are -- terms[9][n] (row-major, host memory), out[k] = 0.0f + terms[k][0] + terms[k][1] + ... in sequential
 * float32 -- instead of on the terms of a session's pairs.  For GPU tests with rows no registration produces (a tie
 * at every step, cancellation to zero, subnormals, overflow, NaN).  stats (may be NULL): the counters of
 * pcgx_debug_icp_strict_stats. */
PCGX_API pcgx_status pcgx_debug_strict_sum_dev(const float *terms, int64_t n, float out[9], int64_t stats[64]);

/* Device memory helpers for hosts that have no HIP binding of their own. */
PCGX_API pcgx_status pcgx_dev_alloc(size_t bytes, void **dptr);
PCGX_API pcgx_status pcgx_dev_free(void *dptr);
PCGX_API pcgx_status pcgx_dev_upload(void *dptr, const void *host, size_t bytes);
PCGX_API pcgx_status pcgx_dev_download(void *host, const void *dptr, size_t bytes);

/* --------------------------------------------------------------- KD-tree
 * replaces pc/storage/kdtree: kdtree.New (kdtree.go:33-56, newNode :348-370),
 * KDTree.Nearest (:83-146, searchLeafNode :199-222), MinDistSq field (:22),
 * behind storage.Search (pc/storage/search.go:13-17).
 */
typedef struct pcgx_kdtree pcgx_kdtree;

/* kdtree.New(ra): builds the median-split tree (dim = depth%3, upper median,
 * ties on the split axis ordered stably by current position) and uploads it.
 * n == 0 -> PCGX_E_NO_POINT (the Go code panics, kdtree.go:355-356). */
PCGX_API pcgx_status pcgx_kdtree_build(const void *data, int64_t n, int32_t stride,
                                       int32_t xyz_off, pcgx_kdtree **out);
PCGX_API pcgx_status pcgx_kdtree_free(pcgx_kdtree *t);
/* Len() of the accessor the tree indexes (pc/randomaccess.go:9). */
PCGX_API pcgx_status pcgx_kdtree_len(const pcgx_kdtree *t, int64_t *n);
/* node.maxDepth(0) (kdtree.go:385-395). */
PCGX_API pcgx_status pcgx_kdtree_max_depth(const pcgx_kdtree *t, int32_t *depth);
/* In-order point ids (child0, node, child1): the final state of the
 * reference's in-place sorted indice slice; defines the whole tree.  After DeletePoint: the
 * remaining ids (live_count of them) in the in-order sequence of the canonical tree over them. */
PCGX_API pcgx_status pcgx_kdtree_inorder(const pcgx_kdtree *t, int64_t *ids /* [n] host */);
/* Vec3At(id) for a batch of ids (pc/randomaccess.go:8). */
PCGX_API pcgx_status pcgx_kdtree_points(const pcgx_kdtree *t, const int64_t *ids, int64_t m,
                                        float *xyz /* [3m] host */);

/* The tree as the reference holds it (kdtree.go:25-29), pre-order: node k = {id, dim, index of
 * child0, index of child1} in the dump (-1 = nil); after DeletePoint the patched tree.  *n_nodes =
 * nodes in the tree, the first min(n_nodes, cap_nodes) quadruples are written. */
PCGX_API pcgx_status pcgx_kdtree_dump(const pcgx_kdtree *t, int64_t *out4, int64_t cap_nodes, int64_t *n_nodes);

/* KDTree.DeletePoint (kdtree.go:322-332) for a batch of ids.  An id outside [0, Len()) is
 * PCGX_E_OUT_OF_RANGE (the reference's "does not correspond to any point in the tree",
 * :323-325) and deletes nothing; deleting a point twice is a no-op (kdtree_test.go:576-650).
 * Len() / Vec3At() keep describing the accessor (all points), as in the reference.
 * The handle then keeps the reference's own patched tree (findMinimumImpl / deleteNodeImpl,
 * :224-320, applied in call order on a host mirror) and Nearest / Range walk an explicit device
 * copy of it in the reference's visit order: ids and DistSq as the Go code returns them, exact ties
 * and MinDistSq > 0 included (slower than the implicit tree: no speculative descent).  max_depth
 * and pcgx_kdtree_dump describe the patched tree.  ICP sessions created on such a handle walk the
 * patched tree as well (and see later deletions); a session created before the first deletion keeps
 * the tree it was created on.  Region growing uses a canonical tree rebuilt over the remaining points
 * (Range hits are a set: same components). */
PCGX_API pcgx_status pcgx_kdtree_delete_points(pcgx_kdtree *t, const int64_t *ids, int64_t m);
/* Points still in the tree (Len() minus deleted). */
PCGX_API pcgx_status pcgx_kdtree_live_count(const pcgx_kdtree *t, int64_t *n_live);

/* Batched KDTree.Nearest: for each query i the exact result of
 * k.Nearest(q[i], max_range) with k.MinDistSq = min_dist_sq, i.e.
 * {ID, DistSq} or {-1, max_range^2} (kdtree.go:84-86,100-103).
 * q is packed xyz float32 [3*nq]. */
PCGX_API pcgx_status pcgx_kdtree_nearest_batch(const pcgx_kdtree *t, const float *q, int64_t nq,
                                               float max_range, float min_dist_sq,
                                               int64_t *ids /* [nq] */, float *dist_sq /* [nq] */);

/* Batched KDTree.Range (kdtree.go:148-197): all points with DistSq < max_range^2, per query
 * sorted by DistSq (ties: discovery order of the reference walk; Go's sort leaves them
 * unspecified).  The result length is data dependent, hence two calls:
 *   count: counts[i] = number of neighbours of q[i];
 *   fill:  offsets[0..nq] = exclusive prefix sum of the counts (offsets[nq] = total); query i's
 *          neighbours are written to ids / dist_sq [offsets[i], offsets[i+1]). */
PCGX_API pcgx_status pcgx_kdtree_range_count(const pcgx_kdtree *t, const float *q, int64_t nq,
                                             float max_range, int64_t *counts /* [nq] */);
PCGX_API pcgx_status pcgx_kdtree_range_fill(const pcgx_kdtree *t, const float *q, int64_t nq,
                                            float max_range, const int64_t *offsets /* [nq+1] */,
                                            int64_t *ids, float *dist_sq);

#define PCGX_KNN_PRESORT 1u /* Morton-order the queries inside the call (results are
                               returned in the caller's order either way) */
/* Same, device resident: d_q packed xyz [3*nq], d_ids int32 [nq], d_dist_sq [nq]. */
PCGX_API pcgx_status pcgx_kdtree_nearest_batch_dev(const pcgx_kdtree *t, const float *d_q,
                                                   int64_t nq, float max_range,
                                                   float min_dist_sq, uint32_t flags,
                                                   int32_t *d_ids, float *d_dist_sq,
                                                   void *stream);

/* ------------------------------------------------------------- VoxelGrid
 * replaces pc/filter/voxelgrid: voxelgrid.New(leaf, WithChunkSize(chunk))
 * .Filter(pp) (voxelgrid.go:23-187, option.go:14-18) behind filter.Filter
 * (pc/filter/filter.go:7-9), incl. pc.MinMaxVec3 (pc/minmax.go:9-26).
 */

/* pc.MinMaxVec3 over an AoS cloud. n == 0 -> PCGX_E_NO_POINT. */
PCGX_API pcgx_status pcgx_minmax(const void *data, int64_t n, int32_t stride, int32_t xyz_off,
                                 float vmin[3], float vmax[3]);

/* Filter: out_data must hold n*stride bytes (worst case); *out_n = number of
 * output records (Width of the returned cloud, Height = 1).  chunk = {0,0,0}
 * (any product == 0) selects the non-chunked path (voxelgrid.go:45-47). */
PCGX_API pcgx_status pcgx_voxel_filter(const void *data, int64_t n, int32_t stride,
                                       int32_t xyz_off, const float leaf[3],
                                       const int32_t chunk[3], void *out_data, int64_t *out_n);
/* Device resident: d_data/d_out are device buffers (d_out >= n*stride bytes);
 * *out_n is known when the call returns (it synchronises internally for the grid set-up and
 * the count), but the kernels that fill d_out may still be running on `stream`. */
PCGX_API pcgx_status pcgx_voxel_filter_dev(const void *d_data, int64_t n, int32_t stride,
                                           int32_t xyz_off, const float leaf[3],
                                           const int32_t chunk[3], void *d_out,
                                           int64_t *out_n, void *stream);

/* ------------------------------------------------------------------- ICP
 * replaces pc/registration/icp: NearestPointCorresponder.Pairs
 * (correspondence.go:22-37), PointToPointEvaluator.Evaluate (evaluator.go:91-189,
 * default weight w == 1), gradientDescentUpdater.Update (updater.go:44-71,
 * rodrigues.go:11-33) and PointToPointICPGradient.Fit (icp.go:23-67).
 */

/* icp.Evaluated (evaluator.go:25-30); Hessian is never written by the
 * reference (HasHessian() == false, :76) and is omitted. */
typedef struct {
  float value;
  float gradient[6];
  float dist_rms;
  int64_t num_pairs;
} pcgx_icp_evaluated;

/* NearestPointCorresponder{MaxDist}, PointToPointEvaluator{MinPairs},
 * KDTree.MinDistSq, GradientDescentUpdaterFactory{Weight,Threshold,MaxIteration}
 * (zero values select the reference defaults 6 / 0.3 / 0.01 / 20,
 * evaluator.go:92-95, updater.go:15-37). */
typedef struct {
  float max_dist;
  float min_dist_sq;
  int32_t min_pairs;
  float weight[6];
  float threshold[6];
  int32_t max_iteration;
  /* PointToPointEvaluator.WeightFn (evaluator.go:19-23,72,110-113,130): the reference takes any Go
   * closure w = WeightFn(distSq); arbitrary host code cannot run on the device, so the weight is
   * one of the built-in forms below, each evaluated in float32 exactly as the Go expression next to
   * it (the Go shim hands out the matching closure, go/pcgx: WeightFn.Func()).  0 = the reference's
   * default.  Not offered for the point-to-plane extension. */
  int32_t weight_fn;
  float weight_fn_param; /* a */
  /* How the evaluator's nine sums (evaluator.go:122-145) are formed: PCGX_SUMS_* below.  0 = the
   * reference's own sums wherever they are defined (one GPU: bit-identical Evaluated and pose). */
  int32_t sums_mode;
} pcgx_icp_params;

PCGX_API pcgx_status pcgx_icp_params_init(pcgx_icp_params *p, size_t sizeof_params);

/* pcgx_icp_params.sums_mode.  The reference adds float32 terms pair after pair in target order; its
 * result carries that chain's rounding (~1.6e-5 on the final transform at 1M pairs), so only a sum
 * formed the same way meets "within 1e-5 of the Go code" at every size.
 *  PCGX_SUMS_REFERENCE  (0, default) the reference's sequential float32 additions, evaluated exactly by
 *                       the whole GPU (csrc/strict_sum.h) -- pcgx_icp_fit / _evaluate / sessions on one
 *                       GPU, and sharded sessions (pcgx_icp_session_step_sharded, pcgx_icp_fit_sharded,
 *                       pcgx_icp_fit_multi: the order is the ranks' tiles one after the other).  The
 *                       point-to-plane extension has no reference sums to reproduce: float64 sums.
 *  PCGX_SUMS_F64_TREE   fixed-order float64 reduction of the same float32 terms: more accurate than the
 *                       reference, equal to it up to ITS rounding noise; what a sharded sum computes.
 *  PCGX_SUMS_REFERENCE_CHAIN  the reference's additions by ONE wave, term after term (milliseconds per
 *                       1M pairs): the on-device cross-check of PCGX_SUMS_REFERENCE. */
enum {
  PCGX_SUMS_REFERENCE = 0,
  PCGX_SUMS_F64_TREE = 1,
  PCGX_SUMS_REFERENCE_CHAIN = 2,
  PCGX_SUMS_KINDS = 3
};

enum {
  PCGX_WEIGHT_ONE = 0,      /* DefaultEvaluateWeightFn: return 1 */
  PCGX_WEIGHT_CONSTANT = 1, /* return a */
  PCGX_WEIGHT_INVERSE = 2,  /* return 1 / (a + d)                                   (Cauchy-like) */
  PCGX_WEIGHT_HUBER = 3,    /* if d <= a { return 1 }; return float32(math.Sqrt(float64(a / d)))   (k^2 = a) */
  PCGX_WEIGHT_TUKEY = 4,    /* if !(d < a) { return 0 }; u := 1 - d/a; return u * u               (c^2 = a) */
  PCGX_WEIGHT_KINDS = 5
};

/* icp.Stat (stat.go:3-6) */
typedef struct {
  pcgx_icp_evaluated evaluated;
  int32_t num_iteration;
} pcgx_icp_stat;

/* Pairs(): order-preserving compaction of matched targets.  Output arrays
 * hold nt entries (worst case, correspondence.go:24). */
PCGX_API pcgx_status pcgx_icp_pairs(const pcgx_kdtree *base, const float *target, int64_t nt,
                                    float max_dist, float min_dist_sq, int64_t *base_id,
                                    int64_t *target_id, float *dist_sq, int64_t *npairs);

/* Evaluate(base, target): fused correspondence + reduction on the GPU.
 * Fewer than min_pairs (0 -> 6) pairs -> PCGX_E_NOT_ENOUGH_PAIRS. */
PCGX_API pcgx_status pcgx_icp_evaluate(const pcgx_kdtree *base, const float *target, int64_t nt,
                                       float max_dist, float min_dist_sq, int32_t min_pairs,
                                       pcgx_icp_evaluated *out);

/* Host-only pieces (no GPU needed), exposed so a host can run the reference's
 * loop around its own exchange step:
 *  sums10 = {sum w*d2, G0..G5 sums, sum w*|pt|^2, sum w, pair count} (f64);
 *  finish = evaluator.go:156-186 (normalise, sqrt, rotation limiter). */
PCGX_API pcgx_status pcgx_icp_finish_evaluate(const double sums10[10], int32_t min_pairs,
                                              pcgx_icp_evaluated *out);
/* Update(trans, ev): *iter is the updater's iteration counter u.i;
 * *converged receives the bool result (updater.go:44-71). */
PCGX_API pcgx_status pcgx_icp_update(const pcgx_icp_params *p, int32_t *iter,
                                     const float gradient[6], float trans16[16],
                                     int32_t *converged);
/* rodriguesToRotation (rodrigues.go:11-33) and Mat4 helpers used by Fit. */
PCGX_API pcgx_status pcgx_rodrigues(const float v[3], float out16[16]);
PCGX_API pcgx_status pcgx_mat4_mul(const float m[16], const float a[16], float out16[16]);
PCGX_API pcgx_status pcgx_mat4_transform(const float m[16], const float *xyz, int64_t n,
                                         float *out_xyz);

/* Evaluate with every parameter of pcgx_icp_params that concerns it (max_dist, min_dist_sq, min_pairs,
 * weight_fn, weight_fn_param). */
PCGX_API pcgx_status pcgx_icp_evaluate_params(const pcgx_kdtree *base, const float *target, int64_t nt,
                                              const pcgx_icp_params *params, pcgx_icp_evaluated *out);

/* Fit(base, target): the whole loop stays on the device (evaluate + update
 * kernels, one download at the end).  On PCGX_E_NOT_ENOUGH_PAIRS trans16 and
 * stat->num_iteration hold the state at failure, like icp.go:49-53. */
PCGX_API pcgx_status pcgx_icp_fit(const pcgx_kdtree *base, const float *target, int64_t nt,
                                  const pcgx_icp_params *params, float trans16[16],
                                  pcgx_icp_stat *stat);

/* Device-resident ICP session: the same loop, cut at the per-iteration
 * exchange so that N processes (one per GPU, each with a replica of the base
 * tree and its own tile of the target) can all-reduce the 10 partial sums
 * between `partials` and `update`.  d_sums10 is a device
 * buffer of 10 doubles owned by the caller (e.g. a torch tensor handed to
 * torch.distributed.all_reduce == RCCL). */
typedef struct pcgx_icp_session pcgx_icp_session;
PCGX_API pcgx_status pcgx_icp_session_create(const pcgx_kdtree *base, const float *target,
                                             int64_t nt, int32_t target_on_device,
                                             const pcgx_icp_params *params, double *d_sums10,
                                             pcgx_icp_session **out);
PCGX_API pcgx_status pcgx_icp_session_free(pcgx_icp_session *s);
/* Restart the loop on the same target: trans = identity, counters cleared
 * (a new Fit, icp.go:46-47). */
PCGX_API pcgx_status pcgx_icp_session_reset(pcgx_icp_session *s, void *stream);
/* Overwrite the loop state's transform and updater counter (host-driven loops:
 * evaluate this rank's partial sums at a given pose).  iter == 0 means "no
 * re-projection yet" (the first Evaluate of Fit sees the raw target, icp.go:27-30). */
PCGX_API pcgx_status pcgx_icp_session_set_pose(pcgx_icp_session *s, const float trans16[16],
                                               int32_t iter, void *stream);
/* Copy the session's d_sums10 to the host (synchronises the stream). */
PCGX_API pcgx_status pcgx_icp_session_read_sums(pcgx_icp_session *s, double sums10[10], void *stream);
/* Enqueue transform(original target, current trans) + nearest + reduction of
 * this rank's tile into d_sums10.  No-op once the session has converged. */
PCGX_API pcgx_status pcgx_icp_session_partials(pcgx_icp_session *s, void *stream);
/* Enqueue evaluate-tail + Update from the (all-reduced) d_sums10 on the device. */
PCGX_API pcgx_status pcgx_icp_session_update(pcgx_icp_session *s, void *stream);
/* partials + update back to back, for a single GPU (no exchange in between): fewer launches. */
PCGX_API pcgx_status pcgx_icp_session_step(pcgx_icp_session *s, void *stream);

/* ---- the exchange of the sharded path -------------------------------------------
 * One process per GPU; every rank holds a replica of the base tree and one spatial tile of the
 * target.  A communicator is made from an id that rank 0 generates (pcgx_comm_unique_id) and the
 * host hands to every rank over any channel it has (a file, a socket, MPI, torch's store);
 * pcgx_comm_init is collective.  RCCL (xGMI inside a node) is bound at run time: a process that
 * never shards needs none.  pcgx_comm_init_callback is the same exchange through a host function
 * that sums `count` float64 in place over the ranks (the sums then make a round trip through host
 * memory): for hosts with a transport of their own, and for tests that run several ranks on one GPU.
 * pcgx_icp_session_step_sharded is one iteration on every rank, pcgx_icp_fit_sharded the whole Fit (icp.go:23-67) on
 * this rank's tile: every rank returns the same transform.  The sums (pcgx_icp_params.sums_mode):
 *  PCGX_SUMS_REFERENCE (default)  the reference's sequential float32 additions (evaluator.go:122-145) over the ranks'
 *      tiles ONE AFTER THE OTHER, rank 0's first: the sharded Fit returns what the reference's Fit returns on that
 *      concatenated target, bit for bit.  Correspondence, summaries and jobs run on all ranks at once; the ranks before
 *      a rank hand it two float64 totals per sum and the states their walk ended in (the walk is one dependent
 *      chain: it goes round the ranks).  Where the ranks can share host memory -- the processes of one node (a POSIX
 *      shared-memory segment every rank maps and registers with HIP, agreed on through the communicator's own
 *      all-reduce on first use), or the device slots of one process -- that is the RING form: no collective per
 *      iteration; every rank owns an inbox of tagged 64-bit words that the other GPUs' kernels write and its own
 *      kernels poll -- in its OWN GPU's memory, mapped by the peers (hipIpcGetMemHandle / hipIpcOpenMemHandle between
 *      processes, hipDeviceEnablePeerAccess between the slots of one: a store over xGMI, a poll of local HBM; the
 *      handles ride on the same set-up all-reduce; the mappings are tried out with a round of tagged words before a
 *      Fit depends on them), or in the host-coherent block where a rank cannot export or map one or the trial fails
 *      (pcgx_debug_ring_kinds says which); the abort words stay in the host block, hosts write them.  Every rank's
 *      kernels are resident at once and only the walkers wait, each for
 *      one word from the rank before it (csrc/strict.hip, strict_enqueue_ring).  Elsewhere (ranks on several nodes,
 *      PCGX_SHARD_RING=0): 2 + world collectives of <= 16 x world doubles per iteration.  Same bits either way.
 *  PCGX_SUMS_F64_TREE  partials -> ONE all-reduce of the 10 (plane: 30) float64 sums -> update: faster, and off the
 *      reference by the reference's own rounding noise (1.6e-5 on the transform at 1M pairs).
 * Every collective also carries the ranks' error flag: a rank whose step fails keeps calling the collectives with its
 * flag up, and all ranks end the Fit in that same iteration (PCGX_E_RCCL on the others) -- none is left inside an
 * all-reduce.  Callers that drive the exchange themselves (pcgx_icp_session_partials -> their own all-reduce ->
 * pcgx_icp_session_update) MUST create the session with PCGX_SUMS_F64_TREE: sums of float32 chains cannot be added
 * across ranks.
 * In the ring form a failing rank raises an abort word in every inbox instead; a wait for a state that never comes is
 * bounded (10 s) and raises it too.
 * One process, several GPUs: pcgx_icp_fit_multi (a host thread per device slot, pcgx_init_devices; the ring in pinned
 * host memory, or the exchange in host memory, sums in rank order) -- the Go shim's FitMulti needs no second
 * process.  (Kernels that wait for one another must not queue up behind each other in one hardware queue: where slots
 * share ONE HIP device -- a test box -- pcgx_init_devices gives every slot's stream a hardware queue of its own.) */
typedef struct pcgx_comm pcgx_comm;
typedef struct { char internal[128]; } pcgx_comm_id;   /* == ncclUniqueId */
typedef int32_t (*pcgx_allreduce_fn)(double *host_buf, int32_t count, void *user);
PCGX_API pcgx_status pcgx_comm_unique_id(pcgx_comm_id *id);
PCGX_API pcgx_status pcgx_comm_init(int32_t rank, int32_t world, const pcgx_comm_id *id, pcgx_comm **out);
PCGX_API pcgx_status pcgx_comm_init_callback(int32_t rank, int32_t world, pcgx_allreduce_fn fn, void *user,
                                             pcgx_comm **out);
PCGX_API pcgx_status pcgx_comm_free(pcgx_comm *c);
PCGX_API pcgx_status pcgx_comm_rank(const pcgx_comm *c, int32_t *rank, int32_t *world);
PCGX_API pcgx_status pcgx_comm_allreduce_f64(pcgx_comm *c, double *d_buf, int32_t count, void *stream);
PCGX_API pcgx_status pcgx_icp_session_step_sharded(pcgx_icp_session *s, pcgx_comm *c, void *stream);
PCGX_API pcgx_status pcgx_icp_fit_sharded(const pcgx_kdtree *base, const float *tile, int64_t nt,
                                          const pcgx_icp_params *params, pcgx_comm *c, float trans16[16],
                                          pcgx_icp_stat *stat);
/* bases[r]: the tree replica built with slot r current; tiles[r] / nt[r]: slot r's part of the target (host memory). */
PCGX_API pcgx_status pcgx_icp_fit_multi(int32_t n, const pcgx_kdtree *const *bases, const float *const *tiles,
                                        const int64_t *nt, const pcgx_icp_params *params, float trans16[16],
                                        pcgx_icp_stat *stat);
/* all-reduce of a few doubles in host memory through `c` (set-up exchanges of the sharded paths) */
PCGX_API pcgx_status pcgx_comm_allreduce_host_f64(pcgx_comm *c, double *h_buf, int32_t count);
/* The voxel filter over several GPUs (second half).  Every rank holds the same cloud
 * in device memory.  Each runs the min/max pass (pc/minmax.go:9-26) over its n / world slice; the six
 * floats are exchanged through `c` (one all-reduce of 7 x world float64) and folded in rank order with
 * the reference's comparisons.  Each rank then keeps the points whose place in the reference's output
 * order -- chunk id, then cell (voxelgrid.go:49-116,137-151) -- lies in its contiguous share of that
 * key range and filters them: *out_n records in d_out (>= n * stride bytes).  The ranks' outputs,
 * rank 0's first, ARE the output of pcgx_voxel_filter_dev on one GPU, byte for byte; putting them
 * together is the caller's (a variable-length gather, or one copy per rank into the host cloud).
 * Collective: every rank of `c` must call it with the same cloud and options.  Errors as
 * pcgx_voxel_filter_dev on every rank alike. */
PCGX_API pcgx_status pcgx_voxel_filter_sharded_dev(pcgx_comm *c, const void *d_data, int64_t n, int32_t stride,
                                                   int32_t xyz_off, const float leaf[3], const int32_t chunk[3],
                                                   void *d_out, int64_t *out_n, void *stream);
/* The same with host buffers (upload, this rank's share, download of its *out_n records). */
PCGX_API pcgx_status pcgx_voxel_filter_sharded(pcgx_comm *c, const void *data, int64_t n, int32_t stride,
                                               int32_t xyz_off, const float leaf[3], const int32_t chunk[3],
                                               void *out_data, int64_t *out_n);
/* Change a session's sums after its creation (pcgx_icp_params.sums_mode sets them at creation; the
 * default is the reference's).  With strict on, the sums are the reference's: float32 additions in
 * target order, every rounding included, so Evaluated and the resulting pose are bit-identical to
 * the Go code's at any size.
 * on = 1: PCGX_SUMS_REFERENCE -- evaluated by the whole GPU (csrc/strict_sum.h: the additions of a
 *         stretch act on the state as a translation of its mantissa that is proven per rounding
 *         class, stretches are composed, one wave applies them; exact by construction);
 * on = 2: PCGX_SUMS_REFERENCE_CHAIN -- one wave adds the terms one after the other (~7 ms per 1M pairs);
 * on = 0: PCGX_SUMS_F64_TREE -- the float64 reduction.
 * Single-GPU sessions only (a sharded sum has no sequential order: step_sharded with world > 1
 * refuses a session whose strict sums were asked for explicitly and runs a default one with float64
 * sums).  Environment PCGX_ICP_STRICT=0 / 1 / 2 overrides sums_mode for every new session
 * (experiments). */
PCGX_API pcgx_status pcgx_icp_session_set_strict(pcgx_icp_session *s, int32_t on);
/* Synchronise and read back trans / stat / converged flag.  Returns
 * PCGX_E_NOT_ENOUGH_PAIRS if an iteration failed. */
PCGX_API pcgx_status pcgx_icp_session_result(pcgx_icp_session *s, void *stream,
                                             float trans16[16], pcgx_icp_stat *stat,
                                             int32_t *converged);

/* ------------------------------------------------ bucket voxel grid + segmentation
 * replaces pc/storage/voxelgrid.VoxelGrid (voxelgrid.go:7-122: dense [][]int buckets addressed by
 * int(pos*resolutionInv + 0.5)), pc/segmentation/voxelgrid.VoxelGrid.Segment (26-neighbour flood
 * fill, segmentation/voxelgrid/voxelgrid.go:39-73) and pc/segmentation/regiongrowing
 * .RegionGrowing.Segment (regiongrowing.go:23-56).  The device computes the buckets with one
 * stable sort and the connected components of the whole grid / cloud with a union-find; a seed
 * query is then a lookup (ids ascending by voxel address / point id).  The *_bfs variants return
 * the same set in the reference's own discovery order (its FIFO search replayed over the
 * device-built buckets / device Range batches); the reference's tests sort before comparing. */
typedef struct pcgx_bucket_grid pcgx_bucket_grid;
/* New(resolution, size, origin) followed by Add(point i, i) for every record of the cloud
 * (voxelgrid.go:15-23,37-45); points outside the grid are not added. */
PCGX_API pcgx_status pcgx_bucket_grid_build(const void *data, int64_t n, int32_t stride, int32_t xyz_off,
                                            float resolution, const int64_t size[3], const float origin[3],
                                            pcgx_bucket_grid **out);
PCGX_API pcgx_status pcgx_bucket_grid_free(pcgx_bucket_grid *g);
/* Len() (voxelgrid.go:110-112), points accepted by Add, occupied voxels; any pointer may be NULL */
PCGX_API pcgx_status pcgx_bucket_grid_counts(const pcgx_bucket_grid *g, int64_t *len, int64_t *n_added,
                                             int64_t *n_occupied);
/* Addr(p) (voxelgrid.go:64-79): *ok = 0 outside the grid */
PCGX_API pcgx_status pcgx_bucket_grid_addr(const pcgx_bucket_grid *g, const float p[3], int64_t *addr, int32_t *ok);
/* voxel address of every offered point, -1 where Add returned false */
PCGX_API pcgx_status pcgx_bucket_grid_point_addrs(const pcgx_bucket_grid *g, int64_t *addrs /* [n] */);
/* GetByAddr / Get (voxelgrid.go:52-62): *count = bucket length (Get: -1 = nil, p outside the grid);
 * the first min(count, cap) ids are written in insertion order.  An address outside
 * [0, Len()) is PCGX_E_OUT_OF_RANGE (the reference panics). */
PCGX_API pcgx_status pcgx_bucket_grid_get_by_addr(const pcgx_bucket_grid *g, int64_t addr, int64_t *out, int64_t cap,
                                                  int64_t *count);
PCGX_API pcgx_status pcgx_bucket_grid_get(const pcgx_bucket_grid *g, const float p[3], int64_t *out, int64_t cap,
                                          int64_t *count);
/* Indice() (voxelgrid.go:114-120): out holds n_added ids */
PCGX_API pcgx_status pcgx_bucket_grid_indice(const pcgx_bucket_grid *g, int64_t *out);
/* Segment(p) for every seed at once: point_comp[i] = smallest voxel address of the 26-connected
 * set of occupied voxels point i's voxel belongs to, -1 for points outside the grid. */
PCGX_API pcgx_status pcgx_bucket_grid_components(pcgx_bucket_grid *g, int64_t *point_comp /* [n] */);
/* Segment(p) (segmentation/voxelgrid/voxelgrid.go:39-73): *count = result length, the first
 * min(count, cap) ids are written; empty when p is outside the grid or its voxel is empty. */
PCGX_API pcgx_status pcgx_bucket_grid_segment(pcgx_bucket_grid *g, const float p[3], int64_t *out, int64_t cap,
                                              int64_t *count);

/* Segment(p) with the ids in the reference's own (FIFO flood-fill) order: the Go algorithm run on the
 * host over the device-built buckets.  Same set as pcgx_bucket_grid_segment. */
PCGX_API pcgx_status pcgx_bucket_grid_segment_bfs(pcgx_bucket_grid *g, const float p[3], int64_t *out, int64_t cap,
                                                  int64_t *count);

/* RegionGrowing (regiongrowing.go:18-56).  labels = the property accessor (Uint32At(id), id in
 * [0, Len())).  _components answers every seed at once for one max_range: comp[i] = smallest id of
 * the points reachable from i through steps with DistSq < max_range^2 between points of i's
 * property value.  _segment is Segment(p, maxRange) given those components. */
PCGX_API pcgx_status pcgx_region_growing_components(const pcgx_kdtree *t, const uint32_t *labels, float max_range,
                                                    int64_t *comp /* [Len()] */);
PCGX_API pcgx_status pcgx_region_growing_segment(const pcgx_kdtree *t, const uint32_t *labels, const int64_t *comp,
                                                 const float p[3], float max_range, int64_t *out, int64_t cap,
                                                 int64_t *count);

/* Segment(p, maxRange) with the ids in the reference's own (FIFO) order: the Range() calls of one
 * BFS level are one device batch, the queue logic runs on the host as in regiongrowing.go:33-54.
 * Same set as pcgx_region_growing_segment; needs no components. */
PCGX_API pcgx_status pcgx_region_growing_segment_bfs(const pcgx_kdtree *t, const uint32_t *labels, const float p[3],
                                                     float max_range, int64_t *out, int64_t cap, int64_t *count);

/* ---------------------------------------------------------------- PCD files
 * replaces pc.UnmarshalHeader / pc.Unmarshal / pc.Marshal (pc/io.go:24-45,47-230,232-285) with a
 * device-resident output: the records of an ascii / binary / binary_compressed file land in HBM
 * (one upload; compressed files: LZF decode on the host, the SoA -> AoS de-interleave on the
 * device) ready for the *_dev entry points.  The reference's de-interleave quirk is kept (COUNT > 1
 * fields of compressed files: only element 0 is filled, io.go:217-226). */
#define PCGX_PCD_MAX_FIELDS 64
enum { PCGX_PCD_ASCII = 0, PCGX_PCD_BINARY = 1, PCGX_PCD_BINARY_COMPRESSED = 2 }; /* pc.Format, io.go:16-22 */
typedef struct {            /* pc.PointCloudHeader (pc/pointcloud.go:9-18) + what Unmarshal derives */
  float version;
  int32_t n_fields;
  char fields[PCGX_PCD_MAX_FIELDS][32]; /* NUL-terminated names */
  int32_t size[PCGX_PCD_MAX_FIELDS];
  char type[PCGX_PCD_MAX_FIELDS];       /* 'F', 'U', 'I' */
  int32_t count[PCGX_PCD_MAX_FIELDS];
  int64_t width, height;
  int32_t n_viewpoint;
  float viewpoint[16];
  int64_t points;                       /* POINTS */
  int32_t format;                       /* PCGX_PCD_* */
  int64_t stride;                       /* sum size*count (pointcloud.go:64-70) */
  int64_t data_offset;                  /* byte offset of the payload in the file */
} pcgx_pcd_header;
PCGX_API pcgx_status pcgx_pcd_unmarshal_header(const void *file, size_t len, pcgx_pcd_header *h);
/* out_data: host buffer of h->points * h->stride bytes */
PCGX_API pcgx_status pcgx_pcd_unmarshal(const void *file, size_t len, const pcgx_pcd_header *h, void *out_data);
/* d_out: DEVICE buffer of h->points * h->stride bytes; returns when the records are in HBM */
PCGX_API pcgx_status pcgx_pcd_unmarshal_dev(const void *file, size_t len, const pcgx_pcd_header *h, void *d_out,
                                            void *stream);
/* Marshal (always "DATA binary"): *out_len = file size; out == NULL only sizes */
PCGX_API pcgx_status pcgx_pcd_marshal(const pcgx_pcd_header *h, const void *data, void *out, size_t cap,
                                      size_t *out_len);

/* ------------------------------------------- point-to-plane ICP (extension)
 * NOT in the reference: pcgol declares only the slots -- Evaluated.Hessian mat.Mat6
 * (evaluator.go:28), Evaluator.HasHessian (evaluator.go:35,76), mat.Mat6 (mat/mat6.go:3),
 * UpdaterGradient (updater.go:11-13).  This fills them: an evaluator whose residual is the
 * point-to-plane distance r = n . (pt - pb) with the 6x6 Gauss-Newton normal equations
 * (J = {n, pt x n}, parameters {t, w} ordered like Evaluated.Gradient), and a Gauss-Newton
 * updater with the reference updater's flat test, pose composition
 * trans = Translate(d0..2) * (Rodrigues(d3..5) * trans) and iteration cap.
 * No reference parity exists; tests check against the CPU oracle's float64 restatement.
 *
 * base_normals: packed xyz float32 per base point, in the tree's id order (unit length).
 * A plane session is a pcgx_icp_session whose exchange vector has 30 doubles:
 *   {sum r^2, sum J r [6], upper triangle of sum J J^T row-major [21], sum w, pair count}.
 * partials / update / step / result / reset / set_pose / free are the session calls above;
 * params->weight is unused, params->threshold / max_iteration / min_pairs / max_dist as above. */
PCGX_API pcgx_status pcgx_icp_plane_session_create(const pcgx_kdtree *base, const float *base_normals,
                                                   const float *target, int64_t nt, int32_t on_device,
                                                   const pcgx_icp_params *params, float damping,
                                                   double *d_sums30, pcgx_icp_session **out);
/* 10 or 30: length of the session's exchange vector. */
PCGX_API pcgx_status pcgx_icp_session_sums_count(const pcgx_icp_session *s, int32_t *count);
PCGX_API pcgx_status pcgx_icp_session_read_sums_n(pcgx_icp_session *s, double *sums, int32_t cap, void *stream);
/* Evaluated.Hessian of the last evaluation (2/sum(w) * sum J J^T, symmetric 6x6). */
PCGX_API pcgx_status pcgx_icp_session_hessian(pcgx_icp_session *s, void *stream, float hessian36[36]);
/* Whole Fit on the device; hessian36 may be NULL.  PCGX_E_SINGULAR if the normal equations
 * are not positive definite (e.g. all normals parallel). */
PCGX_API pcgx_status pcgx_icp_plane_fit(const pcgx_kdtree *base, const float *base_normals, const float *target,
                                        int64_t nt, const pcgx_icp_params *params, float damping,
                                        float trans16[16], pcgx_icp_stat *stat, float hessian36[36]);
/* Host-only pieces for a host-driven exchange loop. */
PCGX_API pcgx_status pcgx_icp_plane_finish_evaluate(const double sums30[30], int32_t min_pairs,
                                                    pcgx_icp_evaluated *out, float hessian36[36]);
PCGX_API pcgx_status pcgx_icp_gauss_newton_update(const pcgx_icp_params *p, float damping, int32_t *iter,
                                                  const float gradient[6], const float hessian36[36],
                                                  float trans16[16], int32_t *converged);

#ifdef __cplusplus
}
#endif
#endif /* PCGX_H */
