// pcgx.hpp -- header-only C++ mirror of the reference's Go interfaces on the
// hot path, over the C ABI (include/pcgx.h).  The reference is compiled code
// (Go) whose toolchain is absent from the build image, so the host side above
// the C ABI is mirrored in C++ (and in Python, pcgol_amd/*.py, for the tests):
// same names, argument meaning and error behaviour.
//
//   pcgx::KDTree            <- pc/storage/kdtree.KDTree      (storage.Search)
//   pcgx::VoxelGrid         <- pc/filter/voxelgrid.New(...)  (filter.Filter)
//   pcgx::PointToPointICP   <- icp.PointToPointICPGradient{Evaluator, UpdaterFactory}
//   pcgx::BucketVoxelGrid   <- pc/storage/voxelgrid.VoxelGrid + pc/segmentation/voxelgrid (Segment)
//   pcgx::RegionGrowing     <- pc/segmentation/regiongrowing.RegionGrowing
//   pcgx::PointToPlaneICP   <- (extension, no counterpart in the reference) the same Fit shape with
//                              the point-to-plane evaluator / Gauss-Newton updater, HasHessian() == true
//   pcgx::Comm              <- (no counterpart: the reference is one process) the exchange of the
//                              several-GPU paths: PointToPointICP::FitSharded, VoxelGrid::FilterSharded
#pragma once
#include <array>
#include <functional>
#include <initializer_list>
#include <memory>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pcgx.h"

namespace pcgx {

struct Error : std::runtime_error {
  pcgx_status code;
  Error(pcgx_status c, const std::string &m) : std::runtime_error(m), code(c) {}
};
struct ErrNoPoint : Error { using Error::Error; };          // pc/minmax.go:11
struct ErrNotEnoughPairs : Error { using Error::Error; };   // icp/evaluator.go:16
struct ErrSingular : Error { using Error::Error; };         // point-to-plane extension

inline void check(pcgx_status rc) {
  if (rc == PCGX_OK) return;
  char buf[512];
  pcgx_last_error(buf, sizeof buf);
  if (rc == PCGX_E_NO_POINT) throw ErrNoPoint(rc, buf);
  if (rc == PCGX_E_NOT_ENOUGH_PAIRS) throw ErrNotEnoughPairs(rc, buf);
  if (rc == PCGX_E_SINGULAR) throw ErrSingular(rc, buf);
  throw Error(rc, buf);
}

using Vec3 = std::array<float, 3>;   // mat.Vec3
using Mat4 = std::array<float, 16>;  // mat.Mat4, column-major

struct Neighbor {  // pc/storage/search.go:8-11
  int64_t ID;
  float DistSq;
};

// pc.PointCloud's layout contract (pc/pointcloud.go:64-78): AoS records.
struct CloudView {
  const void *data;
  int64_t points;
  int32_t stride, xyz_offset;
};

// The exchange between the ranks of a several-GPU run (one process per GPU, include/pcgx.h
// "sharded"): RCCL inside the library from an id that rank 0 makes and the host passes on, or a host
// function that sums `count` float64 in place over the ranks.
class Comm {
 public:
  static pcgx_comm_id UniqueId() { pcgx_comm_id id; check(pcgx_comm_unique_id(&id)); return id; }
  Comm(int rank, int world, const pcgx_comm_id &id) { check(pcgx_comm_init(rank, world, &id, &h_)); }
  Comm(int rank, int world, pcgx_allreduce_fn fn, void *user) { check(pcgx_comm_init_callback(rank, world, fn, user, &h_)); }
  ~Comm() { pcgx_comm_free(h_); }
  Comm(const Comm &) = delete;
  Comm &operator=(const Comm &) = delete;
  int Rank() const { int32_t r, w; check(pcgx_comm_rank(h_, &r, &w)); return r; }
  int World() const { int32_t r, w; check(pcgx_comm_rank(h_, &r, &w)); return w; }
  pcgx_comm *handle() const { return h_; }

 private:
  pcgx_comm *h_ = nullptr;
};

// the struct layouts this header was compiled against must be the library's (call once after loading it)
inline void CheckAbi() {
  if (pcgx_abi_version() != PCGX_ABI_VERSION)
    throw Error(PCGX_E_INVALID, "libpcgx.so speaks another ABI version than the pcgx.h this program was built against");
}

class KDTree;
// kdtree.KDTreeOption (kdtree.go:31): New(ra, opts...) / (*KDTree).With(opts...) apply them to the tree value.
using KDTreeOption = std::function<void(KDTree &)>;

class KDTree {  // pc/storage/kdtree/kdtree.go:14-23
 public:
  float MinDistSq = 0.0f;
  explicit KDTree(const CloudView &c, std::initializer_list<KDTreeOption> opts = {}) {
    pcgx_kdtree *h = nullptr;
    check(pcgx_kdtree_build(c.data, c.points, c.stride, c.xyz_offset, &h));
    h_ = std::shared_ptr<pcgx_kdtree>(h, [](pcgx_kdtree *p) { pcgx_kdtree_free(p); });
    for (const auto &o : opts) o(*this);
  }
  explicit KDTree(const std::vector<Vec3> &pts, std::initializer_list<KDTreeOption> opts = {})
      : KDTree(CloudView{pts.data(), (int64_t)pts.size(), 12, 0}, opts) {}
  // With (kdtree.go:58-65): a shallow copy with the options applied; the copies share the device tree.
  KDTree With(std::initializer_list<KDTreeOption> opts) const {
    KDTree k2(*this);
    for (const auto &o : opts) o(k2);
    return k2;
  }
  static KDTreeOption WithMinDistSq(float d) { return [d](KDTree &k) { k.MinDistSq = d; }; }
  int64_t Len() const { int64_t n; check(pcgx_kdtree_len(h_.get(), &n)); return n; }
  Vec3 Vec3At(int64_t i) const { Vec3 v; check(pcgx_kdtree_points(h_.get(), &i, 1, v.data())); return v; }
  Neighbor Nearest(const Vec3 &p, float maxRange) const { return NearestBatch({p}, maxRange)[0]; }
  std::vector<Neighbor> NearestBatch(const std::vector<Vec3> &q, float maxRange) const {
    std::vector<int64_t> ids(q.size());
    std::vector<float> d(q.size());
    check(pcgx_kdtree_nearest_batch(h_.get(), q.empty() ? nullptr : q[0].data(), (int64_t)q.size(), maxRange, MinDistSq,
                                    ids.data(), d.data()));
    std::vector<Neighbor> out(q.size());
    for (size_t i = 0; i < q.size(); i++) out[i] = Neighbor{ids[i], d[i]};
    return out;
  }
  // KDTree.Range (kdtree.go:148-197): neighbours with DistSq < maxRange^2, sorted by DistSq.
  std::vector<Neighbor> Range(const Vec3 &p, float maxRange) const {
    int64_t cnt = 0;
    check(pcgx_kdtree_range_count(h_.get(), p.data(), 1, maxRange, &cnt));
    const int64_t offs[2] = {0, cnt};
    std::vector<int64_t> ids((size_t)cnt);
    std::vector<float> d((size_t)cnt);
    check(pcgx_kdtree_range_fill(h_.get(), p.data(), 1, maxRange, offs, ids.data(), d.data()));
    std::vector<Neighbor> out((size_t)cnt);
    for (int64_t i = 0; i < cnt; i++) out[(size_t)i] = Neighbor{ids[(size_t)i], d[(size_t)i]};
    return out;
  }
  // KDTree.DeletePoint (kdtree.go:322-332); std::out_of_range for an id outside [0, Len()).
  void DeletePoint(int64_t pID) {
    pcgx_status rc = pcgx_kdtree_delete_points(h_.get(), &pID, 1);
    if (rc == PCGX_E_OUT_OF_RANGE) {
      char buf[256];
      pcgx_last_error(buf, sizeof buf);
      throw std::out_of_range(buf);
    }
    check(rc);
  }
  int32_t MaxDepth() const { int32_t d; check(pcgx_kdtree_max_depth(h_.get(), &d)); return d; }
  const pcgx_kdtree *handle() const { return h_.get(); }

 private:
  KDTree(const KDTree &) = default;
  std::shared_ptr<pcgx_kdtree> h_;
};

class VoxelGrid {  // pc/filter/voxelgrid/voxelgrid.go:23-33 + option.go:14-18
 public:
  Vec3 LeafSize;
  std::array<int32_t, 3> ChunkSize{0, 0, 0};
  explicit VoxelGrid(Vec3 leaf) : LeafSize(leaf) {}
  VoxelGrid &WithChunkSize(std::array<int32_t, 3> s) { ChunkSize = s; return *this; }
  // Filter: returns the output records (Width = size()/stride, Height = 1).
  std::vector<uint8_t> Filter(const CloudView &c) const {
    std::vector<uint8_t> out((size_t)c.points * c.stride);
    int64_t m = 0;
    check(pcgx_voxel_filter(c.data, c.points, c.stride, c.xyz_offset, LeafSize.data(), ChunkSize.data(), out.data(), &m));
    out.resize((size_t)m * c.stride);
    return out;
  }
  // This rank's share of Filter(c) over the ranks of `comm` (every rank passes the same cloud); the
  // ranks' results, rank 0's first, are Filter's output record for record.  Collective.
  std::vector<uint8_t> FilterSharded(const CloudView &c, const Comm &comm) const {
    std::vector<uint8_t> out((size_t)c.points * c.stride);
    int64_t m = 0;
    check(pcgx_voxel_filter_sharded(comm.handle(), c.data, c.points, c.stride, c.xyz_offset, LeafSize.data(),
                                    ChunkSize.data(), out.data(), &m));
    out.resize((size_t)m * c.stride);
    return out;
  }
};

// pc/storage/voxelgrid.VoxelGrid (voxelgrid.go:7-122) filled with Add(point i, i) for a whole
// cloud, plus pc/segmentation/voxelgrid's Segment (voxelgrid.go:39-73).
class BucketVoxelGrid {
 public:
  BucketVoxelGrid(float resolution, std::array<int64_t, 3> size, Vec3 origin, const CloudView &c) {
    check(pcgx_bucket_grid_build(c.data, c.points, c.stride, c.xyz_offset, resolution, size.data(), origin.data(), &h_));
  }
  ~BucketVoxelGrid() { pcgx_bucket_grid_free(h_); }
  BucketVoxelGrid(const BucketVoxelGrid &) = delete;
  BucketVoxelGrid &operator=(const BucketVoxelGrid &) = delete;
  int64_t Len() const { int64_t n; check(pcgx_bucket_grid_counts(h_, &n, nullptr, nullptr)); return n; }
  // Get(p): false = nil (p outside the grid), else the ids of p's voxel in insertion order
  bool Get(const Vec3 &p, std::vector<int64_t> *ids) const {
    int64_t cnt = 0;
    check(pcgx_bucket_grid_get(h_, p.data(), nullptr, 0, &cnt));
    if (cnt < 0) return false;
    ids->resize((size_t)cnt);
    if (cnt > 0) check(pcgx_bucket_grid_get(h_, p.data(), ids->data(), cnt, &cnt));
    return true;
  }
  // Segment(p) in the reference's order (voxelgrid.go:39-73)
  std::vector<int64_t> Segment(const Vec3 &p) {
    int64_t cnt = 0;
    check(pcgx_bucket_grid_segment_bfs(h_, p.data(), nullptr, 0, &cnt));
    std::vector<int64_t> out((size_t)cnt);
    if (cnt > 0) check(pcgx_bucket_grid_segment_bfs(h_, p.data(), out.data(), cnt, &cnt));
    return out;
  }

 private:
  pcgx_bucket_grid *h_ = nullptr;
};

// pc/segmentation/regiongrowing.RegionGrowing (regiongrowing.go:13-56): New(search, propertyIter)
class RegionGrowing {
 public:
  RegionGrowing(const KDTree &search, std::vector<uint32_t> property) : t_(search), labels_(std::move(property)) {
    if ((int64_t)labels_.size() != t_.Len()) throw Error(PCGX_E_INVALID, "one property value per point is required");
  }
  // Segment(p, maxRange) in the reference's order (regiongrowing.go:23-56)
  std::vector<int64_t> Segment(const Vec3 &p, float maxRange) {
    std::vector<int64_t> out(labels_.size());
    int64_t cnt = 0;
    check(pcgx_region_growing_segment_bfs(t_.handle(), labels_.data(), p.data(), maxRange, out.data(),
                                          (int64_t)out.size(), &cnt));
    out.resize((size_t)cnt);
    return out;
  }
  // the same set for many seeds: regions of the whole cloud labelled once per maxRange, ascending id
  std::vector<int64_t> SegmentById(const Vec3 &p, float maxRange) {
    if (comp_.empty() || maxRange != range_) {
      comp_.resize(labels_.size());
      check(pcgx_region_growing_components(t_.handle(), labels_.data(), maxRange, comp_.data()));
      range_ = maxRange;
    }
    std::vector<int64_t> out(labels_.size());
    int64_t cnt = 0;
    check(pcgx_region_growing_segment(t_.handle(), labels_.data(), comp_.data(), p.data(), maxRange, out.data(),
                                      (int64_t)out.size(), &cnt));
    out.resize((size_t)cnt);
    return out;
  }

 private:
  const KDTree &t_;
  std::vector<uint32_t> labels_;
  std::vector<int64_t> comp_;
  float range_ = 0.0f;
};

// icp.PointToPointCorrespondence / NearestPointCorresponder (correspondence.go:8-37): Pairs over pcgx_icp_pairs --
// one batched nearest-neighbour pass, the matched targets compacted in target order.
struct PointToPointCorrespondence {
  int64_t BaseID, TargetID;
  float SquaredDistance;
};
struct NearestPointCorresponder {
  float MaxDist = 0.0f;
  std::vector<PointToPointCorrespondence> Pairs(const KDTree &base, const std::vector<Vec3> &target) const {
    const size_t n = target.size();
    std::vector<int64_t> b(n), t(n);
    std::vector<float> d(n);
    int64_t np = 0;
    check(pcgx_icp_pairs(base.handle(), n ? target[0].data() : nullptr, (int64_t)n, MaxDist, base.MinDistSq, b.data(), t.data(),
                         d.data(), &np));
    std::vector<PointToPointCorrespondence> out((size_t)np);
    for (size_t i = 0; i < out.size(); i++) out[i] = PointToPointCorrespondence{b[i], t[i], d[i]};
    return out;
  }
};

struct Stat {  // icp/stat.go:3-6
  pcgx_icp_evaluated Evaluated;
  int NumIteration;
};

// PointToPointEvaluator.WeightFn (evaluator.go:19-23,72): the reference takes any closure, the device
// one of the built-in forms (include/pcgx.h PCGX_WEIGHT_*); Kind 0 = DefaultEvaluateWeightFn (w = 1).
struct WeightFn {
  int32_t Kind = PCGX_WEIGHT_ONE;
  float A = 0.0f;
};

class PointToPointICP {  // icp.go:18-67 with evaluator.go:69-73 and updater.go:18-22 options
 public:
  float MaxDist = 0.0f;
  int MinPairs = 0;
  WeightFn EvaluateWeight;  // evaluator.go:72
  std::array<float, 6> Weight{}, Threshold{};
  int MaxIteration = 0;
  // Sums (include/pcgx.h PCGX_SUMS_*).  Default PCGX_SUMS_REFERENCE: the evaluator's sums as the reference
  // forms them (sequential float32 additions in target order, evaluator.go:122-145): Evaluated and every
  // pose bit-identical to the Go code's.  PCGX_SUMS_F64_TREE: float64 reductions of the same float32 terms
  // (faster; equal up to the reference's own rounding noise; what FitSharded computes over several ranks).
  int32_t Sums = PCGX_SUMS_REFERENCE;
  std::pair<Mat4, Stat> Fit(const KDTree &base, const std::vector<Vec3> &target) const {
    const pcgx_icp_params p = params(base);
    Mat4 t;
    pcgx_icp_stat st{};
    const float *tp = target.empty() ? nullptr : target[0].data();
    check(pcgx_icp_fit(base.handle(), tp, (int64_t)target.size(), &p, t.data(), &st));
    return {t, Stat{st.evaluated, st.num_iteration}};
  }
  // Fit with the target spread over the device slots of THIS process (pcgx_init_devices): bases[r] the replica built
  // with slot r current, tiles[r] slot r's part.  Default sums: the reference's Fit of the tiles one after the other.
  std::pair<Mat4, Stat> FitMulti(const std::vector<const KDTree *> &bases, const std::vector<std::vector<Vec3>> &tiles) const {
    const pcgx_icp_params p = params(*bases.at(0));
    std::vector<const pcgx_kdtree *> hb;
    std::vector<const float *> ht;
    std::vector<int64_t> hn;
    for (size_t r = 0; r < bases.size(); r++) {
      hb.push_back(bases[r]->handle());
      ht.push_back(tiles.at(r).empty() ? nullptr : tiles[r][0].data());
      hn.push_back((int64_t)tiles[r].size());
    }
    Mat4 t;
    pcgx_icp_stat st{};
    check(pcgx_icp_fit_multi((int32_t)bases.size(), hb.data(), ht.data(), hn.data(), &p, t.data(), &st));
    return {t, Stat{st.evaluated, st.num_iteration}};
  }
  // Fit on this rank's tile of the target; every rank returns the same transform (default sums: the reference's over
  // the ranks' tiles one after the other; PCGX_SUMS_F64_TREE: one all-reduce of float64 sums per iteration; one
  // rank: Fit).  Collective.
  std::pair<Mat4, Stat> FitSharded(const KDTree &base, const std::vector<Vec3> &tile, const Comm &comm) const {
    const pcgx_icp_params p = params(base);
    Mat4 t;
    pcgx_icp_stat st{};
    check(pcgx_icp_fit_sharded(base.handle(), tile.empty() ? nullptr : tile[0].data(), (int64_t)tile.size(), &p,
                               comm.handle(), t.data(), &st));
    return {t, Stat{st.evaluated, st.num_iteration}};
  }

 private:
  pcgx_icp_params params(const KDTree &base) const {
    pcgx_icp_params p{};
    p.max_dist = MaxDist;
    p.min_dist_sq = base.MinDistSq;
    p.min_pairs = MinPairs;
    p.weight_fn = EvaluateWeight.Kind;
    p.weight_fn_param = EvaluateWeight.A;
    p.sums_mode = Sums;
    std::memcpy(p.weight, Weight.data(), sizeof p.weight);
    std::memcpy(p.threshold, Threshold.data(), sizeof p.threshold);
    p.max_iteration = MaxIteration;
    return p;
  }
};

// Extension (no counterpart in the reference; include/pcgx.h "point-to-plane ICP (extension)"):
// the Fit loop with the point-to-plane evaluator (fills Evaluated.Hessian) and a Gauss-Newton updater.
struct PlaneStat {
  pcgx_icp_evaluated Evaluated;
  std::array<float, 36> Hessian;  // Evaluated.Hessian (mat.Mat6)
  int NumIteration;
};

class PointToPlaneICP {
 public:
  float MaxDist = 0.0f;
  int MinPairs = 0;
  std::array<float, 6> Threshold{};
  int MaxIteration = 0;
  float Damping = 0.0f;
  bool HasGradient() const { return true; }
  bool HasHessian() const { return true; }
  // baseNormals: one unit normal per base point, in the tree's id order.
  std::pair<Mat4, PlaneStat> Fit(const KDTree &base, const std::vector<Vec3> &baseNormals,
                                 const std::vector<Vec3> &target) const {
    if ((int64_t)baseNormals.size() != base.Len()) throw Error(PCGX_E_INVALID, "one normal per base point is required");
    pcgx_icp_params p{};
    p.max_dist = MaxDist;
    p.min_pairs = MinPairs;
    std::memcpy(p.threshold, Threshold.data(), sizeof p.threshold);
    p.max_iteration = MaxIteration;
    Mat4 t;
    PlaneStat ps{};
    pcgx_icp_stat st{};
    check(pcgx_icp_plane_fit(base.handle(), baseNormals[0].data(), target.empty() ? nullptr : target[0].data(),
                             (int64_t)target.size(), &p, Damping, t.data(), &st, ps.Hessian.data()));
    ps.Evaluated = st.evaluated;
    ps.NumIteration = st.num_iteration;
    return {t, ps};
  }
};

// ---- the reference's package surface, name for name ------------------------------------------------------------
// What go/pc/storage/kdtree, go/pc/filter/voxelgrid and go/pc/registration/icp are to a Go caller (see go/README.md):
// the exported names of the three reference packages over the classes above, so that code written against
// kdtree.New(ra, opts...), voxelgrid.New(leaf, voxelgrid.WithChunkSize(s)), icp.PointToPointICPGradient{Evaluator,
// UpdaterFactory}.Fit(base, target) reads the same here.  tests/test_cpp_host.py runs them.
namespace kdtree {  // pc/storage/kdtree/kdtree.go:14-65
using KDTree = ::pcgx::KDTree;
using KDTreeOption = ::pcgx::KDTreeOption;
inline KDTree New(const std::vector<Vec3> &ra, std::initializer_list<KDTreeOption> opts = {}) { return KDTree(ra, opts); }
inline KDTree New(const CloudView &ra, std::initializer_list<KDTreeOption> opts = {}) { return KDTree(ra, opts); }
inline KDTreeOption WithMinDistSq(float d) { return KDTree::WithMinDistSq(d); }
}  // namespace kdtree

namespace voxelgrid {  // pc/filter/voxelgrid/voxelgrid.go:23-33, option.go:7-18
struct Options {
  Vec3 LeafSize{};
  std::array<int32_t, 3> ChunkSize{0, 0, 0};
};
using Option = std::function<void(Options &)>;
inline Option WithChunkSize(std::array<int32_t, 3> s) { return [s](Options &o) { o.ChunkSize = s; }; }
// filter.Filter (pc/filter/filter.go:7-9): Filter(cloud) -> the output records
inline ::pcgx::VoxelGrid New(Vec3 leafSize, std::initializer_list<Option> opts = {}) {
  Options o;
  o.LeafSize = leafSize;
  for (const auto &f : opts) f(o);
  ::pcgx::VoxelGrid vg(o.LeafSize);
  vg.WithChunkSize(o.ChunkSize);
  return vg;
}
}  // namespace voxelgrid

namespace icp {  // pc/registration/icp
using PointToPointCorrespondence = ::pcgx::PointToPointCorrespondence;  // correspondence.go:8-12
using NearestPointCorresponder = ::pcgx::NearestPointCorresponder;      // correspondence.go:18-37
using Stat = ::pcgx::Stat;                                              // stat.go:3-6
using Evaluated = pcgx_icp_evaluated;                                   // evaluator.go:25-30
using Weight = ::pcgx::WeightFn;
// PointToPointEvaluator (evaluator.go:69-73); WeightFn closures cannot cross to the device: Weight names a built-in form
struct PointToPointEvaluator {
  NearestPointCorresponder Corresponder;
  int MinPairs = 0;
  icp::Weight Weight;
  int32_t Sums = PCGX_SUMS_REFERENCE;
  bool HasGradient() const { return true; }
  bool HasHessian() const { return false; }
  Evaluated Evaluate(const KDTree &base, const std::vector<Vec3> &target) const {  // evaluator.go:91-189
    pcgx_icp_params p{};
    p.max_dist = Corresponder.MaxDist;
    p.min_dist_sq = base.MinDistSq;
    p.min_pairs = MinPairs;
    p.weight_fn = Weight.Kind;
    p.weight_fn_param = Weight.A;
    p.sums_mode = Sums;
    Evaluated ev{};
    check(pcgx_icp_evaluate_params(base.handle(), target.empty() ? nullptr : target[0].data(), (int64_t)target.size(), &p, &ev));
    return ev;
  }
};
struct GradientDescentUpdaterFactory {  // updater.go:18-37 (zero values: the reference's defaults)
  std::array<float, 6> Weight{}, Threshold{};
  int MaxIteration = 0;
};
struct PointToPointICPGradient {  // icp.go:18-67
  PointToPointEvaluator Evaluator;
  GradientDescentUpdaterFactory UpdaterFactory;
  std::pair<Mat4, Stat> Fit(const KDTree &base, const std::vector<Vec3> &target) const { return impl().Fit(base, target); }
  std::pair<Mat4, Stat> FitSharded(const KDTree &base, const std::vector<Vec3> &tile, const Comm &comm) const {
    return impl().FitSharded(base, tile, comm);
  }
  std::pair<Mat4, Stat> FitMulti(const std::vector<const KDTree *> &bases, const std::vector<std::vector<Vec3>> &tiles) const {
    return impl().FitMulti(bases, tiles);
  }

 private:
  PointToPointICP impl() const {
    PointToPointICP r;
    r.MaxDist = Evaluator.Corresponder.MaxDist;
    r.MinPairs = Evaluator.MinPairs;
    r.EvaluateWeight = Evaluator.Weight;
    r.Sums = Evaluator.Sums;
    r.Weight = UpdaterFactory.Weight;
    r.Threshold = UpdaterFactory.Threshold;
    r.MaxIteration = UpdaterFactory.MaxIteration;
    return r;
  }
};
}  // namespace icp

}  // namespace pcgx
