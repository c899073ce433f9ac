// Drives the C++ host mirror (pcgol_amd/host/pcgx.hpp) over the C ABI and prints results in a
// line format tests/test_cpp_host.py compares with the reference's known-answer tables
// (tests/golden/ref_*.json).  Input: a text file written by the test
//   P n            followed by n lines "x y z"        base cloud
//   Q m max_range  followed by m lines "x y z"        Nearest queries
//   R m            followed by m lines "x y z range"  Range queries
//   V n stride     followed by n lines "x y z label"  voxel cloud (stride 16: x y z label)
//   L lx ly lz cx cy cz                               leaf + chunk, runs the filter
#include <cinttypes>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <sstream>

#include "../../pcgol_amd/host/pcgx.hpp"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  try {
    pcgx::check(pcgx_init(0));
    std::ifstream in(argv[1]);
    std::string tag;
    std::vector<pcgx::Vec3> base;
    std::vector<uint8_t> vox;
    int64_t vox_n = 0;
    std::unique_ptr<pcgx::KDTree> tree;
    while (in >> tag) {
      if (tag == "P") {
        size_t n; in >> n;
        base.resize(n);
        for (auto &p : base) in >> p[0] >> p[1] >> p[2];
        tree.reset(new pcgx::KDTree(base));
        std::printf("tree len %" PRId64 " depth %d\n", tree->Len(), tree->MaxDepth());
      } else if (tag == "M") {
        float md; in >> md;
        tree->MinDistSq = md * md;
      } else if (tag == "Q") {
        size_t m; float mr; in >> m >> mr;
        std::vector<pcgx::Vec3> q(m);
        for (auto &p : q) in >> p[0] >> p[1] >> p[2];
        for (const auto &nb : tree->NearestBatch(q, mr)) std::printf("nearest %" PRId64 " %.9g\n", nb.ID, nb.DistSq);
      } else if (tag == "R") {
        size_t m; in >> m;
        for (size_t i = 0; i < m; i++) {
          pcgx::Vec3 p; float r; in >> p[0] >> p[1] >> p[2] >> r;
          std::printf("range");
          for (const auto &nb : tree->Range(p, r)) std::printf(" %" PRId64 ":%.9g", nb.ID, nb.DistSq);
          std::printf("\n");
        }
      } else if (tag == "V") {
        in >> vox_n;
        vox.resize((size_t)vox_n * 16);
        for (int64_t i = 0; i < vox_n; i++) {
          float xyz[3]; uint32_t label;
          in >> xyz[0] >> xyz[1] >> xyz[2] >> label;
          std::memcpy(&vox[(size_t)i * 16], xyz, 12);
          std::memcpy(&vox[(size_t)i * 16 + 12], &label, 4);
        }
      } else if (tag == "L") {
        pcgx::Vec3 leaf; std::array<int32_t, 3> chunk;
        in >> leaf[0] >> leaf[1] >> leaf[2] >> chunk[0] >> chunk[1] >> chunk[2];
        pcgx::VoxelGrid vg(leaf);
        vg.WithChunkSize(chunk);
        auto out = vg.Filter(pcgx::CloudView{vox.data(), vox_n, 16, 0});
        std::printf("voxel");
        for (size_t i = 0; i < out.size() / 16; i++) {
          float xyz[3]; uint32_t label;
          std::memcpy(xyz, &out[i * 16], 12);
          std::memcpy(&label, &out[i * 16 + 12], 4);
          std::printf(" %.9g,%.9g,%.9g,%u", xyz[0], xyz[1], xyz[2], label);
        }
        std::printf("\n");
      } else if (tag == "G") {  // bucket grid over the current base cloud: resolution, size, origin, then seed
        float res; std::array<int64_t, 3> size; pcgx::Vec3 origin, seed;
        in >> res >> size[0] >> size[1] >> size[2] >> origin[0] >> origin[1] >> origin[2] >> seed[0] >> seed[1] >> seed[2];
        pcgx::BucketVoxelGrid g(res, size, origin, pcgx::CloudView{base.data(), (int64_t)base.size(), 12, 0});
        auto seg = g.Segment(seed);
        std::printf("segment");
        for (int64_t id : seg) std::printf(" %" PRId64, id);
        std::printf("\n");
        std::vector<int64_t> ids;
        std::printf("get %d", g.Get(seed, &ids) ? (int)ids.size() : -1);
        std::printf(" len %" PRId64 "\n", g.Len());
      } else if (tag == "W") {  // region growing: labels = (id % modulo), seed, maxRange
        uint32_t mod; pcgx::Vec3 seed; float mr;
        in >> mod >> seed[0] >> seed[1] >> seed[2] >> mr;
        std::vector<uint32_t> lab(base.size());
        for (size_t i = 0; i < lab.size(); i++) lab[i] = (uint32_t)(i % mod);
        pcgx::RegionGrowing rg(*tree, lab);
        std::printf("region");
        for (int64_t id : rg.Segment(seed, mr)) std::printf(" %" PRId64, id);
        std::printf("\n");
      } else if (tag == "I") {  // ICP: target = base + (dx,dy,dz); point-to-point Fit with MinPairs / MaxDist
        float dx, dy, dz, maxd; int minp;
        in >> dx >> dy >> dz >> maxd >> minp;
        std::vector<pcgx::Vec3> target(base);
        for (auto &p : target) { p[0] += dx; p[1] += dy; p[2] += dz; }
        pcgx::PointToPointICP reg;
        reg.MaxDist = maxd;
        reg.MinPairs = minp;
        auto r = reg.Fit(*tree, target);
        std::printf("icp iters %d value %.9g trans", r.second.NumIteration, r.second.Evaluated.value);
        for (float v : r.first) std::printf(" %.9g", v);
        std::printf("\n");
        {  // the default (reference) sums with a built-in weight; the float64-tree mode; and the one-rank forms of
           // the several-GPU entry points
          pcgx::PointToPointICP f64 = reg;
          f64.Sums = PCGX_SUMS_F64_TREE;
          auto rf = f64.Fit(*tree, target);
          std::printf("icp_f64 iters %d value %.9g trans", rf.second.NumIteration, rf.second.Evaluated.value);
          for (float v : rf.first) std::printf(" %.9g", v);
          std::printf("\n");
          pcgx::PointToPointICP st = reg;
          st.EvaluateWeight = pcgx::WeightFn{PCGX_WEIGHT_HUBER, 0.0004f};
          auto rs = st.Fit(*tree, target);
          std::printf("icp_strict iters %d value %.9g trans", rs.second.NumIteration, rs.second.Evaluated.value);
          for (float v : rs.first) std::printf(" %.9g", v);
          std::printf("\n");
          pcgx::Comm comm(0, 1, [](double *, int32_t, void *) -> int32_t { return 0; }, nullptr);
          auto r1 = reg.FitSharded(*tree, target, comm);
          std::printf("sharded1_icp same %d\n", (int)(r1.first == r.first && r1.second.NumIteration == r.second.NumIteration));
          pcgx::VoxelGrid vg(pcgx::Vec3{0.5f, 0.5f, 0.5f});
          const pcgx::CloudView cv{base.data(), (int64_t)base.size(), 12, 0};
          std::printf("sharded1_voxel same %d world %d\n", (int)(vg.FilterSharded(cv, comm) == vg.Filter(cv)), comm.World());
        }
        {  // the Go seams: KDTreeOption / With (kdtree.go:31-65) and a corresponder over pcgx_icp_pairs (correspondence.go:14-37)
          const pcgx::KDTree approx = tree->With({pcgx::KDTree::WithMinDistSq(0.25f)});
          std::printf("with mindist %.9g shared %d len %" PRId64 "\n", approx.MinDistSq, (int)(approx.handle() == tree->handle()),
                      approx.Len());
          pcgx::NearestPointCorresponder cor;
          cor.MaxDist = maxd;
          const auto pairs = cor.Pairs(*tree, target);
          std::printf("pairs %zu", pairs.size());
          for (size_t i = 0; i < pairs.size() && i < 8; i++)
            std::printf(" %" PRId64 ":%" PRId64 ":%.9g", pairs[i].BaseID, pairs[i].TargetID, pairs[i].SquaredDistance);
          std::printf("\n");
        }
        {  // the reference's package surface, name for name (pcgx::kdtree / voxelgrid / icp): the same results
          const pcgx::kdtree::KDTree kt = pcgx::kdtree::New(base);
          pcgx::icp::PointToPointICPGradient pp;
          pp.Evaluator.Corresponder.MaxDist = maxd;
          pp.Evaluator.MinPairs = minp;
          const auto rn = pp.Fit(kt, target);
          const pcgx::icp::Evaluated ev = pp.Evaluator.Evaluate(kt, target);
          std::printf("named_icp same %d has_gradient %d value0 %.9g\n",
                      (int)(rn.first == r.first && rn.second.NumIteration == r.second.NumIteration), (int)pp.Evaluator.HasGradient(),
                      ev.value);
          const pcgx::CloudView cv{base.data(), (int64_t)base.size(), 12, 0};
          const auto a = pcgx::voxelgrid::New(pcgx::Vec3{0.5f, 0.5f, 0.5f}, {pcgx::voxelgrid::WithChunkSize({3, 3, 3})}).Filter(cv);
          pcgx::VoxelGrid vg2(pcgx::Vec3{0.5f, 0.5f, 0.5f});
          vg2.WithChunkSize({3, 3, 3});
          std::printf("named_voxel same %d records %zu\n", (int)(a == vg2.Filter(cv)), a.size() / 12);
        }
        try {
          reg.MinPairs = (int)base.size() + 1;
          reg.Fit(*tree, target);
          std::printf("icp_minpairs no-error\n");
        } catch (const pcgx::ErrNotEnoughPairs &) {
          std::printf("icp_minpairs ErrNotEnoughPairs\n");
        }
      }
    }
    try {
      pcgx::KDTree empty(std::vector<pcgx::Vec3>{});
      std::printf("empty no-error\n");
    } catch (const pcgx::ErrNoPoint &) {
      std::printf("empty ErrNoPoint\n");
    }
  } catch (const std::exception &e) {
    std::printf("EXCEPTION %s\n", e.what());
    return 1;
  }
  return 0;
}
