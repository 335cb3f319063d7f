// Builds a roadmap the way apps/create_roadmap.cpp:252-331 drives motion_planning::VoxelCachedLazyPRM -- createRoadmap with its
// option flags, a second createRoadmap that grows it, precomputeValidity, clearDisconnectedVertices, then queries -- through
// include/tendon_hip_shim.hpp only, and writes every stage as raw arrays for tests/test_cpp_shim.py to check against the oracle.
//
//   shim_roadmap_test <grid file: 64^3 uint64 blocks of a 256^3 grid over [-0.25, 0.25]^3> <output directory>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "tendon_hip_shim.hpp"

using namespace tendon_hip;
using Planner = motion_planning::VoxelCachedLazyPRM;

static std::string g_dir;

template <class T> static void dump(const std::string &name, const std::vector<T> &v) {
  const std::string path = g_dir + "/" + name;
  FILE *f = std::fopen(path.c_str(), "wb");
  if (!f) { std::perror(path.c_str()); std::exit(2); }
  if (!v.empty() && std::fwrite(v.data(), sizeof(T), v.size(), f) != v.size()) { std::perror(path.c_str()); std::exit(2); }
  std::fclose(f);
}
static void dump_bits(const std::string &name, const std::vector<bool> &b) {
  std::vector<uint8_t> v(b.size());
  for (size_t i = 0; i < b.size(); i++) v[i] = b[i];
  dump(name, v);
}
static void dump_caches(const std::string &tag, const motion_planning::VoxelCaches &c) {
  dump(tag + "_off.i64", c.offsets); dump(tag + "_ids.u32", c.block_ids); dump(tag + "_masks.u64", c.masks); dump_bits(tag + "_usable.u8", c.usable);
}
static void dump_graph(const std::string &tag, Planner &prm) {
  dump(tag + "_states.f64", prm.states()); dump(tag + "_tips.f64", prm.tipPositions()); dump(tag + "_edges.i32", prm.edges());
  dump_caches(tag + "_vc", prm.vertexVoxels()); dump_caches(tag + "_ec", prm.edgeVoxels());
  std::vector<uint8_t> vs, es;
  prm.validity(vs, es);
  dump(tag + "_vstat.u8", vs); dump(tag + "_estat.u8", es);
}
static void dump_report(const std::string &tag, const Planner::BuildReport &r) {
  dump(tag + "_cand.i32", r.candidate_edges); dump_bits(tag + "_acc.u8", r.accepted); dump(tag + "_nfk.i32", r.n_fk);
  dump(tag + "_cidx.i64", r.candidate_index);
  dump(tag + "_meta.i64", std::vector<int64_t>{r.candidates_tried, r.k});
}

int main(int argc, char **argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: %s <grid file> <output directory>\n", argv[0]); return 2; }
  g_dir = argv[2];
  // workloads.robot_config3: 4 tendons, quadratic routing in angle, linear in radius, 129 backbone points
  tendon::TendonRobot robot;
  robot.specs.dL = 0.2 / 128;
  const double c1[4] = {3.0, -2.0, 4.0, -5.0}, c2[4] = {10.0, 15.0, -12.0, 8.0}, d1[4] = {-0.01, 0.005, 0.0, -0.005};
  for (int k = 0; k < 4; k++) {
    tendon::TendonSpecs t;
    t.C = {M_PI * k / 2, c1[k], c2[k]};
    t.D = {0.01, d1[k], 0.0};
    robot.tendons.push_back(t);
  }
  collision::VoxelOctree vox(256);
  vox.set_xlim(-0.25, 0.25); vox.set_ylim(-0.25, 0.25); vox.set_zlim(-0.25, 0.25);
  {
    FILE *f = std::fopen(argv[1], "rb");
    if (!f || std::fread(vox.blocks().data(), sizeof(uint64_t), vox.blocks().size(), f) != vox.blocks().size()) { std::perror(argv[1]); return 2; }
    std::fclose(f);
  }
  motion_planning::VoxelEnvironment env;
  motion_planning::VoxelBackboneValidityChecker vc(robot, env, vox);
  motion_planning::VoxelBackboneMotionValidator mv(vc);

  // stage A: create_roadmap --voxelize-vertices/edges --check-vertex/edge-collision, N = 2000, 8 nearest (the milestone itself counted)
  Planner prm(vc, mv, /*seed=*/11);
  prm.setMaxNearestNeighbors(8);
  prm.setRange(1e9);                                     // no distance bound: the k nearest, whatever their distance
  prm.createRoadmap(2000, Planner::ValidateVertices | Planner::ValidateEdges);
  if (prm.milestoneCount() != 2000) { std::fprintf(stderr, "stage A: %zu milestones\n", prm.milestoneCount()); return 3; }
  dump_graph("A", prm); dump_report("A", prm.lastBuild());
  prm.createRoadmap(1500, Planner::LazyRoadmap);         // already larger: nothing happens (:1387-1391)
  if (prm.milestoneCount() != 2000) return 3;

  // stage B: grow to 2300 with shape checks only (VoxelizeVertices | VoxelizeEdges): the new milestones may collide
  prm.createRoadmap(2300, Planner::VoxelizeVertices | Planner::VoxelizeEdges);
  dump_graph("B", prm); dump_report("B", prm.lastBuild());

  // stage C: precomputeValidity -- colliding milestones and edges leave -- then only the largest component stays
  prm.precomputeValidity();
  dump_graph("C", prm);
  prm.clearDisconnectedVertices();
  dump_graph("D", prm);

  // stage E: queries on the finished roadmap (every item known valid: nothing is tested again)
  {
    const size_t V = prm.milestoneCount();
    std::vector<int32_t> starts, goals;
    uint64_t x = 88172645463325252ull;
    for (int q = 0; q < 300; q++) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17; starts.push_back((int32_t)(x % V));
      x ^= x << 13; x ^= x >> 7; x ^= x << 17; goals.push_back((int32_t)(x % V));
    }
    prm.prepare(8);
    auto sol = prm.solveWithRoadmap(starts, goals);
    std::vector<int32_t> flat; std::vector<int64_t> off{0};
    for (auto &p : sol.paths) { flat.insert(flat.end(), p.begin(), p.end()); off.push_back((int64_t)flat.size()); }
    dump("E_starts.i32", starts); dump("E_goals.i32", goals); dump("E_status.i32", sol.status); dump("E_cost.f64", sol.cost);
    dump("E_paths.i32", flat); dump("E_poff.i64", off);
    dump("E_stats.i64", std::vector<int64_t>{sol.stats.rounds, sol.stats.items_checked, sol.stats.astar_runs, sol.stats.expanded});
  }

  // stage F: a LAZY roadmap (no option: sampled states and empty edges, :1417-1419) on a second planner with the same seed: its
  // milestones are the first candidates unchecked; the first query voxelises what it needs (here: everything, in two batched calls)
  {
    Planner lazy(vc, mv, /*seed=*/11);
    lazy.setStarConnectionStrategy();
    lazy.createRoadmap(600);
    dump_graph("F0", lazy); dump_report("F0", lazy.lastBuild());
    std::vector<int32_t> starts, goals;
    for (int q = 0; q < 64; q++) { starts.push_back((q * 37) % 600); goals.push_back((q * 101 + 17) % 600); }
    auto sol = lazy.solveWithRoadmap(starts, goals);
    std::vector<int32_t> flat; std::vector<int64_t> off{0};
    for (auto &p : sol.paths) { flat.insert(flat.end(), p.begin(), p.end()); off.push_back((int64_t)flat.size()); }
    dump("F_starts.i32", starts); dump("F_goals.i32", goals); dump("F_status.i32", sol.status); dump("F_cost.f64", sol.cost);
    dump("F_paths.i32", flat); dump("F_poff.i64", off);
    dump_graph("F", lazy);
    // error behaviour of the builder calls
    int caught = 0;
    Planner query_only(vc, lazy.states(), lazy.milestoneCount(), lazy.edges());
    try { query_only.createRoadmap(700); } catch (const std::runtime_error &) { caught++; }
    try { lazy.setMaxNearestNeighbors(3); } catch (const std::runtime_error &) { caught++; }       // star strategy: as the reference (:1328-1330)
    try { lazy.setSamplingBounds({0.0}, {1.0}); } catch (const std::invalid_argument &) { caught++; }
    if (caught != 3) { std::fprintf(stderr, "stage F: %d of 3 errors\n", caught); return 3; }
  }
  std::printf("roadmap stages written\n");
  return 0;
}
