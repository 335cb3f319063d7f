// Exercises include/tendon_hip_shim.hpp the way reference-side C++ would: TendonRobot::shape,
// VoxelBackboneValidityChecker::isValid / isValidBatch, and the exception mapping.  Prints
// results for tests/test_gpu_cpp_shim.py to compare with the oracle.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tendon_hip_shim.hpp"

using namespace tendon_hip;

int main(int argc, char **argv) {
  const bool compile_only = argc > 1 && std::string(argv[1]) == "--no-gpu";
  tendon::TendonRobot robot;
  robot.specs.dL = 0.2 / 128;
  for (int k = 0; k < 3; k++) {
    tendon::TendonSpecs t;
    t.C = {2 * M_PI * k / 3, 5.0};
    t.D = {0.01};
    robot.tendons.push_back(t);
  }
  // error mapping that needs no GPU
  int caught = 0;
  try { collision::VoxelOctree bad(100); } catch (const std::invalid_argument &) { caught++; }
  try { collision::VoxelOctree v(8); v.set_xlim(1, 1); } catch (const std::length_error &) { caught++; }
  try { robot.calc_dl({1.0}, {1.0, 2.0}); } catch (const std::out_of_range &) { caught++; }
  std::printf("caught %d\n", caught);
  if (compile_only) return caught == 3 ? 0 : 1;

  try { robot.shape({1.0, 2.0}); } catch (const std::invalid_argument &e) { std::printf("invalid_argument: %s\n", e.what()); }
  const std::vector<std::vector<double>> states = {{0, 0, 0}, {8, 3, 1}, {2.5, 9.0, 4.0}, {19.0, 0.5, 17.0}};
  for (auto &s : states) {
    auto res = robot.shape(s);
    std::printf("shape %zu %.17g %.17g %.17g %.17g %d\n", res.p.size(), res.p.back()[0], res.p.back()[1], res.p.back()[2],
                res.L_i[0], (int)res.converged);
  }
  collision::VoxelOctree vox(256);
  vox.set_xlim(-0.25, 0.25); vox.set_ylim(-0.25, 0.25); vox.set_zlim(-0.25, 0.25);
  // a slab of obstacles at x in [0.03, 0.05], z >= 0.1
  for (size_t ix = 143; ix < 154; ix++) for (size_t iy = 64; iy < 192; iy++) for (size_t iz = 179; iz < 240; iz++) vox.set_cell(ix, iy, iz);
  motion_planning::VoxelEnvironment env;
  motion_planning::VoxelBackboneValidityChecker vc(robot, env, vox);
  std::vector<double> flat;
  for (auto &s : states) { std::printf("isValid %d\n", (int)vc.isValid(s)); flat.insert(flat.end(), s.begin(), s.end()); }
  std::vector<double> tips; std::vector<uint8_t> flags;
  auto v = vc.isValidBatch(flat, states.size(), &tips, &flags);
  for (size_t i = 0; i < v.size(); i++) std::printf("batch %d %u %.17g\n", (int)v[i], flags[i], tips[3 * i]);
  // edges: adaptive bisection, its last_valid form, the discrete validator, swept-volume caches
  motion_planning::VoxelBackboneMotionValidator mv(vc);
  motion_planning::VoxelBackboneDiscreteMotionValidator dmv(vc);
  const std::vector<std::vector<double>> ea = {{0, 0, 0}, {8, 3, 1}, {1.0, 2.0, 0.5}}, eb = {{2.5, 9.0, 4.0}, {8.5, 3.5, 1.2}, {1.5, 2.2, 0.4}};
  std::vector<double> fa, fb;
  for (size_t i = 0; i < ea.size(); i++) {
    std::pair<std::vector<double>, double> lv, dlv;
    const bool ok = mv.checkMotion(ea[i], eb[i]), ok2 = mv.checkMotion(ea[i], eb[i], lv), okd = dmv.checkMotion(ea[i], eb[i], dlv);
    std::printf("edge %d %d %.17g %.17g %d %.17g\n", (int)ok, (int)ok2, lv.second, lv.first[0], (int)okd, dlv.second);
    fa.insert(fa.end(), ea[i].begin(), ea[i].end()); fb.insert(fb.end(), eb[i].begin(), eb[i].end());
  }
  {
    // the same three edges as index pairs into one vertex array
    std::vector<double> verts; std::vector<int32_t> eidx;
    for (size_t i = 0; i < ea.size(); i++) {
      verts.insert(verts.end(), ea[i].begin(), ea[i].end()); verts.insert(verts.end(), eb[i].begin(), eb[i].end());
      eidx.push_back((int32_t)(2 * i)); eidx.push_back((int32_t)(2 * i + 1));
    }
    std::vector<int32_t> infk;
    auto iv = mv.checkMotionIndexed(verts, 2 * ea.size(), eidx, &infk);
    auto pv = mv.checkMotionBatch(fa, fb, ea.size());
    for (size_t i = 0; i < iv.size(); i++) std::printf("indexed %d %d %d\n", (int)iv[i], (int)pv[i], infk[i]);
  }
  std::vector<int32_t> nfk;
  auto ev = dmv.checkMotionBatch(fa, fb, ea.size(), &nfk);
  for (size_t i = 0; i < ev.size(); i++) std::printf("dbatch %d %d\n", (int)ev[i], nfk[i]);
  auto ec = mv.voxelizeBatch(fa, fb, ea.size());
  auto vcaches = motion_planning::voxelize_states(vc, flat, states.size());
  auto ehit = motion_planning::caches_collide(vc, ec), vhit = motion_planning::caches_collide(vc, vcaches);
  for (size_t i = 0; i < ec.items(); i++)
    std::printf("ecache %d %lld %d\n", (int)ec.usable[i], (long long)(ec.offsets[i + 1] - ec.offsets[i]), (int)ehit[i]);
  for (size_t i = 0; i < vcaches.items(); i++)
    std::printf("vcache %d %lld %d\n", (int)vcaches.usable[i], (long long)(vcaches.offsets[i + 1] - vcaches.offsets[i]), (int)vhit[i]);
  // obstacle edits on the device + k nearest states
  {
    vc.add_spheres({0.05, 0.05, 0.1, 0.02, -0.1, 0.0, 0.05, 0.01});
    vc.dilate_sphere(0.004);
    vc.remove_interior();
    collision::VoxelOctree now(256);
    vc.obstacles(now);
    size_t cells = 0;
    for (uint64_t b : now.blocks()) cells += (size_t)__builtin_popcountll(b);
    std::printf("edited %zu %d %d\n", cells, (int)now.cell(166, 166, 179), (int)now.cell(128, 128, 128));
    std::vector<int32_t> idx; std::vector<double> dist;
    vc.nearest_k(flat, states.size(), 2, idx, dist);
    for (size_t i = 0; i < states.size(); i++) std::printf("knn %d %d %.17g\n", idx[2 * i], idx[2 * i + 1], dist[2 * i + 1]);
  }
  // a small cached roadmap through the shim: the four states as vertices, a ring of edges, queries in the slab environment
  {
    tendon::TendonRobot robot3 = robot;
    motion_planning::VoxelBackboneValidityChecker vc3(robot3, env, vox);
    motion_planning::VoxelBackboneMotionValidator mv3(vc3);
    const std::vector<int32_t> redges = {0, 1, 1, 2, 2, 3, 3, 0, 0, 2};
    auto rvc = motion_planning::voxelize_states(vc3, flat, states.size());
    auto rec = mv3.voxelizeIndexed(flat, states.size(), redges);
    for (size_t i = 0; i < rec.items(); i++) std::printf("iecache %d %lld\n", (int)rec.usable[i], (long long)(rec.offsets[i + 1] - rec.offsets[i]));
    // connectVertices + voxelizeEdge in one traversal: checkMotion's verdicts, sets for the accepted edges only
    auto con = mv3.voxelizeIndexed(flat, states.size(), redges, /*validate=*/true);
    for (size_t i = 0; i < con.items(); i++) std::printf("connect %d %lld\n", (int)con.usable[i], (long long)(con.offsets[i + 1] - con.offsets[i]));
    motion_planning::VoxelCachedLazyPRM prm(vc3, flat, states.size(), redges);
    prm.setCaches(rvc, rec);
    prm.prepare(2);                                       // landmark bounds: the answers below do not depend on them
    auto sol = prm.solveWithRoadmap({0, 0, 2}, {2, 0, 3});
    for (size_t q = 0; q < sol.status.size(); q++) {
      std::printf("query %d %.17g", sol.status[q], sol.cost[q]);
      for (int32_t v : sol.paths[q]) std::printf(" %d", v);
      std::printf("\n");
    }
    {
      const auto ss = prm.searchStats();                  // three queries: the host threads' round (the kernel takes rounds of 512 or more)
      if (ss.on_device != 0 || ss.handed_back != 0 || ss.expanded_on_host < 0 || ss.answered_by_components != 0) { std::printf("searchStats wrong\n"); return 3; }
    }
    auto inv = prm.revalidate();
    std::printf("revalidate %lld %lld\n", (long long)inv.first, (long long)inv.second);
    vc3.add_capsules({0.0, 0.0, 0.3, 0.05, 0.0, 0.3, 0.01});
  }
  // the sphere-swept checker on its own copy of the robot, same obstacle slab
  {
    tendon::TendonRobot robot2 = robot;                  // a copy has its own GPU context
    motion_planning::VoxelValidityChecker sv(robot2, env, vox);
    auto v2 = sv.isValidBatch(flat, states.size());
    for (size_t i = 0; i < v2.size(); i++) std::printf("spheres %d\n", (int)v2[i]);
  }
  try {
    collision::VoxelOctree fine(512);
    fine.set_xlim(-0.25, 0.25); fine.set_ylim(-0.25, 0.25); fine.set_zlim(-0.25, 0.25);
    motion_planning::VoxelBackboneValidityChecker bad(robot, env, fine);
  } catch (const std::invalid_argument &e) { std::printf("invalid_argument: %s\n", e.what()); }
  return 0;
}
