"""Voxel-cache construction alone (tr_voxelize_batch / tr_voxelize_edges_indexed): first call (allocations) and steady state."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
V = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
states, _ = rb.sample_valid_vertices(V, batch=1 << 17)
edges = rb.knn_edges_gpu(states, 11)
chk.engine.reserve_edges(len(edges))
valid, nfk = rb.validate_edges(states, edges)
e_ok = edges[valid]
for rep in range(3):
    chk.engine.profile_begin()
    t0 = time.perf_counter(); vc = chk.engine.voxelize_batch(states); t1 = time.perf_counter()
    p = chk.engine.profile_read(); chk.engine.profile_end()
    print("vertex caches rep %d: %.1f ms (%.3g/s), kernels: %s" % (rep, 1e3 * (t1 - t0), V / (t1 - t0), {k: round(v["total_ms"], 2) for k, v in p.items() if v["launches"]}))
for rep in range(2):
    chk.engine.profile_begin()
    t0 = time.perf_counter(); ec = chk.engine.voxelize_edges_indexed(states, e_ok); t1 = time.perf_counter()
    p = chk.engine.profile_read(); chk.engine.profile_end()
    print("edge caches rep %d: %.1f ms (%.3g/s), kernels: %s" % (rep, 1e3 * (t1 - t0), len(e_ok) / (t1 - t0), {k: round(v["total_ms"], 2) for k, v in p.items() if v["launches"]}))
