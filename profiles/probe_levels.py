"""Per-launch durations of the edge bisection (run under rocprofv3 --kernel-trace): indexed validation of config 3's roadmap."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3(); vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
states, tips = rb.sample_valid_vertices(100000, batch=1 << 17)
edges = rb.knn_edges_gpu(states, 11)
chk.engine.reserve_edges(len(edges))
rb.validate_edges(states, edges[:1000])
print("MARK begin", flush=True)
t0 = time.perf_counter(); valid, nfk = rb.validate_edges(states, edges); print("indexed edges wall ms", 1e3 * (time.perf_counter() - t0), len(edges))
