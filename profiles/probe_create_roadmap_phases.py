"""createRoadmap phase by phase (the calls RoadmapBuilder.create_roadmap makes), config 3's robot; `full` adds rotation + retraction."""
import importlib, os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
robot.enable_rotation = robot.enable_retraction = len(sys.argv) > 1 and sys.argv[1] == "full"      # "full": rotation + retraction
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
for rep in range(3):
    t = [time.perf_counter()]
    states, tips = rb.sample_valid_vertices(100000, batch=1 << 17); t.append(time.perf_counter())
    cand = rb.knn_edges_gpu(states, 11); t.append(time.perf_counter())
    chk.engine.reserve_edges(len(cand)); t.append(time.perf_counter())
    edges, ec = rb.connect(states, cand, device=True); t.append(time.perf_counter())
    vc = rb.vertex_caches(states, device=True); t.append(time.perf_counter())
    prm = irt.VoxelCachedLazyPRM(chk, states, edges); t.append(time.perf_counter())
    prm.set_caches(vc, ec); t.append(time.perf_counter())
    prm.prepare(16); t.append(time.perf_counter())
    names = ["vertices", "knn", "reserve", "connect", "vertex_caches", "VoxelCachedLazyPRM()", "set_caches", "prepare(16 landmarks)"]
    print(rep, {n: round(1e3 * (b - a), 1) for n, a, b in zip(names, t[:-1], t[1:])}, "total %.1f" % (1e3 * (t[-1] - t[0])), flush=True)
