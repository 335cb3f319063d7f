"""What the last third of create_roadmap costs: VoxelCachedLazyPRM(...) (tr_roadmap_create), set_caches (device lists), prepare(16)."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
states, _ = rb.sample_valid_vertices(100000)
cand = rb.knn_edges_gpu(states, 11)
chk.engine.reserve_edges(len(cand))
edges, ec = rb.connect(states, cand, device=True)
vc = rb.vertex_caches(states, device=True)
for it in range(4):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    prm = irt.VoxelCachedLazyPRM(chk, states, edges); t.append(time.perf_counter())
    prm.set_caches(vc, ec); torch.cuda.synchronize(); t.append(time.perf_counter())
    prm.prepare(16); torch.cuda.synchronize(); t.append(time.perf_counter())
    print("iteration %d: VoxelCachedLazyPRM() %.2f ms, set_caches %.2f ms, prepare(16) %.2f ms" % ((it,) + tuple(1e3 * np.diff(t))), flush=True)
    del prm
for it in range(3):
    t0 = time.perf_counter(); rb.create_roadmap(100000, k=10, device=True); dt = time.perf_counter() - t0
    print("create_roadmap %.2f ms:" % (1e3 * dt), {k: round(1e3 * v["seconds"], 2) for k, v in rb.timing.items() if "seconds" in v}, flush=True)
