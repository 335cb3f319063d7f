"""Config 4's neighbour and edge phases on one GPU at 2^20 candidates (~600 k vertices): host wall per step and, under
rocprofv3 --kernel-trace --stats, the kernels' share of it."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W, D = irt.workloads, irt.distributed
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
mv = irt.VoxelBackboneMotionValidator(chk)
eng = chk.engine
rb = irt.RoadmapBuilder(chk, mv, seed=3)
M = 1 << 20
box = D.sampling_box(robot)
vv = D.ShardedVertexValidator(robot, seed=3, device="cuda", box=box, validate_candidates=D.device_candidate_validator(eng, 3, box))
for it in range(3):
    torch.cuda.synchronize(); t = [time.perf_counter()]
    mask = vv.run(M, keep_on_device=True); torch.cuda.synchronize(); t.append(time.perf_counter())
    verts_dev, _ = D.gather_valid_vertices_dev(eng, 3, M, mask, box=box); verts = verts_dev.cpu().numpy(); t.append(time.perf_counter())
    idx, dist = eng.knn(verts, 11); t.append(time.perf_counter())
    e1 = eng.edges_from_knn(idx); t.append(time.perf_counter())
    e2 = eng.knn_edges(verts, 11); t.append(time.perf_counter())
    ev, nf = rb.validate_edges(verts, e2); t.append(time.perf_counter())
    d = 1e3 * np.diff(t)
    print("it %d: vertices %.1f | compact+download %.1f | knn (rows to host) %.1f | edges_from_knn (host table) %.1f | knn_edges (device table) %.1f | "
          "validate %d edges %.1f ms (%d FK samples)" % (it, d[0], d[1], d[2], d[3], d[4], len(e2), d[5], int(nf.sum())), flush=True)
