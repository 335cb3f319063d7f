"""Config 4 on one GPU, device-resident (vertices compacted in HBM -> tr_knn_edges_dev -> tr_validate_edges_indexed_dev): time of
each call, fastest of five, and the rocprofv3-free split of the neighbour phase by TENDON_HIP profile slots."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W, D = irt.workloads, irt.distributed
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
eng = chk.engine
mv = irt.VoxelBackboneMotionValidator(chk)
M, k, seed = 1 << 20, 10, 3
box = D.sampling_box(robot)
vv = D.ShardedVertexValidator(robot, seed=seed, device="cuda", box=box, validate_candidates=D.device_candidate_validator(eng, seed, box))
best = None
d_edges = d_bits = None
for it in range(6):
    torch.cuda.synchronize(); t = [time.perf_counter()]
    mask = vv.run(M, rank=0, world_size=1, keep_on_device=True); torch.cuda.synchronize(); t.append(time.perf_counter())
    d_verts = D.gather_valid_vertices_dev(eng, seed, M, mask, box=box)[0]; torch.cuda.synchronize(); t.append(time.perf_counter())
    nv = d_verts.shape[0]
    if d_edges is None:
        d_edges = torch.empty((nv * (k + 1) * 9 // 8, 2), dtype=torch.int32, device="cuda")
        d_bits = torch.empty((d_edges.shape[0] + 63) // 64, dtype=torch.int64, device="cuda")
    ne = eng.knn_edges_dev(d_verts, nv, k + 1, d_edges); torch.cuda.synchronize(); t.append(time.perf_counter())
    eng.validate_edges_indexed_dev(d_verts, nv, d_edges, ne, d_bits, None, mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change)
    torch.cuda.synchronize(); t.append(time.perf_counter())
    d = np.diff(t)
    if it and (best is None or d.sum() < best.sum()):
        best = d
print("vertices %d edges %d: vertex phase %.2f ms, compaction %.2f ms, tr_knn_edges_dev %.2f ms, tr_validate_edges_indexed_dev %.2f ms, sum %.2f ms"
      % (nv, ne, *(1e3 * best), 1e3 * best.sum()))
