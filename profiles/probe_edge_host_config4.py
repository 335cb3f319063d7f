"""Host-side phases of tr_validate_edges_indexed at config 4's size (6 x 10^5 vertices, 3.5 M edges; TENDON_HIP_EDGE_TIMING=1)."""
import importlib, os, sys, time
os.environ["TENDON_HIP_EDGE_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W, D = irt.workloads, irt.distributed
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
checker = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
eng = checker.engine
rb = irt.RoadmapBuilder(checker, irt.VoxelBackboneMotionValidator(checker), seed=3)
M, k, seed = 1 << 20, 10, 3
box = D.sampling_box(robot)
vv = D.ShardedVertexValidator(robot, seed=seed, device="cuda", box=box, validate_candidates=D.device_candidate_validator(eng, seed, box))
mask = vv.run(M, rank=0, world_size=1, keep_on_device=True)
verts = D.gather_valid_vertices_dev(eng, seed, M, mask, box=box)[0].cpu().numpy()
edges = rb.knn_edges_gpu(verts, k + 1)
for i in range(3):
    sys.stderr.write("--- call %d\n" % i); sys.stderr.flush()
    t0 = time.perf_counter()
    ev, nfk = rb.validate_edges(verts, edges)
    dt = time.perf_counter() - t0
    sys.stderr.write("python wall %.3f ms, %d edges, %.3g edges/s, %d FK samples\n" % (1e3 * dt, len(edges), len(edges) / dt, int(nfk.sum())))
