"""Host-side phases of tr_validate_edges_indexed (TENDON_HIP_EDGE_TIMING=1 prints them to stderr): config 3, 100 k vertices, 10-NN."""
import importlib, os, sys, time
os.environ["TENDON_HIP_EDGE_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
states, _ = rb.sample_valid_vertices(100000)
edges = rb.knn_edges_gpu(states, 11)
chk.engine.reserve_edges(len(edges))
for i in range(4):
    sys.stderr.write("--- call %d\n" % i); sys.stderr.flush()
    t0 = time.perf_counter()
    v, nf = rb.validate_edges(states, edges)
    dt = time.perf_counter() - t0
    sys.stderr.write("python wall %.3f ms, %d edges, %.3g edges/s, %d FK samples\n" % (1e3 * dt, len(edges), len(edges) / dt, int(nf.sum())))
