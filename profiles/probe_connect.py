#!/usr/bin/env python3
"""Edge phase of a roadmap build: tr_validate_edges_indexed, tr_voxelize_edges_indexed and tr_connect_edges_indexed side
by side, wall time and per-slot kernel times."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
states, _ = rb.sample_valid_vertices(100000, batch=1 << 17)
edges = rb.knn_edges_gpu(states, 11)
chk.engine.reserve_edges(len(edges))
eng = chk.engine
for name, fn in (("validate", lambda: rb.validate_edges(states, edges)),
                 ("voxelize (device lists)", lambda: rb.edge_caches(states, edges, device=True)),
                 ("connect (device lists)", lambda: rb.connect(states, edges, device=True))):
    fn()
    for rep in range(2):
        eng.profile_begin()
        t0 = time.perf_counter(); fn(); t1 = time.perf_counter()
        p = eng.profile_read(); eng.profile_end()
        print("%s rep %d: %.1f ms, kernels: %s" % (name, rep, 1e3 * (t1 - t0), {k: (round(v["total_ms"], 2), v["launches"]) for k, v in p.items() if v["launches"]}), flush=True)
