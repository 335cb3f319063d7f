#!/usr/bin/env python3
"""Expansions per search of config 5's query loop on the host (TENDON_HIP_SEARCH_HIST): found against not found, per round."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["TENDON_HIP_SEARCH"] = "host"
os.environ["TENDON_HIP_SEARCH_HIST"] = "1"
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
new_vox, _ = W.reach_environment(seed=7, n_spheres=72)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
states, _ = rb.sample_valid_vertices(100000, batch=1 << 17)
edges = rb.knn_edges_gpu(states, 11)
valid, _ = rb.validate_edges(states, edges)
e_ok = edges[valid]
prm = irt.VoxelCachedLazyPRM(chk, states, e_ok)
prm.set_caches(rb.vertex_caches(states), rb.edge_caches(states, e_ok))
prm.set_obstacles(new_vox)
pairs = np.random.default_rng(17).integers(0, len(states), size=(10000, 2))
for nl in (16, 0):
    prm.prepare(nl)
    for form in ("eager", "lazy"):
        print("==== landmarks", nl, form, flush=True); sys.stderr.flush()
        prm.clearValidity()
        if form == "eager":
            prm.revalidate()
        prm.solveWithRoadmap(pairs[:, 0], pairs[:, 1])
        sys.stderr.flush()
