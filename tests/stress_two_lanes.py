#!/usr/bin/env python3
"""Opt-in stress of the multi-lane edge bisection (not collected by pytest): the same roadmap's edges validated over and over on two
and on four streams -- verdicts, FK counts and domain-error counts must be those of the first call and of the one-lane path every time (a race
between the lanes would show up as a changing result), for a tension-only, a rotating and a rotating + retracting robot, with
state batches and cached-set checks interleaved between the edge calls.

    python tests/stress_two_lanes.py [iterations]        # on the GPU box; exits non-zero on a difference
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    irt = importlib.import_module("interactive-rate-tendons_amd")
    W = irt.workloads
    bad = 0
    for rot, ret in ((False, False), (True, False), (True, True)):
        robot = W.robot_config3()
        robot.enable_rotation, robot.enable_retraction = rot, ret
        vox, _ = W.reach_environment(seed=7, n_spheres=64)
        res = {}
        for lanes in ("1", "2", "4"):
            os.environ["TENDON_HIP_EDGE_LANES"] = lanes
            chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
            rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
            states, _ = rb.sample_valid_vertices(30000, batch=1 << 16)
            edges = rb.knn_edges_gpu(states, 9)
            first = None
            for it in range(iters if lanes != "1" else 3):
                out = chk.engine.validate_edges_indexed(states, edges, rb.mv.min_tension_change, rb.mv.min_rotation_change, rb.mv.min_retraction_change)
                key = (out["valid"].tobytes(), out["n_fk"].tobytes(), out["n_domain_errors"])
                if first is None:
                    first = key
                elif key != first:
                    bad += 1
                    print("rotation %s retraction %s lanes %s: iteration %d differs from the first" % (rot, ret, lanes, it), flush=True)
                if it % 3 == 0:                     # other work of the context between the edge calls
                    chk.is_valid(states[: 20000 + 97 * it])
            res[lanes] = first
        same = res["1"] == res["2"] == res["4"]
        bad += not same
        print("rotation %s retraction %s: %d edges, %d two-lane and %d four-lane calls identical, equal to one lane: %s" % (rot, ret, len(edges), iters, iters, same), flush=True)
    print("differences:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
