#!/usr/bin/env python3
"""Generates the golden fixtures under tests/golden/ with the CPU oracle (strict build,
-O2 -ffp-contract=off).  The reference ships no tests, fixtures or robot files and cannot be built
here, so these vectors pin the ORACLE (regression) and give the GPU tests fixed
expected outputs; they are not outputs of the reference itself ("parity unpinned").

    python tests/golden/make_golden.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import oracle as orc  # noqa: E402

irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads


def oracle_robot(robot):
    s = robot.specs
    return orc.Robot([t.C for t in robot.tendons], [t.D for t in robot.tendons], r=robot.r, L=s.L, dL=s.dL,
                     ro=s.ro, ri=s.ri, E=s.E, nu=s.nu, max_tension=[t.max_tension for t in robot.tendons],
                     min_length=[t.min_length for t in robot.tendons], max_length=[t.max_length for t in robot.tendons],
                     enable_rotation=robot.enable_rotation, enable_retraction=robot.enable_retraction,
                     residual_threshold=robot.residual_threshold)


def main():
    # config 1: 3-tendon linear-routed robot, 1000 random configs, FK only
    r1 = W.robot_config1()
    s1 = W.random_states(r1, 1000, seed=42)
    fk = oracle_robot(r1).fk_batch(s1)
    np.savez_compressed(os.path.join(HERE, "config1_fk.npz"), states=s1, tips=fk["p"][:, -1], L=fk["L"], L_i=fk["L_i"],
                        converged=fk["converged"], p_first8=fk["p"][:8])
    # config 2: helical robot + 256^3 sphere environment, validity verdicts
    r2 = W.robot_config2()
    vox, centres = W.reach_environment(seed=7, n_spheres=64)
    og = orc.Grid(256, vox.limits())
    og.blocks()[...] = vox.blocks
    s2 = W.random_states(r2, 4096, seed=11, tau_max=20.0)
    valid, tips, _ = orc.validate_batch(oracle_robot(r2), og, s2)
    flags = np.array([orc.is_valid_state(oracle_robot(r2), og, s)[2] for s in s2[:512]], dtype=np.uint8)
    ids, masks = vox.to_sparse()
    np.savez_compressed(os.path.join(HERE, "config2_validity.npz"), states=s2, valid=valid, tips=tips, flags512=flags,
                        sphere_centres=centres, grid_ids=ids, grid_masks=masks)
    # config 3 robot: FK of the 4-tendon quadratic-routed robot
    r3 = W.robot_config3()
    s3 = W.random_states(r3, 256, seed=44)
    fk3 = oracle_robot(r3).fk_batch(s3)
    np.savez_compressed(os.path.join(HERE, "config3_fk.npz"), states=s3, tips=fk3["p"][:, -1], L_i=fk3["L_i"],
                        converged=fk3["converged"], home_L_i=oracle_robot(r3).home_shape()["L_i"])
    # add_line cell sequences (documenting the two quirks of VoxelOctree::add_line)
    g = orc.Grid(16, (0, 1.6, 0, 1.6, 0, 1.6))
    cases = {
        "axis_x": ([0.05, 0.05, 0.05], [0.45, 0.05, 0.05]),
        "diag": ([0.05, 0.05, 0.05], [0.35, 0.35, 0.35]),
        "zero_len": ([0.55, 0.55, 0.55], [0.55, 0.55, 0.55]),
        "enter_from_outside": ([-0.25, 0.45, 0.45], [0.25, 0.45, 0.45]),
        "leave_domain": ([1.45, 1.45, 1.45], [1.85, 1.55, 1.45]),
        "miss": ([-0.5, -0.5, -0.5], [-0.1, -0.6, -0.2]),
        "negative_dir": ([0.85, 0.95, 0.25], [0.55, 0.75, 0.15]),
    }
    out = {}
    for name, (a, b) in cases.items():
        g.clear()
        g.add_line(a, b)
        out[name + "_a"], out[name + "_b"] = np.array(a), np.array(b)
        out[name + "_cells"] = np.array(g.cells(), dtype=np.int32).reshape(-1, 3)
    np.savez_compressed(os.path.join(HERE, "add_line_cells.npz"), **out)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
