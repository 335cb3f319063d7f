"""The synthetic, seeded workloads of BASELINE.json `configs`: robots,
obstacle environments and configuration batches.  The reference ships no robot / problem files,
so these definitions are the fixtures; tests and bench.py share them.
"""
import numpy as np

from .collision import VoxelOctree
from .tendon import BackboneSpecs, TendonRobot, TendonSpecs

PI = np.pi


def robot_config1():
    """3 straight tendons, default backbone (L=0.2, dL=0.005 -> 41 points)."""
    tendons = [TendonSpecs(C=[2 * PI * k / 3], D=[0.01]) for k in range(3)]
    return TendonRobot(tendons=tendons, specs=BackboneSpecs())


def robot_config2():
    """3 helical tendons C=[2 pi k/3, 5], dL = L/128 -> 129 points (dL <= voxel edge of a 256^3 grid
    over [-0.25, 0.25]^3, as VoxelBackboneValidityChecker.h:37-45 requires)."""
    tendons = [TendonSpecs(C=[2 * PI * k / 3, 5.0], D=[0.01]) for k in range(3)]
    return TendonRobot(tendons=tendons, specs=BackboneSpecs(dL=0.2 / 128))


def robot_config3():
    """4 tendons with quadratic routing in angle and linear in radius, padded to N_a = N_m = 3."""
    c1 = [3.0, -2.0, 4.0, -5.0]
    c2 = [10.0, 15.0, -12.0, 8.0]
    d1 = [-0.01, 0.005, 0.0, -0.005]
    tendons = [TendonSpecs(C=[PI * k / 2, c1[k], c2[k]], D=[0.01, d1[k], 0.0]) for k in range(4)]
    return TendonRobot(tendons=tendons, specs=BackboneSpecs(dL=0.2 / 128))


def sphere_environment(seed=7, n_spheres=64, radius=0.02, N=256, half=0.25, keepout=0.05):
    """256^3 grid over [-half, half]^3 with seeded spheres (voxel centre inside, add_sphere
    semantics), none within `keepout` of the home backbone segment (0,0,0)-(0,0,0.2)."""
    rng = np.random.default_rng(seed)
    vox = VoxelOctree(N)
    vox.set_xlim(-half, half); vox.set_ylim(-half, half); vox.set_zlim(-half, half)
    centres = []
    while len(centres) < n_spheres:
        c = rng.uniform(-half, half, 3)
        # distance to the segment x = y = 0, z in [0, 0.2]
        zc = min(max(c[2], 0.0), 0.2)
        if np.sqrt(c[0] ** 2 + c[1] ** 2 + (c[2] - zc) ** 2) < keepout + radius:
            continue
        centres.append(c)
    for c in centres:
        vox.add_sphere(c, radius)
    return vox, np.array(centres)


def reach_environment(seed=7, n_spheres=64, radius=0.02, N=256, half=0.25, keepout=0.03, reach=0.21):
    """Like sphere_environment but with every sphere inside the robot's reach ball, so that a
    useful fraction of random configurations collides."""
    rng = np.random.default_rng(seed)
    vox = VoxelOctree(N)
    vox.set_xlim(-half, half); vox.set_ylim(-half, half); vox.set_zlim(-half, half)
    centres = []
    while len(centres) < n_spheres:
        c = rng.uniform(-reach, reach, 3)
        if np.linalg.norm(c) > reach or c[2] < -0.02:
            continue
        zc = min(max(c[2], 0.0), 0.2)
        if np.sqrt(c[0] ** 2 + c[1] ** 2 + (c[2] - zc) ** 2) < keepout + radius:
            continue
        centres.append(c)
    for c in centres:
        vox.add_sphere(c, radius)
    return vox, np.array(centres)


def random_states(robot, n, seed, tau_max=None):
    """tau ~ U[0, tau_max) per tendon (default: the tendon's max_tension), rotation ~ U[-pi, pi)."""
    rng = np.random.default_rng(seed)
    cols = []
    for t in robot.tendons:
        cols.append(rng.uniform(0.0, t.max_tension if tau_max is None else tau_max, n))
    if robot.enable_rotation:
        cols.append(rng.uniform(-PI, PI, n))
    if robot.enable_retraction:
        cols.append(rng.uniform(0.0, robot.specs.L, n))
    return np.ascontiguousarray(np.stack(cols, axis=1))
