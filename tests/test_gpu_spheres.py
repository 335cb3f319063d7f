"""VoxelValidityChecker (motion-planning/VoxelValidityChecker.h:18-26): the robot voxelised as a sphere of its
radius at every backbone point, against a raw (un-dilated) environment -- verdicts and flags identical to the
oracle's add_sphere + collides restatement."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _env(irt, seed, n, N=256, lim=0.25, rmin=0.004, rmax=0.012):
    rng = np.random.default_rng(seed)
    vox = irt.VoxelOctree(N)
    vox.set_xlim(-lim, lim); vox.set_ylim(-lim, lim); vox.set_zlim(-lim, lim)
    k = 0
    while k < n:
        c = rng.uniform(-0.2, 0.2, 3)
        if np.hypot(c[0], c[1]) < 0.03 and -0.02 < c[2] < 0.08:
            continue                                   # keep the base of the robot free
        vox.add_sphere(c, rng.uniform(rmin, rmax))
        k += 1
    return vox


@pytest.mark.parametrize("rotated", [False, True])
def test_sphere_checker_matches_oracle(irt, orc, helpers, rotated):
    W = irt.workloads
    robot = W.robot_config2()
    vox = _env(irt, 21, 90)
    env = irt.VoxelEnvironment()
    if rotated:
        a = 0.3
        env.inv_rotation = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    chk = irt.VoxelValidityChecker(robot, env, vox)
    states = W.random_states(robot, 1200, seed=5, tau_max=14.0)
    got = chk.is_valid_detail(states)
    orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
    want = [orc.is_valid_state_spheres(orb, og, s, env.inv_rotation) for s in states]
    assert np.array_equal(got["valid"], [w[0] for w in want])
    assert np.array_equal(got["flags"] & 15, [w[2] for w in want])
    v = np.array([w[0] for w in want])
    assert 0.15 < v.mean() < 0.9
    # the backbone checker on the same raw environment accepts strictly more
    back = irt.VoxelBackboneValidityChecker(robot, env, vox).is_valid(states)
    assert (back | ~v).all() and (back & ~v).sum() > 20


def test_sphere_checker_domain_edges_and_small_grid(irt, orc, helpers):
    """Spheres reaching over the grid boundary (block range clamped), backbone points outside the domain
    (no add_point cell, sphere may still reach in), coarse grid with voxels larger than the radius."""
    W = irt.workloads
    robot = W.robot_config1()                           # dL = 5 mm
    for N, lim in ((64, 0.12), (32, 0.25)):
        vox = _env(irt, 33, 60, N=N, lim=lim, rmin=0.01, rmax=0.03)
        chk = irt.VoxelValidityChecker(robot, irt.VoxelEnvironment(), vox)
        states = W.random_states(robot, 600, seed=6)
        got = chk.is_valid_detail(states)
        orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
        want = [orc.is_valid_state_spheres(orb, og, s) for s in states]
        assert np.array_equal(got["valid"], [w[0] for w in want]), (N, lim)
        assert np.array_equal(got["flags"] & 15, [w[2] for w in want])
        assert 0.02 < np.mean([w[0] for w in want]) < 0.98


def test_sphere_checker_follows_grid_edits(irt, orc, helpers):
    W = irt.workloads
    robot = W.robot_config2()
    vox = _env(irt, 22, 30)
    chk = irt.VoxelValidityChecker(robot, irt.VoxelEnvironment(), vox)
    states = W.random_states(robot, 500, seed=7, tau_max=14.0)
    before = chk.is_valid(states)
    extra = np.array([[0.05, 0.0, 0.15, 0.02], [-0.04, 0.06, 0.12, 0.015]])
    chk.add_spheres(extra)                              # the dilated lookup grid must follow
    after = chk.is_valid(states)
    og = helpers.oracle_grid(orc, vox)
    for row in extra:
        og.add_sphere(row[:3], row[3])
    orb = helpers.oracle_robot(orc, robot)
    want = np.array([orc.is_valid_state_spheres(orb, og, s)[0] for s in states])
    assert np.array_equal(after, want) and (before & ~after).sum() > 5


def test_sphere_checker_with_retraction(irt, orc, helpers):
    """per-configuration point counts, rows aligned at the tip (fk_retract_kernel.hpp)"""
    W = irt.workloads
    robot = W.robot_config2()
    robot.enable_retraction = True
    vox = _env(irt, 23, 200, rmin=0.006, rmax=0.016)
    chk = irt.VoxelValidityChecker(robot, irt.VoxelEnvironment(), vox)
    states = W.random_states(robot, 600, seed=8, tau_max=14.0)
    states[:, -1] = np.random.default_rng(9).uniform(0.0, 0.07, len(states))
    states[:4, -1] = [0.0, robot.specs.L, robot.specs.L - robot.specs.dL / 4, 0.1]
    got = chk.is_valid_detail(states)
    orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
    want = [orc.is_valid_state_spheres(orb, og, s) for s in states]
    assert np.array_equal(got["valid"], [w[0] for w in want])
    assert np.array_equal(got["flags"] & 15, [w[2] for w in want])
    assert 0.1 < np.mean([w[0] for w in want]) < 0.97
