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


def _edge_pairs(irt, robot, n, seed, step):
    rng = np.random.default_rng(seed)
    a = irt.workloads.random_states(robot, n, seed=seed, tau_max=12.0)
    d = rng.normal(size=a.shape)
    d *= step / np.linalg.norm(d, axis=1, keepdims=True)
    b = np.clip(a + d, 0.0, 19.9)
    return a, b


def test_motion_validators_next_to_the_sphere_checker(irt, orc, helpers):
    """Problem.h:175-210 installs VoxelValidityChecker next to VoxelBackboneMotionValidator: checkMotion(s1, s2)
    keeps sweeping the backbone against the same voxels (AbstractVoxelMotionValidator.h:143-151), while
    checkMotion(s1, s2, last_valid) asks the installed checker about every sample (`_vc->collides`,
    VoxelBackboneMotionValidator.cpp:83-91) -- the sphere-swept robot.  Both against the oracle's restatement."""
    robot = irt.workloads.robot_config2()
    vox = _env(irt, 21, 90)
    chk = irt.VoxelValidityChecker(robot, irt.VoxelEnvironment(), vox)
    orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
    a, b = _edge_pairs(irt, robot, 96, seed=9, step=1.2)
    mv = irt.VoxelBackboneMotionValidator(chk)
    two = mv.check_motion_detail(a, b)
    valid3, lvt = mv.check_motion_last_valid(a, b)
    dm = irt.VoxelBackboneDiscreteMotionValidator(chk)
    d2 = dm.check_motion_detail(a, b, last_valid=False)
    d3 = dm.check_motion_detail(a, b, last_valid=True)
    n_diff = 0
    for i in range(len(a)):
        w2 = orc.check_motion(orb, og, a[i], b[i])
        assert two["valid"][i] == w2["valid"] and (not w2["valid"] or two["n_fk"][i] == w2["n_fk"]), i
        w3 = orc.check_motion_until_invalid(orb, og, a[i], b[i], vc_spheres=True)
        assert valid3[i] == w3["is_fully_valid"] and lvt[i] == w3["last_valid_t"], (i, lvt[i], w3)
        wd2 = orc.check_motion_discrete(orb, og, a[i], b[i], until_invalid=False)
        wd3 = orc.check_motion_discrete(orb, og, a[i], b[i], until_invalid=True, vc_spheres=True)
        assert d2["valid"][i] == wd2["valid"], i
        assert d3["valid"][i] == wd3["is_fully_valid"] and d3["last_valid_t"][i] == wd3["last_valid_t"] and d3["n_fk"][i] == wd3["n_fk"], (i, wd3)
        n_diff += int(w2["valid"] != w3["is_fully_valid"])
    assert n_diff > 3 and 0 < valid3.sum() < len(a)       # the two forms really differ under this checker


def test_two_checkers_on_one_robot_keep_their_own_environment(irt, orc, helpers):
    """Each checker owns its obstacle set (AbstractVoxelValidityChecker.h:63-64): building a second checker on the
    same robot, with other obstacles and another rotation, must not change what the first one tests against."""
    W = irt.workloads
    robot = W.robot_config2()
    voxA, _ = W.reach_environment(seed=7, n_spheres=48)
    voxB, _ = W.reach_environment(seed=8, n_spheres=80)
    envB = irt.VoxelEnvironment()
    envB.inv_rotation = np.array([[0.0, -1.0, 0], [1.0, 0, 0], [0, 0, 1.0]])
    states = W.random_states(robot, 2000, seed=77, tau_max=14.0)
    chkA = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), voxA)
    before = chkA.is_valid(states)
    mvA = irt.VoxelBackboneMotionValidator(chkA)
    a, b = _edge_pairs(irt, robot, 200, seed=3, step=1.0)
    eA = mvA.check_motion(a, b)
    chkB = irt.VoxelBackboneValidityChecker(robot, envB, voxB)
    robot.r = 0.02                                            # a later edit of the robot reaches only checkers built after it
    gotB = chkB.is_valid(states)
    assert chkA.engine is not chkB.engine
    assert np.array_equal(chkA.is_valid(states), before) and np.array_equal(mvA.check_motion(a, b), eA)
    robot.r = 0.015
    orb = helpers.oracle_robot(orc, robot, lib="omp")
    wantA, _, _ = orc.validate_batch(orb, helpers.oracle_grid(orc, voxA), states, nthreads=0, lib=orc.omp_lib())
    wantB, _, _ = orc.validate_batch(orb, helpers.oracle_grid(orc, voxB), states, envB.inv_rotation, nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(before, wantA) and np.array_equal(gotB, wantB) and (wantA != wantB).sum() > 50


def _with_env(env, fn):
    import os
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_sphere_checker_same_bits_in_every_schedule(irt):
    """The sphere-swept checker through fk_verdict<.., SPH> (no backbone stored; points classified by the distance field as
    they are produced, the shell between queued for the exact scan) against K1 + K2 + K8 on stored points -- separate
    launches and fused -- where the rarer branches are busy: self collisions decided by the fallback pass (which takes the
    sphere test's answer from the flags), length limits, non-converged solves, a rotated environment, a rotating robot,
    points outside the voxel domain, a coarse grid, and the debug switches."""
    W = irt.workloads
    thin = W.robot_config1()
    thin.specs.dL = 0.2 / 128
    thin.r = 0.01
    for t in thin.tendons:
        t.max_tension, t.min_length, t.max_length = 100.0, -1.0, 0.08
    hard = W.robot_config2()
    for t in hard.tendons:
        t.max_tension, t.max_length = 60.0, 0.02
    spin = W.robot_config2()
    spin.enable_rotation = True
    a = 0.4
    rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    big = _env(irt, 21, 90)
    small = _env(irt, 33, 50, N=64, lim=0.12, rmin=0.01, rmax=0.03)       # the robot reaches out of it; cells of 3.75 mm
    cases = [(thin, 100.0, None, big, 6000), (hard, 45.0, None, big, 4000), (spin, 12.0, rot, big, 5000),
             (W.robot_config3(), 20.0, rot, big, 4000), (W.robot_config2(), 14.0, None, small, 5000)]
    seen = np.zeros(32, int)
    for robot, tau_max, inv_rot, vox, n in cases:
        env = irt.VoxelEnvironment()
        if inv_rot is not None:
            env.inv_rotation = inv_rot
        states = W.random_states(robot, n + 21, seed=19, tau_max=tau_max)

        def run(debug=0, detail=True):
            chk = irt.VoxelValidityChecker(robot, env, vox)
            chk.engine.set_debug(debug)
            return chk.is_valid_detail(states) if detail else dict(valid=chk.is_valid(states))

        want = _with_env({"TENDON_HIP_FUSED": "0"}, run)
        fused = _with_env({"TENDON_HIP_FUSED": "1"}, run)
        for k in ("valid", "flags"):
            assert np.array_equal(fused[k], want[k]), k
        for debug in (0, 2, 3):
            got = _with_env({"TENDON_HIP_FB_CAP": "64"}, lambda: run(debug))
            for k in ("valid", "flags"):
                assert np.array_equal(got[k], want[k]), (k, debug, np.flatnonzero(got[k] != want[k])[:8], got[k][got[k] != want[k]][:8], want[k][got[k] != want[k]][:8])
            ok = want["flags"] & 1 > 0
            assert np.abs(got["tips"][ok] - want["tips"][ok]).max() <= 1e-12
        assert np.array_equal(run(0, detail=False)["valid"], want["valid"])
        seen += np.bincount(want["flags"], minlength=32)
    print("flag histogram", seen)
    assert seen[15] > 0 and seen[7] > 0 and seen[3] > 0 and seen[1] > 0 and seen[0] > 0
