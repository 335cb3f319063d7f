"""GPU parity of batched edge (motion) validation, tr_validate_edges, against the oracle's
depth-first restatement of VoxelEnvironment::voxelize_valid_backbone_motion + checkMotion."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _edges(robot, irt, n, seed, step):
    rng = np.random.default_rng(seed)
    a = irt.workloads.random_states(robot, n, seed=seed)
    d = rng.normal(size=a.shape)
    d *= (rng.uniform(0.05, step, n) / np.linalg.norm(d[:, :len(robot.tendons)], axis=1))[:, None]
    b = a + d
    nt = len(robot.tendons)
    b[:, :nt] = np.clip(b[:, :nt], 0.0, [t.max_tension for t in robot.tendons])
    if robot.enable_rotation:
        b[:, nt] = (b[:, nt] + np.pi) % (2 * np.pi) - np.pi
    return a, b


def _compare(irt, orc, helpers, robot, vox, a, b):
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    mv = irt.VoxelBackboneMotionValidator(chk)
    got = mv.check_motion_detail(a, b)
    orb = helpers.oracle_robot(orc, robot, lib="omp")
    og = helpers.oracle_grid(orc, vox)
    want, want_nfk, _ = orc.check_motion_batch(orb, og, a, b, nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(got["valid"], want), np.flatnonzero(got["valid"] != want)[:10]
    # for valid edges nothing is pruned, so the sample sets (and counts) coincide
    assert np.array_equal(got["n_fk"][want], want_nfk[want])
    assert (got["n_fk"] >= 2).all() and got["n_domain_errors"] == 0
    return got, want, want_nfk


def test_edges_config3_match_oracle(irt, orc, helpers):
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    a, b = _edges(robot, irt, 1500, seed=31, step=3.0)
    got, want, nfk = _compare(irt, orc, helpers, robot, vox, a, b)
    print("edges valid %.2f, FK samples/edge mean %.1f max %d" % (want.mean(), nfk.mean(), nfk.max()))
    assert 0.1 < want.mean() < 0.95 and nfk.max() > 8


def test_edges_with_rotation_wraparound(irt, orc, helpers):
    W = irt.workloads
    robot = W.robot_config2()
    robot.enable_rotation = True
    vox, _ = W.reach_environment(seed=8, n_spheres=48)
    a, b = _edges(robot, irt, 600, seed=32, step=2.0)
    # force a share of the edges across the +-pi seam
    a[:100, 3] = np.pi - 0.05
    b[:100, 3] = -np.pi + 0.07
    _compare(irt, orc, helpers, robot, vox, a, b)


def test_edge_degenerate_and_single(irt, orc, helpers):
    W = irt.workloads
    robot = W.robot_config2()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    mv = irt.VoxelBackboneMotionValidator(chk)
    s = np.array([[3.0, 1.0, 2.0]])
    d = mv.check_motion_detail(s, s)                       # a == b: two samples, verdict = isValid(a)
    assert d["n_fk"][0] == 2 and d["valid"][0] == chk.isValid(s[0])
    assert mv.checkMotion([3.0, 1.0, 2.0], [3.5, 1.2, 2.1]) in (True, False)
    assert mv.check_motion(np.zeros((0, 3)), np.zeros((0, 3))).size == 0
    with pytest.raises(irt.InvalidArgument):
        mv.check_motion(np.zeros((2, 3)), np.zeros((3, 3)))


def test_edges_leaving_the_voxel_domain_are_reported(irt):
    """find_cell throws std::domain_error in the reference; here the edge is invalid and counted."""
    W = irt.workloads
    robot = W.robot_config2()
    small = irt.VoxelOctree(64)                            # domain smaller than the robot's reach
    small.set_xlim(-0.06, 0.06); small.set_ylim(-0.06, 0.06); small.set_zlim(-0.01, 0.11)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), small)
    mv = irt.VoxelBackboneMotionValidator(chk)
    a = np.array([[0.0, 0.0, 0.0], [0.1, 0.0, 0.0]])
    b = np.array([[6.0, 0.0, 0.0], [0.2, 0.1, 0.0]])
    d = mv.check_motion_detail(a, b)
    assert d["n_domain_errors"] >= 1 and not d["valid"][0]


def test_check_motion_with_last_valid(irt, orc, helpers):
    """checkMotion(s1, s2, last_valid): verdict and PartialVoxelization::t against the oracle's DFS."""
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    mv = irt.VoxelBackboneMotionValidator(chk)
    a, b = _edges(robot, irt, 500, seed=33, step=6.0)
    valid, lvt = mv.check_motion_last_valid(a, b)
    assert np.array_equal(valid, mv.check_motion(a, b))          # same boolean as the two-argument form
    orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
    for i in range(len(a)):
        w = orc.check_motion_until_invalid(orb, og, a[i], b[i])
        assert valid[i] == w["is_fully_valid"] and lvt[i] == w["last_valid_t"], (i, lvt[i], w)
    assert (lvt[valid] == 1.0).all() and ((lvt[~valid] > 0) & (lvt[~valid] < 1)).sum() > 10


def test_discrete_motion_validator_matches_oracle(irt, orc, helpers):
    """VoxelBackboneDiscreteMotionValidator: verdict of checkMotion(s1, s2) (union tested against the
    obstacles) and verdict / t / sample count of checkMotion(s1, s2, last_valid) (sequential loop)."""
    W = irt.workloads
    rot = W.robot_config2()
    rot.enable_rotation = True
    for robot, seed, step in ((W.robot_config3(), 35, 1.5), (rot, 36, 2.5)):
        vox, _ = W.reach_environment(seed=7, n_spheres=64)
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        mv = irt.VoxelBackboneDiscreteMotionValidator(chk)
        a, b = _edges(robot, irt, 160, seed=seed, step=step)
        b[:3] = a[:3]                                                 # nd = 0: a and b only
        d = mv.check_motion_detail(a, b)
        orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
        for i in range(len(a)):
            w0 = orc.check_motion_discrete(orb, og, a[i], b[i], until_invalid=False)
            w1 = orc.check_motion_discrete(orb, og, a[i], b[i], until_invalid=True)
            assert d["valid"][i] == w0["valid"] == w1["is_fully_valid"], (i, w0, w1)
            assert d["last_valid_t"][i] == w1["last_valid_t"] and d["n_fk"][i] == w1["n_fk"], (i, d["n_fk"][i], w1)
        assert (d["n_fk"][:3] <= 2).all() and (d["n_fk"][:3][d["valid"][:3]] == 2).all()
        assert 0.05 < d["valid"].mean() < 0.98 and d["n_fk"].max() > 50
