"""GPU parity of K4 cached_blocks_vs_grid (tr_check_cached): roadmap voxel caches (sparse block
lists) re-validated against a changed obstacle grid -- BASELINE config 5's inner operation
(VoxelCachedLazyPRM.cpp:2397-2411: obstacles.collides(*cached_voxels)).  Bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _caches(irt, orc, helpers, robot, states):
    """Per-state backbone voxel sets as CSR (oracle voxelisation of oracle shapes = what a roadmap stores)."""
    orb = helpers.oracle_robot(orc, robot)
    ref = orc.Grid(256, (-0.25, 0.25) * 3)
    ids, masks, offsets = [], [], [0]
    for s in states:
        g = ref.empty_copy()
        g.add_piecewise_line(orb.shape(s)["p"])
        i, m = g.export_blocks()
        ids.append(i); masks.append(m); offsets.append(offsets[-1] + len(i))
    return np.concatenate(ids), np.concatenate(masks), np.array(offsets, dtype=np.int64)


def test_cached_sets_vs_changed_obstacles(irt, orc, helpers):
    W = irt.workloads
    robot = W.robot_config2()
    states = W.random_states(robot, 800, seed=51, tau_max=15.0)
    ids, masks, offsets = _caches(irt, orc, helpers, robot, states)
    assert 20 < (offsets[1] - offsets[0]) < 200
    for seed, n_s in ((7, 64), (99, 72)):                    # the environment changes between queries
        vox, _ = W.reach_environment(seed=seed, n_spheres=n_s)
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        got = chk.engine.check_cached(ids, masks, offsets)
        want = orc.check_cached(helpers.oracle_grid(orc, vox), ids, masks, offsets)
        assert np.array_equal(got, want)
        assert 0.05 < want.mean() < 0.95
        # a cached set collides exactly when the state's voxel collision flag says so
        fl = chk.is_valid_detail(states)["flags"]
        reach = (fl & 7) == 7
        assert np.array_equal(((fl & 8) == 0)[reach], want[reach])


def test_cached_edge_cases(irt):
    W = irt.workloads
    robot = W.robot_config2()
    vox, _ = W.reach_environment(seed=7, n_spheres=8)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    e = chk.engine
    oid, om = vox.to_sparse()
    # empty item, item equal to one obstacle block, item with a disjoint mask in an occupied block
    free_bit = np.uint64(1) << np.uint64(int(np.flatnonzero([(int(om[0]) >> b) & 1 == 0 for b in range(64)])[0])) \
        if int(om[0]) != 2 ** 64 - 1 else np.uint64(0)
    ids = np.array([oid[0], oid[0]], dtype=np.uint32)
    masks = np.array([om[0], free_bit], dtype=np.uint64)
    offsets = np.array([0, 0, 1, 2], dtype=np.int64)
    assert e.check_cached(ids, masks, offsets).tolist() == [False, True, False]
    assert e.check_cached(np.zeros(0, np.uint32), np.zeros(0, np.uint64), np.zeros(1, np.int64)).size == 0
    with pytest.raises(irt.InvalidArgument):                 # block id beyond the grid = dimension mismatch
        e.check_cached(np.array([64 ** 3], np.uint32), np.array([1], np.uint64), np.array([0, 1], np.int64))
    with pytest.raises(irt.InvalidArgument):
        e.check_cached(ids, masks, np.array([0, 2, 1], dtype=np.int64))
