"""GPU parity of the robot voxel sets used for roadmap caches (tr_voxelize_batch / tr_voxelize_edges):
voxelizeVertex / voxelizeEdge of VoxelCachedLazyPRM.cpp:2803-2837,2879-2902."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _item(out, i):
    a, b = out["offsets"][i], out["offsets"][i + 1]
    o = np.argsort(out["block_ids"][a:b])
    return out["block_ids"][a:b][o], out["masks"][a:b][o]


def test_vertex_voxel_sets_bit_exact(irt, orc, helpers):
    """Block lists == oracle add_piecewise_line of the SAME (GPU-computed) backbone points."""
    W = irt.workloads
    robot = W.robot_config2()
    for t in robot.tendons:
        t.max_tension, t.max_length = 60.0, 0.02
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    rot = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])
    env = irt.VoxelEnvironment()
    env.inv_rotation = rot
    chk = irt.VoxelBackboneValidityChecker(robot, env, vox)
    st = W.random_states(robot, 1200, seed=71, tau_max=40.0)
    out = chk.engine.voxelize_batch(st)
    det = chk.is_valid_detail(st)
    want_shape = (det["flags"] & 7) == 7
    assert np.array_equal(out["shape_valid"], want_shape)
    ok = (det["flags"] & 1) > 0                     # stored-point kernel vs fk_verdict: equal to rounding where the solve converged
    assert np.abs(out["tips"][ok] - det["tips"][ok]).max() <= 1e-12
    assert 0 < want_shape.sum() < len(st)
    pts = robot.shape_batch(st)["p"]
    ref = orc.Grid(256, vox.limits())
    for i in range(len(st)):
        ids, masks = _item(out, i)
        if not want_shape[i]:
            assert ids.size == 0
            continue
        g = ref.empty_copy()
        g.add_piecewise_line(pts[i] @ rot.T)
        wi, wm = g.export_blocks()
        assert np.array_equal(ids, wi) and np.array_equal(masks, wm), i
    # a cached set collides with the obstacles exactly when the state's voxel flag says so
    hit = chk.engine.check_cached(out["block_ids"], out["masks"], out["offsets"])
    assert np.array_equal(hit[want_shape], ((det["flags"] & 8) == 0)[want_shape])
    assert not hit[~want_shape].any()


def test_edge_swept_volumes_match_oracle(irt, orc, helpers):
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rng = np.random.default_rng(72)
    a = W.random_states(robot, 300, seed=72, tau_max=15.0)
    b = np.clip(a + rng.normal(size=a.shape) * 0.8, 0, 20)
    out = chk.engine.voxelize_edges(a, b)
    orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
    n_same = n_full = 0
    for i in range(len(a)):
        w = orc.check_motion(orb, og, a[i], b[i], want_swept=True)
        assert out["fully_valid"][i] == w["is_fully_valid"]
        ids, masks = _item(out, i)
        if not w["is_fully_valid"]:
            assert ids.size == 0
            continue
        n_full += 1
        assert out["n_fk"][i] == w["n_fk"]
        wi, wm = w["swept"].export_blocks()
        if np.array_equal(ids, wi) and np.array_equal(masks, wm):
            n_same += 1
        else:   # GPU FK vs oracle FK points differ by ~1e-15 m: a point on a voxel face may flip one cell
            d = {int(k): int(v) for k, v in zip(ids, masks)}
            e = {int(k): int(v) for k, v in zip(wi, wm)}
            diff = sum(bin(d.get(k, 0) ^ e.get(k, 0)).count("1") for k in set(d) | set(e))
            assert diff <= 2
    assert n_full > 50 and n_same >= n_full - 1
    # cached edge sets vs obstacles == checkMotion verdict for the fully valid edges
    hit = chk.engine.check_cached(out["block_ids"], out["masks"], out["offsets"])
    verdict = irt.VoxelBackboneMotionValidator(chk).check_motion(a, b)
    assert np.array_equal(verdict, out["fully_valid"] & ~hit)


def test_voxelize_empty(irt):
    W = irt.workloads
    robot = W.robot_config2()
    vox, _ = W.reach_environment(seed=7, n_spheres=4)
    e = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox).engine
    out = e.voxelize_batch(np.zeros((0, 3)))
    assert out["offsets"].tolist() == [0] and out["block_ids"].size == 0


def test_voxel_sets_with_retraction(irt, orc, helpers):
    """Retraction robots: per-configuration point counts and rows aligned at the tip -- vertex sets equal the
    oracle's add_piecewise_line of the same points, edge sets are consistent with checkMotion."""
    W = irt.workloads
    robot = W.robot_config2()
    robot.enable_retraction = True
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    st = W.random_states(robot, 500, seed=73, tau_max=14.0)
    st[:, -1] = np.random.default_rng(74).uniform(0.0, 0.12, len(st))
    st[:3, -1] = [0.0, robot.specs.L, robot.specs.L - robot.specs.dL / 4]
    out = chk.engine.voxelize_batch(st)
    det = chk.is_valid_detail(st)
    want_shape = (det["flags"] & 7) == 7
    assert np.array_equal(out["shape_valid"], want_shape) and want_shape.sum() > 100
    fk = robot.shape_batch(st)
    ref = orc.Grid(256, vox.limits())
    for i in range(len(st)):
        ids, masks = _item(out, i)
        if not want_shape[i]:
            assert ids.size == 0
            continue
        g = ref.empty_copy()
        g.add_piecewise_line(fk["p"][i, :fk["n_points"][i]])
        wi, wm = g.export_blocks()
        assert np.array_equal(ids, wi) and np.array_equal(masks, wm), i
    hit = chk.engine.check_cached(out["block_ids"], out["masks"], out["offsets"])
    assert np.array_equal(hit[want_shape], ((det["flags"] & 8) == 0)[want_shape])
    rng = np.random.default_rng(75)
    a = st[:200]
    b = a + rng.normal(size=a.shape) * np.array([0.8, 0.8, 0.8, 0.004])
    b[:, :3] = np.clip(b[:, :3], 0, 20); b[:, 3] = np.clip(b[:, 3], 0, 0.2)
    ec = chk.engine.voxelize_edges(a, b)
    ehit = chk.engine.check_cached(ec["block_ids"], ec["masks"], ec["offsets"])
    verdict = irt.VoxelBackboneMotionValidator(chk).check_motion(a, b)
    assert np.array_equal(verdict, ec["fully_valid"] & ~ehit) and 0.05 < verdict.mean() < 0.98
    orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
    for i in range(0, 200, 9):
        w = orc.check_motion(orb, og, a[i], b[i], want_swept=True)
        assert ec["fully_valid"][i] == w["is_fully_valid"]
        if w["is_fully_valid"]:
            ids, masks = _item(ec, i)
            wi, wm = w["swept"].export_blocks()
            d = {int(k): int(v) for k, v in zip(ids, masks)}
            e = {int(k): int(v) for k, v in zip(wi, wm)}
            assert sum(bin(d.get(k, 0) ^ e.get(k, 0)).count("1") for k in set(d) | set(e)) <= 2


def test_indexed_edge_caches_equal_the_pairwise_form(irt):
    """tr_voxelize_edges_indexed integrates and voxelises every roadmap vertex once for all of its edges; the swept-volume
    sets, fully-valid bits and sample counts equal tr_voxelize_edges on the gathered end states -- also for a retraction
    robot (rows aligned at the tip), with shape-invalid vertices, and with a sample pool so small that the chunks overflow
    and the vertex block does not fit (host-gather fallback)."""
    import os
    W = irt.workloads
    ret = W.robot_config2()
    ret.enable_retraction = True
    hard = W.robot_config3()
    for t in hard.tendons:
        t.max_length = 0.012                                   # many length-limit violations: edges without a cache
    for robot, tau in ((W.robot_config3(), 20.0), (ret, 14.0), (hard, 20.0)):
        vox, _ = W.reach_environment(seed=7, n_spheres=64)
        states = W.random_states(robot, 700, seed=81, tau_max=tau)
        if robot.enable_retraction:
            states[:, -1] = np.random.default_rng(82).uniform(0.0, 0.1, len(states))
        nt = len(robot.tendons)
        near = np.argsort(np.linalg.norm(states[:, None, :nt] - states[None, :, :nt], axis=2), axis=1)[:, 1:6]
        edges = np.stack([np.repeat(np.arange(len(states)), 5), near.reshape(-1)], 1)
        edges = edges[edges[:, 0] < edges[:, 1]]

        def run():
            eng = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox).engine
            return eng.voxelize_edges_indexed(states, edges), eng.voxelize_edges(states[edges[:, 0]], states[edges[:, 1]])

        for env in ({}, {"TENDON_HIP_EDGE_POOL": "4096"}, {"TENDON_HIP_EDGE_POOL": "1024"}):
            old = {k: os.environ.get(k) for k in env}
            os.environ.update(env)
            try:
                ix, pw = run()
            finally:
                for k, v in old.items():
                    os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
            for k in ("offsets", "block_ids", "masks", "fully_valid", "n_fk"):
                assert np.array_equal(ix[k], pw[k]), (k, env)
        assert 0.02 < ix["fully_valid"].mean() <= 1.0 and ix["offsets"][-1] > 1000
    assert not ix["fully_valid"].all()                         # the last robot leaves edges without a cache


def test_vertex_sets_are_duplicate_free_and_ordered(irt):
    """The per-sample lists may hold a block twice (a backbone that returns to it); what tr_voxelize_batch delivers is
    duplicate-free and ordered by block id -- also for tightly curled robots, where re-entries are common."""
    W = irt.workloads
    thin = W.robot_config1()
    thin.specs.dL = 0.2 / 128
    thin.r = 0.002
    for t in thin.tendons:
        t.max_tension, t.min_length, t.max_length = 100.0, -1.0, 1.0
    vox, _ = W.reach_environment(seed=7, n_spheres=8)
    eng = irt.VoxelBackboneValidityChecker(thin, irt.VoxelEnvironment(), vox).engine
    st = W.random_states(thin, 3000, seed=9, tau_max=100.0)
    out = eng.voxelize_batch(st)
    off = out["offsets"]
    assert out["shape_valid"].sum() > 500
    for i in np.flatnonzero(out["shape_valid"])[:800]:
        ids = out["block_ids"][off[i]:off[i + 1]]
        assert (np.diff(ids.astype(np.int64)) > 0).all() and (out["masks"][off[i]:off[i + 1]] != 0).all()


def test_connect_is_check_motion_and_voxelize_edge_in_one_pass(irt):
    """tr_connect_edges_indexed: verdicts and FK counts of tr_validate_edges_indexed, voxel sets of
    tr_voxelize_edges_indexed for exactly the accepted edges -- also with a pool so small that chunks are retried and with
    more vertices than the pool holds (host-gather fallback to the pairwise form)."""
    import os
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=110)
    states = W.random_states(robot, 1200, seed=61, tau_max=20.0)
    near = np.argsort(np.linalg.norm(states[:, None, :] - states[None, :, :], axis=2), axis=1)[:, 1:6]
    edges = np.stack([np.repeat(np.arange(len(states)), 5), near.reshape(-1)], 1).astype(np.int32)

    def run():
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        mv = irt.VoxelBackboneMotionValidator(chk)
        rb = irt.RoadmapBuilder(chk, mv, seed=1)
        return mv.check_motion_indexed(states, edges), chk.engine.voxelize_edges_indexed(states, edges), rb.connect(states, edges)

    for env in ({}, {"TENDON_HIP_EDGE_POOL": "4096"}, {"TENDON_HIP_EDGE_POOL": "1024"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            val, cache, (e_ok, got) = run()
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        ok = val["valid"]
        assert 0.2 < ok.mean() < 0.98                             # both outcomes are well represented
        assert np.array_equal(e_ok, edges[ok]) and np.array_equal(got["n_fk"], val["n_fk"][ok])
        assert (cache["fully_valid"] | ~ok).all()                 # an edge checkMotion accepts has a valid shape all along
        off = cache["offsets"]
        idx = np.concatenate([np.arange(off[e], off[e + 1]) for e in np.flatnonzero(ok)]) if ok.any() else np.zeros(0, int)
        assert np.array_equal(got["block_ids"], cache["block_ids"][idx]) and np.array_equal(got["masks"], cache["masks"][idx])
        assert np.array_equal(np.diff(got["offsets"]), np.diff(off)[ok]) and got["offsets"][-1] == len(got["block_ids"])


def test_retraction_levels_dealt_by_length_give_the_same_sets(irt):
    """Stored-point edge forms of a retraction robot (tr_voxelize_edges, the indexed form, connect) re-deal every bisection level
    of TENDON_HIP_RETRACT_SORT samples or more in the order of the samples' backbone lengths (edge_level_gather), so that K1r's waves
    are homogeneous.  Which pool slot a sample gets is bookkeeping: per edge the voxel set (block ids and masks), the fully-valid
    bit, checkMotion's verdict and the FK count are those of arrival order."""
    import os
    W = irt.workloads
    robot = W.robot_config3()
    robot.enable_rotation = True
    robot.enable_retraction = True
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    rng = np.random.default_rng(91)
    nv = 700
    states = W.random_states(robot, nv, seed=92, tau_max=18.0)
    states[:, -1] = rng.uniform(0.0, 0.16, nv)
    nt = len(robot.tendons)
    scale = np.array([1.0] * nt + [3.0, 400.0])
    near = np.argsort(np.linalg.norm((states[:, None, :] - states[None, :, :]) * scale, axis=2), axis=1)[:, 1:6]
    edges = np.stack([np.repeat(np.arange(nv), 5), near.reshape(-1)], 1)

    def run():
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        eng = chk.engine
        mv = irt.VoxelBackboneMotionValidator(chk)
        return (eng.voxelize_edges_indexed(states, edges, mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change),
                eng.voxelize_edges(states[edges[:400, 0]], states[edges[:400, 1]]),
                eng.voxelize_edges_indexed(states, edges, mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change, validate=True))

    def with_sort(v):
        old = os.environ.get("TENDON_HIP_RETRACT_SORT")
        os.environ["TENDON_HIP_RETRACT_SORT"] = v
        try:
            return run()
        finally:
            os.environ.pop("TENDON_HIP_RETRACT_SORT", None) if old is None else os.environ.__setitem__("TENDON_HIP_RETRACT_SORT", old)

    want, got = with_sort("0"), with_sort("64")
    for w, g in zip(want, got):
        for k in w:
            if isinstance(w[k], np.ndarray):
                assert np.array_equal(w[k], g[k]), k
    assert want[0]["fully_valid"].sum() > 500 and want[0]["offsets"][-1] > 10000
