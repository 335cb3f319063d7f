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
        d = mv.check_motion_detail(a, b, last_valid=True)
        assert np.array_equal(mv.check_motion(a, b), d["valid"])     # the two-argument form: same verdict under the backbone checker
        orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
        for i in range(len(a)):
            w0 = orc.check_motion_discrete(orb, og, a[i], b[i], until_invalid=False)
            w1 = orc.check_motion_discrete(orb, og, a[i], b[i], until_invalid=True)
            assert d["valid"][i] == w0["valid"] == w1["is_fully_valid"], (i, w0, w1)
            assert d["last_valid_t"][i] == w1["last_valid_t"] and d["n_fk"][i] == w1["n_fk"], (i, d["n_fk"][i], w1)
        assert (d["n_fk"][:3] <= 2).all() and (d["n_fk"][:3][d["valid"][:3]] == 2).all()
        assert 0.05 < d["valid"].mean() < 0.98 and d["n_fk"].max() > 50


def _with_env(irt, env, fn):
    """Run fn with environment overrides that libtendon_hip reads when a context is created."""
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


def test_small_sample_pool_gives_the_same_edge_results(irt):
    """The edge calls process edges in chunks that fit the sample pool and halve a chunk whose bisection
    outgrows it; a pool of 1024 samples forces both paths.  Results must not depend on the pool size."""
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    a, b = _edges(robot, irt, 700, seed=41, step=4.0)

    def run():
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        mv = irt.VoxelBackboneMotionValidator(chk)
        d, (v2, lvt) = mv.check_motion_detail(a, b), mv.check_motion_last_valid(a, b)
        dd = irt.VoxelBackboneDiscreteMotionValidator(chk).check_motion_detail(a[:60], b[:60], last_valid=True)
        ec = chk.engine.voxelize_edges(a[:200], b[:200])
        return d, v2, lvt, dd, ec

    big = run()
    small = _with_env(irt, {"TENDON_HIP_EDGE_POOL": "1024"}, run)
    assert np.array_equal(big[0]["valid"], small[0]["valid"]) and np.array_equal(big[0]["n_fk"], small[0]["n_fk"])
    assert big[0]["n_domain_errors"] == small[0]["n_domain_errors"]
    assert np.array_equal(big[1], small[1]) and np.array_equal(big[2], small[2])
    for k in ("valid", "n_fk", "last_valid_t"):
        assert np.array_equal(big[3][k], small[3][k]), k
    for k in ("offsets", "block_ids", "masks", "fully_valid"):
        assert np.array_equal(big[4][k], small[4][k]), k
    assert big[0]["n_fk"].max() > 8 and 0.05 < big[0]["valid"].mean() < 0.95


def test_fused_and_separate_kernels_give_identical_bits(irt):
    """The three schedules of tr_validate_batch* -- fk_verdict (no point storage; the default), fk_sweep_fused (K1's and
    K2's bodies in one launch over stored points) and the two kernels launched one after the other -- give the same
    verdict bits, flags and tips."""
    W = irt.workloads
    for robot in (W.robot_config2(), W.robot_config3()):
        vox, _ = W.reach_environment(seed=7, n_spheres=64)
        states = W.random_states(robot, 20000 + 37, seed=91, tau_max=14.0)

        def run():
            chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
            return chk.is_valid_detail(states)

        v, f, s = run(), _with_env(irt, {"TENDON_HIP_FUSED": "1"}, run), _with_env(irt, {"TENDON_HIP_FUSED": "0"}, run)
        # The three kernels hold the same RK4 body, but hipcc is free to pair the multiply-adds of a sum of products differently
        # in another kernel (the FK is compiled with fp-contract=fast; its parity bar is a tolerance): the tips agree to rounding
        # -- one ulp in about a hundredth of the configurations since round 4's L D L^T solve; rounds 1 - 3 happened to compile
        # the stored-point kernels to the same bits -- and the verdicts and flags are the same
        for o in (f, v):
            assert np.array_equal(o["valid"], s["valid"]) and np.array_equal(o["flags"], s["flags"])
            assert np.abs(o["tips"] - s["tips"]).max() <= 1e-13
        assert 0.2 < f["valid"].mean() < 0.95


def test_verdict_only_kernel_all_branches(irt):
    """fk_verdict against the stored-point kernels where the predicate's rarer branches are busy: self collisions (the
    fallback pass: exact pairwise sweep on re-integrated points, here through a 64-column workspace so that the list
    is worked off in many rounds), length limits, non-converged base solves, a rotated environment, a rotating robot,
    points leaving the voxel domain, and the debug switches (every lane through the fallback; brute-force pairs; no
    dilated-grid fast path = every segment through the deferred DDA walk)."""
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
    small, _ = W.sphere_environment(seed=5, n_spheres=40, radius=0.01, N=128, half=0.12, keepout=0.02)   # the robot reaches out of it
    a = 0.4
    rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    cases = [(thin, 100.0, None, None, 6000), (hard, 45.0, None, None, 4000), (spin, 12.0, rot, None, 6000), (W.robot_config3(), 20.0, rot, None, 5000),
             (W.robot_config2(), 14.0, None, small, 5000)]
    seen = np.zeros(32, int)
    for robot, tau_max, inv_rot, vox, n in cases:
        if vox is None:
            vox, _ = W.reach_environment(seed=3, n_spheres=48)
        env = irt.VoxelEnvironment()
        if inv_rot is not None:
            env.inv_rotation = inv_rot
        states = W.random_states(robot, n + 21, seed=17, tau_max=tau_max)

        def run(debug=0, detail=True):
            chk = irt.VoxelBackboneValidityChecker(robot, env, vox)
            chk.engine.set_debug(debug)
            return chk.is_valid_detail(states) if detail else dict(valid=chk.is_valid(states))

        want = _with_env(irt, {"TENDON_HIP_FUSED": "0"}, run)
        for debug in (0, 2, 3, 4):
            got = _with_env(irt, {"TENDON_HIP_FB_CAP": "64"}, lambda: run(debug))
            for k in ("valid", "flags"):
                assert np.array_equal(got[k], want[k]), (k, debug, np.flatnonzero(got[k] != want[k])[:8])
            ok = want["flags"] & 1 > 0                             # (tips of non-converged solves are garbage in, garbage out)
            assert np.abs(got["tips"][ok] - want["tips"][ok]).max() <= 1e-12
        # without the flags output an obstacle hit settles a configuration whatever its self-collision test would say
        assert np.array_equal(run(0, detail=False)["valid"], want["valid"])
        seen += np.bincount(want["flags"], minlength=32)
    print("flag histogram", seen)
    assert seen[15] > 0 and seen[7] > 0 and seen[3] > 0 and seen[1] > 0 and seen[0] > 0


def test_indexed_edges_equal_the_pairwise_form(irt):
    """tr_validate_edges_indexed evaluates every vertex once for all of its edges; verdicts, n_fk and the count of
    domain errors equal tr_validate_edges on the gathered end states -- also with a pool so small that the vertex
    block does not fit (host-gather fallback) and chunks overflow."""
    W = irt.workloads
    for mk, rot in ((W.robot_config3, False), (W.robot_config2, True)):
        robot = mk()
        robot.enable_rotation = rot
        vox, _ = W.reach_environment(seed=7, n_spheres=64)
        states = W.random_states(robot, 900, seed=45)
        rng = np.random.default_rng(46)
        near = np.argsort(np.linalg.norm(states[:, None, :len(robot.tendons)] - states[None, :, :len(robot.tendons)], axis=2), axis=1)[:, 1:7]
        edges = np.stack([np.repeat(np.arange(900), 6), near.reshape(-1)], 1)
        edges = np.concatenate([edges, [[5, 5], [7, 3], [3, 7]]])                       # a == b, both orientations

        def run():
            chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
            mv = irt.VoxelBackboneMotionValidator(chk)
            return mv.check_motion_indexed(states, edges), mv.check_motion_detail(states[edges[:, 0]], states[edges[:, 1]])

        for env in ({}, {"TENDON_HIP_EDGE_POOL": "1024"}, {"TENDON_HIP_EDGE_POOL": "4096"}):
            got, want = _with_env(irt, env, run)
            for k in ("valid", "n_fk", "n_domain_errors"):
                assert np.array_equal(got[k], want[k]), (k, env)
        assert 0.05 < got["valid"].mean() < 0.99 and got["n_fk"].max() > 4
    with pytest.raises(irt.OutOfRange):
        irt.VoxelBackboneMotionValidator(irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)).check_motion_indexed(states, [[0, 900]])


def test_two_lane_bisection_equals_one_lane(irt, orc, helpers):
    """tr_validate_edges_indexed bisects the parts of a roadmap's edge list side by side on two to four streams (each lane with its
    share of the sample pool, its own frontier, counters and fallback list; one host thread alternating between them).  Verdicts,
    FK counts and the domain-error count equal the one-lane path's (TENDON_HIP_EDGE_LANES=1) -- where samples need the fallback
    pass (a slender robot under high tension, through a 64-column fallback workspace), with a rotating robot in a rotated
    environment, and with a pool so small that a lane overflows and the call falls back to one lane -- and the oracle's on a
    sample of the edges."""
    W = irt.workloads
    thin = W.robot_config1()
    thin.specs.dL = 0.2 / 128
    thin.r = 0.008
    for t in thin.tendons:
        t.max_tension, t.min_length, t.max_length = 100.0, -1.0, 0.12
    spin = W.robot_config3()
    spin.enable_rotation = True
    a = 0.4
    rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    pull = W.robot_config2()                       # retraction: every lane orders its own launches by backbone length
    pull.enable_retraction = True
    for robot, tau_max, inv_rot, nv in ((thin, 100.0, None, 1500), (spin, 20.0, rot, 1500), (pull, 14.0, None, 1500)):
        vox, _ = W.reach_environment(seed=7, n_spheres=48)
        env = irt.VoxelEnvironment()
        if inv_rot is not None:
            env.inv_rotation = inv_rot
        states = W.random_states(robot, nv, seed=81, tau_max=tau_max)
        if robot.enable_retraction:
            states[:, -1] = 0.03 + 0.004 * np.random.default_rng(83).random(nv) + 0.1 * (np.arange(nv) % 2)     # two groups of lengths
        nt = len(robot.tendons)
        near = np.argsort(np.linalg.norm(states[:, None, :nt] - states[None, :, :nt], axis=2), axis=1)[:, 1:8]
        edges = np.stack([np.repeat(np.arange(nv), 7), near.reshape(-1)], 1)            # 10 500 edges: above the two-lane threshold

        def run():
            chk = irt.VoxelBackboneValidityChecker(robot, env, vox)
            return irt.VoxelBackboneMotionValidator(chk).check_motion_indexed(states, edges)

        if robot.enable_retraction:                 # neighbours within a group of lengths
            near = np.stack([np.flatnonzero(np.arange(nv) % 2 == i % 2)[np.argsort(np.linalg.norm(
                states[np.arange(nv) % 2 == i % 2, :nt] - states[i, :nt], axis=1))[1:8]] for i in range(nv)])
            edges = np.stack([np.repeat(np.arange(nv), 7), near.reshape(-1)], 1)
        base_env = {"TENDON_HIP_FB_CAP": "64", "TENDON_HIP_RETRACT_SORT": "64"}          # (launches of 64 samples or more are ordered by length)
        want = _with_env(irt, dict(base_env, TENDON_HIP_EDGE_LANES="1"), run)
        for lanes, extra in (("2", {}), ("3", {}), ("4", {}),
                             ("2", {"TENDON_HIP_EDGE_POOL": "40000", "TENDON_HIP_EDGE_LANE_GUESS": "1"}),    # lanes overflow their share
                             ("4", {"TENDON_HIP_EDGE_POOL": "40000", "TENDON_HIP_EDGE_LANE_GUESS": "1"})):
            got = _with_env(irt, dict(base_env, TENDON_HIP_EDGE_LANES=lanes, **extra), run)
            for k in ("valid", "n_fk", "n_domain_errors"):
                assert np.array_equal(got[k], want[k]), (k, lanes, extra, np.flatnonzero(np.asarray(got[k]) != np.asarray(want[k]))[:8])
        assert 0.2 < want["valid"].mean() < 0.99 and want["n_fk"].max() > 6
        idx = np.random.default_rng(82).choice(len(edges), 400, replace=False)
        ov, onf, _ = orc.check_motion_batch(helpers.oracle_robot(orc, robot, lib="omp"), helpers.oracle_grid(orc, vox), states[edges[idx, 0]],
                                            states[edges[idx, 1]], inv_rot=np.eye(3) if inv_rot is None else inv_rot, nthreads=0, lib=orc.omp_lib())
        assert np.array_equal(want["valid"][idx], ov)
        assert np.array_equal(want["n_fk"][idx][ov], onf[ov])


def test_device_resident_connect_and_validate_equal_the_host_forms(irt):
    """tr_knn_edges_dev + tr_validate_edges_indexed_dev (vertex states, edge list, mask and FK counts all in HBM) give the edge list,
    verdicts, FK counts and domain-error count of tr_knn_edges + tr_validate_edges_indexed -- on one lane and on two, with and
    without the FK counts, and with a pool too small for the vertex block (both forms then gather the end states) -- and the same
    errors: an index outside the vertex array."""
    import torch
    W = irt.workloads
    for mk, rot, nv, k in ((W.robot_config3, False, 700, 6), (W.robot_config2, True, 2600, 8)):
        robot = mk()
        robot.enable_rotation = rot
        vox, _ = W.reach_environment(seed=7, n_spheres=64)
        states = W.random_states(robot, nv, seed=145)
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        eng = chk.engine
        edges = eng.knn_edges(states, k)
        want = eng.validate_edges_indexed(states, edges)
        d_states = torch.from_numpy(states).cuda()
        d_edges = torch.full((nv * k, 2), -7, dtype=torch.int32, device="cuda")
        ne = eng.knn_edges_dev(d_states, nv, k, d_edges)
        assert ne == len(edges) and np.array_equal(d_edges[:ne].cpu().numpy(), edges)
        assert (d_edges[ne:] == -7).all()                                  # nothing written past the list
        d_bits = torch.zeros((ne + 63) // 64, dtype=torch.int64, device="cuda")
        d_nfk = torch.zeros(ne, dtype=torch.int32, device="cuda")
        nd = eng.validate_edges_indexed_dev(d_states, nv, d_edges, ne, d_bits, d_nfk)
        bits = d_bits.cpu().numpy().view(np.uint64)
        assert np.array_equal(irt.unpack_bits(bits, ne), want["valid"])
        assert np.array_equal(d_nfk.cpu().numpy(), want["n_fk"]) and nd == want["n_domain_errors"]
        d_bits.zero_()
        assert eng.validate_edges_indexed_dev(d_states, nv, d_edges, ne, d_bits) == nd        # without the counts
        assert np.array_equal(d_bits.cpu().numpy().view(np.uint64), bits)
        assert 0.05 < want["valid"].mean() < 0.99 and (nv < 2000 or ne >= 8192)          # (the second case runs on two lanes)
        # the edge list truncated at the capacity: the count is still the whole list's
        d_few = torch.zeros((10, 2), dtype=torch.int32, device="cuda")
        assert eng.knn_edges_dev(d_states, nv, k, d_few) == ne and np.array_equal(d_few.cpu().numpy(), edges[:10])
        bad = d_edges[:ne].clone()
        bad[ne // 2, 1] = nv
        with pytest.raises(irt.OutOfRange):
            eng.validate_edges_indexed_dev(d_states, nv, bad, ne, d_bits)
        bad[ne // 2, 1] = -1
        with pytest.raises(irt.OutOfRange):
            eng.validate_edges_indexed_dev(d_states, nv, bad, ne, d_bits)
        assert eng.validate_edges_indexed_dev(d_states, nv, d_edges, ne, d_bits) == nd        # and the context is usable afterwards
        assert np.array_equal(d_bits.cpu().numpy().view(np.uint64), bits)
        with pytest.raises(irt.InvalidArgument):
            eng.validate_edges_indexed_dev(d_states.cpu(), nv, d_edges, ne, d_bits)

    def small_pool():
        # a pool too small for the vertex block (more vertices than half of it): the device form gathers the end states like the
        # host form (round 3 reported TR_ERR_UNSUPPORTED here) -- same mask, counts and domain errors
        c2 = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        b2, n2 = torch.zeros_like(d_bits), torch.zeros(ne, dtype=torch.int32, device="cuda")
        nd2 = c2.engine.validate_edges_indexed_dev(d_states, nv, d_edges, ne, b2, n2)
        return b2.cpu().numpy().view(np.uint64), n2.cpu().numpy(), nd2

    sb, sn, snd = _with_env(irt, {"TENDON_HIP_EDGE_POOL": "1024"}, small_pool)
    assert np.array_equal(sb, bits) and np.array_equal(sn, want["n_fk"]) and snd == nd


def test_vertex_signatures_handed_over_from_the_vertex_phase(irt):
    """tr_validate_candidates_sig_dev writes each candidate's backbone signature next to its verdict; the accepted candidates' rows,
    compacted like their states, let tr_validate_edges_indexed_sig_dev skip its vertex pass: same mask as the plain candidate call,
    same verdicts, FK counts and domain-error count as tr_validate_edges_indexed_dev (one lane and several), and the rows are the ones the
    edge call computes for itself.  Contexts that cannot hand signatures over say so."""
    import torch
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    eng = chk.engine
    sw = eng.signature_words()
    assert sw > 0 and sw % 2 == 0 and sw >= eng.num_points
    box = D.sampling_box(robot)
    S = eng.state_size
    for M, k in ((1536, 6), (6016, 9)):                                   # the second: two lanes
        seed = 17
        d_bits0 = torch.zeros((M + 63) // 64, dtype=torch.int64, device="cuda")
        d_bits = torch.zeros_like(d_bits0)
        d_sig = torch.full((M, sw), -1, dtype=torch.int32, device="cuda")
        d_tips0, d_tips = torch.zeros(3 * M, dtype=torch.float64, device="cuda"), torch.zeros(3 * M, dtype=torch.float64, device="cuda")
        eng.validate_candidates_dev(seed, 0, M, d_bits0, d_tips0, box=box)
        eng.validate_candidates_sig_dev(seed, 0, M, d_bits, d_sig, d_tips, box=box)
        torch.cuda.synchronize()
        assert torch.equal(d_bits, d_bits0) and torch.equal(d_tips, d_tips0)
        cand = torch.empty(M * S, dtype=torch.float64, device="cuda")
        eng.candidate_states_dev(seed, 0, M, cand, box=box)
        d_verts = torch.empty(M * S, dtype=torch.float64, device="cuda")
        nv = eng.compact_rows_dev(d_bits, M, cand, S, d_verts, M)
        d_vsig = torch.empty((M, sw), dtype=torch.int32, device="cuda")
        assert eng.compact_rows_dev(d_bits, M, d_sig.view(torch.float64).reshape(-1), sw // 2, d_vsig.view(torch.float64).reshape(-1), M) == nv
        assert 0.3 * M < nv < M
        d_verts, d_vsig = d_verts[: nv * S], d_vsig[:nv].contiguous()
        d_edges = torch.empty((nv * k, 2), dtype=torch.int32, device="cuda")
        ne = eng.knn_edges_dev(d_verts, nv, k, d_edges)
        assert (M < 2000) == (ne < 8192)
        want_bits = torch.zeros((ne + 63) // 64, dtype=torch.int64, device="cuda")
        want_nfk = torch.zeros(ne, dtype=torch.int32, device="cuda")
        want_nd = eng.validate_edges_indexed_dev(d_verts, nv, d_edges, ne, want_bits, want_nfk)
        got_bits, got_nfk = torch.zeros_like(want_bits), torch.zeros_like(want_nfk)
        got_nd = eng.validate_edges_indexed_dev(d_verts, nv, d_edges, ne, got_bits, got_nfk, d_vertex_sig=d_vsig)
        assert torch.equal(got_bits, want_bits) and torch.equal(got_nfk, want_nfk) and got_nd == want_nd
        assert 0.05 < irt.unpack_bits(want_bits.cpu().numpy().view(np.uint64), ne).mean() < 0.999
    # a retraction robot's context has no signatures to hand over
    pull = W.robot_config2()
    pull.enable_retraction = True
    e2 = irt.VoxelBackboneValidityChecker(pull, irt.VoxelEnvironment(), vox).engine
    assert e2.signature_words() == 0
    with pytest.raises(irt.Unsupported):
        e2.validate_candidates_sig_dev(1, 0, 64, torch.zeros(1, dtype=torch.int64, device="cuda"), torch.zeros(64 * 160, dtype=torch.int32, device="cuda"))


def test_signature_rows_equal_the_cells_of_the_stored_points(irt):
    """The FK kernels agree on tips to rounding only (test_fused_and_separate_kernels_give_identical_bits), and the edge verdicts
    hang on the signature rows: the rows fk_verdict<SIG> writes for a batch of candidates are, word for word, the cells
    (collision/VoxelOctree.cpp:309-317: closed domain check, truncated quotient) of the points the stored-point kernel fk_rk4_batch
    returns for the same states -- a one-ulp difference between the kernels moves no point across a cell wall on this batch."""
    import torch
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    eng = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox).engine
    sw, S, P, box, seed, M = eng.signature_words(), eng.state_size, eng.num_points, D.sampling_box(robot), 29, 8192
    d_bits = torch.zeros(M // 64, dtype=torch.int64, device="cuda")
    d_sig = torch.full((M, sw), -1, dtype=torch.int32, device="cuda")
    eng.validate_candidates_sig_dev(seed, 0, M, d_bits, d_sig, box=box)
    cand = torch.empty(M * S, dtype=torch.float64, device="cuda")
    eng.candidate_states_dev(seed, 0, M, cand, box=box)
    torch.cuda.synchronize()
    pts = eng.fk_batch(cand.cpu().numpy().reshape(M, S))["p"]                       # [M][P][3] from the stored-point kernel
    lims = (vox.xlim(), vox.ylim(), vox.zlim())
    inside = np.ones((M, P), dtype=bool)
    word = np.zeros((M, P), dtype=np.int64)
    for a, ((lo, hi), d) in enumerate(zip(lims, (vox.dx(), vox.dy(), vox.dz()))):
        x = pts[:, :, a]
        with np.errstate(invalid="ignore"):
            inside &= ~((x < lo) | (hi < x)) & (np.abs(x) < 1e300)
        with np.errstate(invalid="ignore"):
            word |= (((x - lo) / d).astype(np.int64) & 1023) << (10 * a)
    word[~inside] = 1 << 30
    got = d_sig[:, :P].cpu().numpy().view(np.uint32).astype(np.int64)
    assert inside.mean() > 0.99 and np.array_equal(got, word)


def test_sampler_returns_the_signature_rows_of_its_vertices(irt):
    """tr_sample_valid_vertices_sig_dev: the rejection loop's accepted vertices come with their signature rows -- the rows
    tr_validate_candidates_sig_dev writes for those candidates, whatever the batch sizes and whether or not the caller keeps the
    candidate indices -- and the edge call that takes them gives the verdicts of the one that integrates the vertices itself."""
    import torch
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    eng = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox).engine
    sw, S, box, seed, n_want = eng.signature_words(), eng.state_size, D.sampling_box(robot), 23, 3000
    d_states = torch.zeros(n_want * S, dtype=torch.float64, device="cuda")
    d_index = torch.zeros(n_want, dtype=torch.int64, device="cuda")
    d_sig = torch.full((n_want, sw), -1, dtype=torch.int32, device="cuda")
    acc, tried = eng.sample_valid_vertices_dev(n_want, d_states, d_index=d_index, seed=seed, box=box, d_sig=d_sig)
    torch.cuda.synchronize()
    assert acc == n_want and tried > n_want
    plain = torch.zeros_like(d_states)
    assert eng.sample_valid_vertices_dev(n_want, plain, seed=seed, box=box) == (acc, tried) and torch.equal(plain, d_states)
    # the candidates' rows from the batch call, picked by the accepted indices
    M = (tried + 63) // 64 * 64
    d_bits = torch.zeros(M // 64, dtype=torch.int64, device="cuda")
    d_all = torch.zeros((M, sw), dtype=torch.int32, device="cuda")
    eng.validate_candidates_sig_dev(seed, 0, M, d_bits, d_all, box=box)
    torch.cuda.synchronize()
    P = eng.num_points
    assert torch.equal(d_sig[:, :P], d_all[d_index][:, :P])
    # several small batches (a cap on the candidates per call) and no index array: the same rows
    d_states2, d_sig2 = torch.zeros_like(d_states), torch.full((n_want, sw), -1, dtype=torch.int32, device="cuda")
    got, first = 0, 0
    while got < n_want:                                                  # calls capped at 1024 candidates each, run after run
        a, t = eng.sample_valid_vertices_dev(n_want - got, d_states2[got * S:], seed=seed, first_candidate=first, box=box, max_candidates=1024,
                                             d_sig=d_sig2[got:])
        assert 0 < a and t <= 1024
        got, first = got + a, first + 1024
    torch.cuda.synchronize()
    assert got == n_want and torch.equal(d_states2, d_states)            # the same candidates in the same order, so the same vertices
    assert torch.equal(d_sig2[:, :P], d_sig[:, :P])
    k = 7
    for st, sg in ((d_states, d_sig), (d_states2, d_sig2)):
        d_edges = torch.empty((n_want * k, 2), dtype=torch.int32, device="cuda")
        ne = eng.knn_edges_dev(st, n_want, k, d_edges)
        want, got_b = torch.zeros((ne + 63) // 64, dtype=torch.int64, device="cuda"), torch.zeros((ne + 63) // 64, dtype=torch.int64, device="cuda")
        wn, gn = torch.zeros(ne, dtype=torch.int32, device="cuda"), torch.zeros(ne, dtype=torch.int32, device="cuda")
        assert eng.validate_edges_indexed_dev(st, n_want, d_edges, ne, want, wn) == eng.validate_edges_indexed_dev(st, n_want, d_edges, ne, got_b, gn, d_vertex_sig=sg)
        assert torch.equal(want, got_b) and torch.equal(wn, gn) and ne >= 8192


def test_device_resident_entry_points_reject_bad_arguments(irt):
    """The C entry points added for the device-resident build return the reference's error kinds for null / inconsistent arguments
    (through the raw ctypes functions: the Python wrappers never pass these)."""
    import ctypes as C
    import torch
    L = irt._lib
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=16)
    eng = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox).engine
    lib, ctx = eng.lib, eng._ctx
    sp = L.TrSpaceParams(0.02, 0.01, 0.0001)
    n64 = C.c_int64(0)
    d = torch.zeros(4096, dtype=torch.float64, device="cuda")
    p = C.c_void_p(d.data_ptr())
    INV, OOR = L.TR_ERR_INVALID_ARG, L.TR_ERR_OUT_OF_RANGE
    assert lib.tr_knn_edges_dev(ctx, p, 10, 3, 1e300, None, 5, C.byref(n64)) == INV            # capacity without an array
    assert lib.tr_knn_edges_dev(ctx, p, 10, 3, 1e300, p, 5, None) == INV
    assert lib.tr_knn_range_dev(ctx, p, 10, 8, 3, 3, 1e300, p) == OOR                           # queries beyond the states
    assert lib.tr_knn_range_dev(ctx, p, 10, 0, 3, 3, 1e300, None) == INV
    assert lib.tr_knn_table_edges_dev(ctx, None, 10, 3, p, 30, C.byref(n64)) == INV
    assert lib.tr_validate_edges_indexed_dev(ctx, C.byref(sp), p, 10, p, 5, None, None, C.byref(n64)) == INV
    assert lib.tr_validate_edges_indexed_sig_dev(ctx, C.byref(sp), p, 10, None, p, 5, p, None, C.byref(n64)) == INV
    assert lib.tr_validate_candidates_sig_dev(ctx, 1, 0, 64, None, None, p, None, None, None) == INV
    assert lib.tr_validate_candidates_sig_dev(ctx, 1, 32, 64, None, None, p, None, p, None) == INV     # first candidate not a multiple of 64
    assert lib.tr_sample_valid_vertices_sig_dev(ctx, 1, 0, None, None, 8, 0, p, None, None, None, C.byref(n64), C.byref(n64), None) == INV
    assert lib.tr_signature_words(None) == 0
    for f in (lib.tr_knn_edges_dev, lib.tr_knn_table_edges_dev):
        assert f(None, p, 10, 3, *((1e300,) if f is lib.tr_knn_edges_dev else ()), p, 5, C.byref(n64)) == INV
    # zero sizes are no-ops
    assert lib.tr_validate_edges_indexed_dev(ctx, C.byref(sp), None, 0, None, 0, None, None, C.byref(n64)) == 0
    assert lib.tr_knn_range_dev(ctx, p, 10, 4, 0, 3, 1e300, None) == 0


def test_edge_queue_equals_the_level_synchronous_lanes(irt, orc, helpers):
    """The indexed edge check as ONE persistent launch over a device work queue with a barrier per edge (csrc/edge_queue_kernel.hpp, the
    default) gives the verdicts, the reference's FK counts and the domain-error count of the level-synchronous lanes
    (TENDON_HIP_EDGE_QUEUE=0), edge by edge: for a 3-tendon and a 4-tendon robot, a rotating robot in a rotated environment, a
    slender robot under high tension whose samples need the exact pairwise self-collision sweep (taken INSIDE the queue by the wave that
    finds them), with every sample forced through that sweep (debug bit 1), with a queue of a handful of waves (each wave's rounds
    then depend on other waves' pushes from the first level on), and when the pool is too small for the queue (it gives up, the lanes
    take the call).  tr_edge_schedule_last says which schedule ran.  The oracle's checkMotion on a sample of the edges."""
    W = irt.workloads
    thin = W.robot_config1()
    thin.specs.dL = 0.2 / 128
    thin.r = 0.008
    for t in thin.tendons:
        t.max_tension, t.min_length, t.max_length = 100.0, -1.0, 0.12
    spin = W.robot_config3()
    spin.enable_rotation = True
    a = 0.4
    rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    cases = ((W.robot_config2(), 12.0, None, 0), (W.robot_config3(), 20.0, None, 0), (spin, 20.0, rot, 0), (thin, 100.0, None, 0),
             (W.robot_config3(), 20.0, None, 2))
    for robot, tau_max, inv_rot, debug in cases:
        nv = 1500
        vox, _ = W.reach_environment(seed=7, n_spheres=48)
        env = irt.VoxelEnvironment()
        if inv_rot is not None:
            env.inv_rotation = inv_rot
        states = W.random_states(robot, nv, seed=91, tau_max=tau_max)
        nt = len(robot.tendons)
        near = np.argsort(np.linalg.norm(states[:, None, :nt] - states[None, :, :nt], axis=2), axis=1)[:, 1:8]
        edges = np.stack([np.repeat(np.arange(nv), 7), near.reshape(-1)], 1)

        def run():
            chk = irt.VoxelBackboneValidityChecker(robot, env, vox)
            if debug:
                chk.engine.set_debug(debug)
            out = irt.VoxelBackboneMotionValidator(chk).check_motion_indexed(states, edges)
            out["schedule"] = chk.engine.edge_schedule_last()
            return out

        want = _with_env(irt, {"TENDON_HIP_EDGE_QUEUE": "0"}, run)
        assert want["schedule"]["samples"] == 0
        own = int(np.asarray(want["n_fk"]).sum()) - 2 * len(edges)
        for extra in ({}, {"TENDON_HIP_EDGE_QUEUE_WAVES": "5"}, {"TENDON_HIP_EDGE_POOL": "24000"}):
            got = _with_env(irt, dict({"TENDON_HIP_EDGE_QUEUE": "1"}, **extra), run)
            for k in ("valid", "n_fk", "n_domain_errors"):
                assert np.array_equal(got[k], want[k]), (k, extra, debug, np.flatnonzero(np.asarray(got[k]) != np.asarray(want[k]))[:8])
            sch = got["schedule"]
            if "TENDON_HIP_EDGE_POOL" in extra and own + nv + 64 > 24000:
                assert sch["flags"] == 1, sch                     # the pool is too small: the queue says so and the lanes took the call
            else:
                assert sch["flags"] == 0 and sch["samples"] == own, (sch, own)
                if debug & 2:
                    assert sch["exact_sweep"] > 0.5 * own, sch    # (samples that fail before the self-collision test do not take it)
        assert 0.2 < want["valid"].mean() < 0.995 and want["n_fk"].max() > 4
        if not debug:
            idx = np.random.default_rng(92).choice(len(edges), 300, replace=False)
            ov, onf, _ = orc.check_motion_batch(helpers.oracle_robot(orc, robot, lib="omp"), helpers.oracle_grid(orc, vox), states[edges[idx, 0]],
                                                states[edges[idx, 1]], inv_rot=np.eye(3) if inv_rot is None else inv_rot, nthreads=0, lib=orc.omp_lib())
            assert np.array_equal(want["valid"][idx], ov)
            assert np.array_equal(want["n_fk"][idx][ov], onf[ov])
