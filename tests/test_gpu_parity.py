"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded
inputs.  Tolerances: FK (floating point) tip / point position <= 1e-9 m absolute (north_star's
"stated fp64 tip-position tolerance"); everything downstream of the points (validity predicate,
voxel sweep) bit-exact when fed identical points, and verdict-identical end to end on these sets.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

TIP_TOL = 1e-9      # metres
LEN_TOL = 1e-10


def _soa(p, ld):
    """(n, P, 3) -> three (P, ld) arrays."""
    n, P, _ = p.shape
    out = []
    for k in range(3):
        a = np.zeros((P, ld))
        a[:, :n] = p[:, :, k].T
        out.append(a)
    return out


def _fk_check(irt, orc, helpers, robot, states):
    orb = helpers.oracle_robot(orc, robot)
    want = orb.fk_batch(states)
    got = robot.shape_batch(states)
    assert got["p"].shape == want["p"].shape
    assert np.array_equal(got["converged"], want["converged"])
    err = np.abs(got["p"] - want["p"]).max()
    tip = np.abs(got["p"][:, -1] - want["p"][:, -1]).max()
    assert tip <= TIP_TOL and err <= TIP_TOL, (tip, err)
    assert np.abs(got["L"] - want["L"]).max() <= LEN_TOL
    assert np.abs(got["L_i"] - want["L_i"]).max() <= LEN_TOL
    assert (got["n_points"] == got["p"].shape[1]).all()
    return err


def test_fk_config1_linear_routed(irt, orc, helpers):
    """BASELINE config 1: 3-tendon linear-routed robot, 1000 random configs, FK only."""
    robot = irt.workloads.robot_config1()
    states = irt.workloads.random_states(robot, 1000, seed=42)
    err = _fk_check(irt, orc, helpers, robot, states)
    print("config1 max point error %.3g m" % err)


def test_fk_config2_helical(irt, orc, helpers):
    robot = irt.workloads.robot_config2()
    states = irt.workloads.random_states(robot, 1500, seed=43, tau_max=20.0)
    _fk_check(irt, orc, helpers, robot, states)


def test_fk_config3_quadratic_4_tendons(irt, orc, helpers):
    robot = irt.workloads.robot_config3()
    states = irt.workloads.random_states(robot, 1000, seed=44)
    _fk_check(irt, orc, helpers, robot, states)


def test_fk_rotation_and_R(irt, orc, helpers):
    robot = irt.workloads.robot_config1()
    robot.enable_rotation = True
    states = irt.workloads.random_states(robot, 300, seed=45)
    _fk_check(irt, orc, helpers, robot, states)
    orb = helpers.oracle_robot(orc, robot)
    for s in states[:5]:
        want = orb.shape(s)
        got = robot.shape(s)
        assert np.abs(got.p - want["p"]).max() <= TIP_TOL
        Rw = want["R"].reshape(-1, 3, 3).transpose(0, 2, 1)     # oracle stores column-major
        assert np.abs(got.R - Rw).max() <= 1e-8
        assert np.allclose(got.t, want["t"], rtol=0, atol=0)


def test_fk_zero_tension_is_home_shape(irt):
    """Known answer: shape(0) is the straight home shape (TendonRobot.cpp:272-292)."""
    robot = irt.workloads.robot_config2()
    res = robot.shape([0.0, 0.0, 0.0])
    assert res.converged
    assert np.abs(res.p[:, :2]).max() == 0.0
    assert np.abs(res.p[:, 2] - res.t).max() <= 1e-13
    home = robot.home_shape()
    assert np.abs(res.L_i - home.L_i).max() <= 1e-12
    assert abs(res.L - 0.2) <= 1e-13


def test_fk_odd_batch_sizes(irt, orc, helpers):
    robot = irt.workloads.robot_config1()
    orb = helpers.oracle_robot(orc, robot)
    for n in (1, 63, 64, 65, 130):
        states = irt.workloads.random_states(robot, n, seed=100 + n)
        got = robot.shape_batch(states)
        want = orb.fk_batch(states)
        assert np.abs(got["p"] - want["p"]).max() <= TIP_TOL


def _checker(irt, robot, vox, inv_rot=None):
    env = irt.VoxelEnvironment()
    if inv_rot is not None:
        env.inv_rotation = inv_rot
    return irt.VoxelBackboneValidityChecker(robot, env, vox)


def _sweep_case(irt, orc, helpers, robot, n, seed, tau_max):
    import torch
    W = irt.workloads
    vox, _ = W.reach_environment(seed=3, n_spheres=48)
    chk = _checker(irt, robot, vox)
    orb = helpers.oracle_robot(orc, robot)
    og = helpers.oracle_grid(orc, vox)
    states = W.random_states(robot, n, seed=seed, tau_max=tau_max)
    fk = orb.fk_batch(states)
    want_valid = np.zeros(n, bool)
    want_flags = np.zeros(n, np.uint8)
    for i in range(n):
        v, _, fl = orc.is_valid_state(orb, og, states[i])
        want_valid[i], want_flags[i] = v, fl
    ld = (n + 63) // 64 * 64
    N = len(robot.tendons)
    px, py, pz = (torch.from_numpy(a).cuda() for a in _soa(fk["p"], ld))
    Li = np.zeros((N, ld)); Li[:, :n] = fk["L_i"].T
    d_Li = torch.from_numpy(Li).cuda()
    d_conv = torch.from_numpy(fk["converged"].astype(np.uint8)).cuda()
    d_bits = torch.zeros(ld // 64, dtype=torch.int64, device="cuda")
    d_flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    for debug in (0, 2, 3, 4):        # default | exact skip sweep for all | brute-force pairs for all | no dilated-grid fast path
        chk.engine.set_debug(debug)
        chk.engine.validate_shapes_dev(n, ld, px, py, pz, d_Li, d_conv, d_bits, d_flags)
        torch.cuda.synchronize()
        got_valid = irt.unpack_bits(d_bits.cpu().numpy().view(np.uint64), n)
        got_flags = d_flags.cpu().numpy()
        assert np.array_equal(got_flags, want_flags), np.flatnonzero(got_flags != want_flags)[:10]
        assert np.array_equal(got_valid, want_valid)
    chk.engine.set_debug(0)
    return np.bincount(want_flags, minlength=16)


def test_sweep_bit_exact_on_oracle_points(irt, orc, helpers):
    """K2 alone on the ORACLE's fp64 points: flags and verdicts must match bit for bit, with the
    conservative-skip self-collision sweep and with the brute-force one."""
    W = irt.workloads
    # (A) straight tendons, thin robot, high tension: length limits, self collision, obstacles
    ra = W.robot_config1()
    ra.specs.dL = 0.2 / 128
    ra.r = 0.01
    for t in ra.tendons:
        t.max_tension, t.min_length, t.max_length = 100.0, -1.0, 0.08
    ha = _sweep_case(irt, orc, helpers, ra, 2500, 5, 100.0)
    # (B) helical tendons at high tension: non-converged base solves
    rb = W.robot_config2()
    for t in rb.tendons:
        t.max_tension, t.max_length = 60.0, 0.02
    hb = _sweep_case(irt, orc, helpers, rb, 1500, 6, 45.0)
    hist = ha + hb
    print("flag histogram", hist)
    # together the sets exercise every branch of the predicate
    assert hist[15] > 0 and hist[7] > 0 and hist[3] > 0 and hist[1] > 0 and hist[0] > 0


def test_validate_batch_matches_oracle(irt, orc, helpers):
    """End to end (GPU FK -> GPU sweep) against oracle verdicts and tips."""
    W = irt.workloads
    robot = W.robot_config2()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = _checker(irt, robot, vox)
    orb = helpers.oracle_robot(orc, robot)
    og = helpers.oracle_grid(orc, vox)
    states = W.random_states(robot, 20000, seed=11, tau_max=20.0)
    got = chk.is_valid_detail(states)
    want, tips, _ = orc.validate_batch(orb, og, states, nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(got["valid"], want), int((got["valid"] != want).sum())
    assert np.abs(got["tips"] - tips).max() <= TIP_TOL
    assert 0.2 < want.mean() < 0.95
    # padding bits of the last word are zero
    assert got["bits"][-1] >> np.uint64(20000 % 64) == 0 if 20000 % 64 else True


def test_validate_batch_rotated_environment(irt, orc, helpers):
    W = irt.workloads
    robot = W.robot_config2()
    vox, _ = W.reach_environment(seed=9, n_spheres=64)
    a = 0.3
    rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    chk = _checker(irt, robot, vox, rot)
    orb = helpers.oracle_robot(orc, robot)
    og = helpers.oracle_grid(orc, vox)
    states = W.random_states(robot, 4000, seed=12, tau_max=20.0)
    want, _, _ = orc.validate_batch(orb, og, states, inv_rot=rot, nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(chk.is_valid(states), want)


def test_self_collision_cases(irt, orc, helpers):
    """Tight curls: self-collision verdicts bit-exact against the oracle on the GPU's own points."""
    W = irt.workloads
    robot = W.robot_config1()
    robot.specs.dL = 0.2 / 128
    robot.r = 0.012
    for t in robot.tendons:
        t.max_tension = 80.0
        t.min_length, t.max_length = -1.0, 1.0
    vox = irt.VoxelOctree(256)
    vox.set_xlim(-0.25, 0.25); vox.set_ylim(-0.25, 0.25); vox.set_zlim(-0.25, 0.25)
    chk = _checker(irt, robot, vox)
    orb = helpers.oracle_robot(orc, robot)
    states = W.random_states(robot, 2000, seed=21, tau_max=80.0)
    got = chk.is_valid_detail(states)
    fk = robot.shape_batch(states)
    want_self = np.array([orb.collides_self(fk["p"][i]) for i in range(len(states))])
    got_self = (got["flags"] & 4) == 0
    conv_len = (got["flags"] & 3) == 3
    assert np.array_equal(got_self[conv_len], want_self[conv_len])
    assert want_self[conv_len].sum() > 20 and (~want_self[conv_len]).sum() > 20
    # TendonRobot::collides_self(shape) on single shapes, whatever the length limits are
    for t in robot.tendons:
        t.min_length, t.max_length = 0.01, 0.02                        # the home lengths themselves would fail these
    tight = W.robot_config1()
    tight.specs.dL, tight.r = robot.specs.dL, robot.r
    for t in tight.tendons:
        t.min_length, t.max_length = 0.01, 0.02
    for i in list(np.flatnonzero(want_self & conv_len)[:6]) + list(np.flatnonzero(~want_self & conv_len)[:6]):
        assert tight.collides_self(fk["p"][i]) == bool(want_self[i])
    with pytest.raises(irt.InvalidArgument):
        tight.collides_self(fk["p"][0][:50])


def test_empty_and_error_paths(irt):
    W = irt.workloads
    robot = W.robot_config2()
    vox, _ = W.reach_environment(seed=7, n_spheres=4)
    chk = _checker(irt, robot, vox)
    assert chk.is_valid(np.zeros((0, 3))).size == 0
    with pytest.raises(irt.InvalidArgument):          # "State is not the right size"
        chk.is_valid(np.zeros((4, 5)))
    coarse = irt.VoxelOctree(64)                      # voxel 7.8 mm > dL is fine; dL > voxel is not
    coarse.set_xlim(-0.25, 0.25); coarse.set_ylim(-0.25, 0.25); coarse.set_zlim(-0.25, 0.25)
    _checker(irt, robot, coarse)
    fine = irt.VoxelOctree(512)
    fine.set_xlim(-0.25, 0.25); fine.set_ylim(-0.25, 0.25); fine.set_zlim(-0.25, 0.25)
    with pytest.raises(irt.InvalidArgument):          # VoxelBackboneValidityChecker.h:37-45
        _checker(irt, robot, fine)
    with pytest.raises(irt.LengthError):
        v = irt.VoxelOctree(8)
        v.set_xlim(1.0, 1.0)


def test_tip_jacobian_batch(irt, orc, helpers):
    """Batched central-difference tip Jacobian (what levmar builds inside the reference's IK) vs the same
    differences of oracle FK calls."""
    W = irt.workloads
    robot = W.robot_config3()
    robot.enable_rotation = True
    orb = helpers.oracle_robot(orc, robot)
    st = W.random_states(robot, 40, seed=95)
    J = robot.tip_jacobian_batch(st, delta=1e-6)
    assert J.shape == (40, 3, 5)
    for i in range(0, 40, 7):
        for j in range(5):
            d = max(abs(1e-4 * st[i, j]), 1e-6)
            a, b = st[i].copy(), st[i].copy()
            a[j] -= d; b[j] += d
            want = (orb.shape(b)["p"][-1] - orb.shape(a)["p"][-1]) * (0.5 / d)
            assert np.abs(J[i, :, j] - want).max() <= 1e-6          # tip error 1e-13 m / 2e-6
    # rotation column: d tip / d theta = e_z x tip
    tips = robot.shape_batch(st)["p"][:, -1]
    assert np.abs(J[:, 0, 4] + tips[:, 1]).max() < 1e-6 and np.abs(J[:, 1, 4] - tips[:, 0]).max() < 1e-6


@pytest.mark.parametrize("n_tendons", [1, 2, 5, 6, 8])
def test_fk_other_tendon_counts(irt, orc, helpers, n_tendons):
    """Every instantiated kernel width (1..8 tendons), with rotation, mixed routing."""
    rng = np.random.default_rng(n_tendons)
    tendons = [irt.TendonSpecs(C=[2 * np.pi * k / n_tendons, float(rng.uniform(-6, 6)), float(rng.uniform(-10, 10))],
                               D=[0.01, float(rng.uniform(-0.01, 0.01))], max_tension=12.0) for k in range(n_tendons)]
    robot = irt.TendonRobot(tendons=tendons, specs=irt.BackboneSpecs(dL=0.004), enable_rotation=True)
    states = irt.workloads.random_states(robot, 200, seed=7 + n_tendons, tau_max=12.0 / np.sqrt(n_tendons))
    _fk_check(irt, orc, helpers, robot, states)


@pytest.mark.parametrize("n_tendons", [1, 2, 5, 7, 8])
def test_validity_other_tendon_counts_against_oracle(irt, orc, helpers, n_tendons):
    """fk_verdict (and its fallback pass) in every instantiated width, with rotation and a rotated environment: verdicts
    and flags equal the oracle's isValid, state by state."""
    rng = np.random.default_rng(40 + n_tendons)
    tendons = [irt.TendonSpecs(C=[2 * np.pi * k / n_tendons, float(rng.uniform(-6, 6))], D=[0.01, float(rng.uniform(-0.01, 0.01))],
                               max_tension=12.0) for k in range(n_tendons)]
    robot = irt.TendonRobot(tendons=tendons, specs=irt.BackboneSpecs(dL=0.2 / 128), enable_rotation=True)
    vox, _ = irt.workloads.reach_environment(seed=5, n_spheres=48)
    env = irt.VoxelEnvironment()
    a = -0.7
    env.inv_rotation = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    chk = irt.VoxelBackboneValidityChecker(robot, env, vox)
    states = irt.workloads.random_states(robot, 400, seed=9 + n_tendons, tau_max=22.0 / np.sqrt(n_tendons))
    got = chk.is_valid_detail(states)
    orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
    want = [orc.is_valid_state(orb, og, s, env.inv_rotation) for s in states]
    assert np.array_equal(got["valid"], [w[0] for w in want])
    assert np.array_equal(got["flags"] & 15, [w[2] for w in want])
    assert np.abs(got["tips"] - np.array([w[1] for w in want])).max() <= 1e-9
    assert 0.05 < got["valid"].mean() < 0.99


def test_fk_step_not_dividing_length(irt, orc, helpers):
    """L is not a multiple of dL: with dL = 3.5 mm the first interval is 4 mm = 1.14 dL and takes two RK4
    steps (integrate_times); with dL = 3 mm it is 2 mm (one short step).  Both run on the shared-grid
    kernel's host-built step list."""
    robot = irt.workloads.robot_config2()
    robot.specs.dL = 0.003
    _fk_check(irt, orc, helpers, robot, irt.workloads.random_states(robot, 200, seed=16, tau_max=15.0))
    robot = irt.workloads.robot_config2()
    robot.specs.dL = 0.0035
    orb = helpers.oracle_robot(orc, robot)
    t = orb.t_range()
    assert t[1] - t[0] > 0.0035 * 1.1 and len(t) == 58
    _fk_check(irt, orc, helpers, robot, irt.workloads.random_states(robot, 300, seed=17, tau_max=15.0))


@pytest.mark.parametrize("N,lim", [(128, (-0.25, 0.25, -0.25, 0.25, -0.05, 0.45)), (64, (-0.3, 0.3, -0.2, 0.4, -0.1, 0.3)),
                                   (512, (-0.4, 0.4, -0.4, 0.4, -0.4, 0.4))])
def test_validity_other_grids(irt, orc, helpers, N, lim):
    """Grid sizes other than 256 and anisotropic voxels (dx != dy != dz)."""
    W = irt.workloads
    robot = W.robot_config2()
    vox = irt.VoxelOctree(N)
    vox.set_xlim(lim[0], lim[1]); vox.set_ylim(lim[2], lim[3]); vox.set_zlim(lim[4], lim[5])
    rng = np.random.default_rng(N)
    for _ in range(40):
        c = rng.uniform(-0.2, 0.2, 3)
        c[2] = abs(c[2])
        if np.hypot(c[0], c[1]) > 0.05:
            vox.add_sphere(c, 0.02)
    chk = _checker(irt, robot, vox)
    states = W.random_states(robot, 3000, seed=N, tau_max=20.0)
    want, _, _ = orc.validate_batch(helpers.oracle_robot(orc, robot, lib="omp"), helpers.oracle_grid(orc, vox), states,
                                    nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(chk.is_valid(states), want)
    assert 0.1 < want.mean() < 0.95


def test_concurrent_single_state_calls(irt, orc, helpers):
    """isValid from many host threads on ONE context (the reference calls it from OpenMP threads,
    VoxelCachedLazyPRM.cpp:1448-1455): calls serialise on the context mutex and stay correct."""
    import threading
    W = irt.workloads
    robot = W.robot_config2()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = _checker(irt, robot, vox)
    states = W.random_states(robot, 256, seed=23, tau_max=20.0)
    want = chk.is_valid(states)
    got = np.zeros(len(states), bool)

    def work(lo, hi):
        for i in range(lo, hi):
            got[i] = chk.isValid(states[i])
            if i % 16 == 0:                                  # interleave a different entry point
                robot.shape_batch(states[i:i + 3])
    ths = [threading.Thread(target=work, args=(k * 32, (k + 1) * 32)) for k in range(8)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert np.array_equal(got, want) and 0 < want.sum() < len(want)


def test_error_paths_of_the_batch_entry_points(irt):
    """Status codes of the C ABI mapped to the reference's exception types: calls that need a grid before
    tr_set_grid, size mismatches, bad parameters -- none of them may touch the device state."""
    W = irt.workloads
    robot = W.robot_config2()
    eng = irt.Engine(robot)                            # no grid yet
    st = W.random_states(robot, 8, seed=1)
    for call in (lambda: eng.validate_batch(st, False, False), lambda: eng.validate_edges(st[:4], st[4:]),
                 lambda: eng.validate_edges_discrete(st[:4], st[4:]), lambda: eng.voxelize_batch(st),
                 lambda: eng.grid_add_spheres([[0, 0, 0, 0.01]]), lambda: eng.grid_dilate(1)):
        with pytest.raises(irt.InvalidArgument):
            call()
    assert eng.fk_batch(st)["p"].shape == (8, eng.num_points, 3)          # FK needs no grid
    vox, _ = W.reach_environment(seed=7, n_spheres=4)
    eng.set_grid(vox.Nx(), vox.limits(), vox.blocks)
    with pytest.raises(irt.InvalidArgument):
        eng.validate_edges(st[:4], st[4:], min_tension_change=0.0)       # would divide by a zero segment length
    with pytest.raises(irt.InvalidArgument):
        eng.validate_edges(st[:4], st[3:])                                # different sizes
    with pytest.raises(irt.InvalidArgument):
        eng.knn(st, 0)
    with pytest.raises(irt.InvalidArgument):
        irt._lib.check(eng._ctx, eng.lib.tr_set_checker(eng._ctx, 7))                    # unknown checker
    assert eng.validate_edges(np.zeros((0, 3)), np.zeros((0, 3)))["valid"].size == 0
    assert eng.validate_edges_discrete(np.zeros((0, 3)), np.zeros((0, 3)))["valid"].size == 0
    out = eng.voxelize_edges(np.zeros((0, 3)), np.zeros((0, 3)))
    assert out["offsets"].tolist() == [0] and out["block_ids"].size == 0
    # after all that the context still works
    assert eng.validate_batch(st, False, False)["valid"].shape == (8,)


def test_contexts_on_concurrent_host_threads(irt):
    """One context per host thread, every family of call at once (ctypes releases the GIL for the duration of a call): the
    contexts share nothing -- results equal those of the same work done one thread after the other."""
    import threading
    W = irt.workloads
    vox, _ = W.reach_environment(seed=7, n_spheres=64)

    def work(t):
        robot = W.robot_config3() if t % 2 else W.robot_config2()
        chk = (irt.VoxelValidityChecker if t == 3 else irt.VoxelBackboneValidityChecker)(robot, irt.VoxelEnvironment(), vox)
        mv = irt.VoxelBackboneMotionValidator(chk)
        rb = irt.RoadmapBuilder(chk, mv, seed=t)
        st = W.random_states(robot, 6000, seed=40 + t, tau_max=12.0)
        out = []
        for _ in range(3):
            v = chk.is_valid(st)
            e = rb.knn_edges_gpu(st, 7)[:6000]
            ev, nf = rb.validate_edges(st, e)
            e_ok, ec = rb.connect(st, e[:1500])
            prm = irt.VoxelCachedLazyPRM(chk, st, e_ok)
            prm.set_caches(rb.vertex_caches(st), ec)
            q = prm.solveWithRoadmap(np.arange(100), np.arange(100, 200), n_threads=2)
            out.append((v.tobytes(), e.tobytes(), ev.tobytes(), nf.tobytes(), e_ok.tobytes(), ec["masks"].tobytes(), q["status"].tobytes(), q["cost"].tobytes()))
        assert out[0] == out[1] == out[2]
        return out[0]

    serial = [work(t) for t in range(4)]
    res, err = [None] * 4, []

    def run(t):
        try:
            res[t] = work(t)
        except Exception as e:                                   # noqa: BLE001
            err.append((t, repr(e)))
    ths = [threading.Thread(target=run, args=(t,)) for t in range(4)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    assert not err, err
    assert res == serial
