"""GPU parity for retraction-enabled robots (per-configuration arc-length grid, variable point
count, two-step first interval, per-configuration home lengths) against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TIP_TOL = 1e-9


def _robot(irt, kind):
    W = irt.workloads
    r = {"straight": W.robot_config1, "helix": W.robot_config2, "quad": W.robot_config3}[kind]()
    r.specs.dL = 0.2 / 128
    r.enable_retraction = True
    return r


def _states(irt, robot, n, seed):
    st = irt.workloads.random_states(robot, n, seed=seed, tau_max=15.0)
    L, dL = robot.specs.L, robot.specs.dL
    special = [0.0, L, L + 0.01, L - dL / 4, L - dL / 2, L - 0.75 * dL, L - 1.25 * dL, L - 1.5 * dL, 0.0975, 0.1, dL / 3,
               L - 2.49 * dL, 17 * dL, 17.5 * dL]
    st[:len(special), -1] = special
    return st


@pytest.mark.parametrize("kind", ["straight", "helix", "quad"])
def test_fk_with_retraction(irt, orc, helpers, kind):
    robot = _robot(irt, kind)
    orb = helpers.oracle_robot(orc, robot)
    st = _states(irt, robot, 700, seed=61)
    got = robot.shape_batch(st)
    want = orb.fk_batch(st)
    P = got["p"].shape[1]
    wp = want["p"][:, :P]
    want_n = (~np.isnan(wp[:, :, 0])).sum(1)
    assert np.array_equal(got["n_points"], want_n)
    assert np.array_equal(np.isnan(got["p"]), np.isnan(wp))
    assert np.nanmax(np.abs(got["p"] - wp)) <= TIP_TOL
    assert np.abs(got["L"] - want["L"]).max() <= 1e-10 and np.abs(got["L_i"] - want["L_i"]).max() <= 1e-10
    assert np.array_equal(got["converged"], want["converged"])
    assert got["n_points"].min() == 1 and got["n_points"].max() == P
    # single-state API: t grid and home shape
    for s in st[8:12]:
        res, w = robot.shape(s), orb.shape(s)
        assert np.array_equal(res.t, w["t"]) and np.abs(res.p - w["p"]).max() <= TIP_TOL
        assert np.abs(robot.home_shape(s[-1]).L_i - orb.home_shape(s[-1])["L_i"]).max() <= 1e-14


def test_negative_retraction_is_reported_unconverged(irt):
    robot = _robot(irt, "helix")
    out = robot.shape_batch(np.array([[1.0, 2.0, 3.0, -0.01]]))
    assert not out["converged"][0] and out["n_points"][0] == 1


@pytest.mark.parametrize("kind", ["helix", "quad"])
def test_validity_with_retraction(irt, orc, helpers, kind):
    W = irt.workloads
    robot = _robot(irt, kind)
    for t in robot.tendons:
        t.max_length = 0.02
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    st = _states(irt, robot, 6000, seed=62)
    got = chk.is_valid_detail(st)
    want, tips, _ = orc.validate_batch(helpers.oracle_robot(orc, robot, lib="omp"), helpers.oracle_grid(orc, vox), st,
                                       nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(got["valid"], want), np.flatnonzero(got["valid"] != want)[:10]
    assert np.abs(got["tips"] - tips).max() <= TIP_TOL
    hist = np.bincount(got["flags"], minlength=16)
    assert hist[15] > 0 and hist[7] > 0 and hist[1] > 0          # valid, voxel hit, length-limit failures


def test_edges_with_retraction(irt, orc, helpers):
    W = irt.workloads
    robot = _robot(irt, "helix")
    vox, _ = W.reach_environment(seed=7, n_spheres=48)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    mv = irt.VoxelBackboneMotionValidator(chk)
    rng = np.random.default_rng(63)
    a = W.random_states(robot, 500, seed=63, tau_max=15.0)
    b = a + rng.normal(size=a.shape) * np.array([1.0, 1.0, 1.0, 0.01])
    b[:, :3] = np.clip(b[:, :3], 0, 20)
    b[:, 3] = np.clip(b[:, 3], 0, 0.2)
    got = mv.check_motion_detail(a, b)
    want, nfk, _ = orc.check_motion_batch(helpers.oracle_robot(orc, robot, lib="omp"), helpers.oracle_grid(orc, vox), a, b,
                                          nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(got["valid"], want)
    assert np.array_equal(got["n_fk"][want], nfk[want])
    assert 0.05 < want.mean() < 0.98


def test_discrete_edges_and_last_valid_with_retraction(irt, orc, helpers):
    W = irt.workloads
    robot = _robot(irt, "helix")
    vox, _ = W.reach_environment(seed=7, n_spheres=48)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rng = np.random.default_rng(64)
    a = W.random_states(robot, 120, seed=64, tau_max=15.0)
    b = a + rng.normal(size=a.shape) * np.array([0.6, 0.6, 0.6, 0.004])
    b[:, :3] = np.clip(b[:, :3], 0, 20)
    b[:, 3] = np.clip(b[:, 3], 0, 0.2)
    orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
    d = irt.VoxelBackboneDiscreteMotionValidator(chk).check_motion_detail(a, b, last_valid=True)
    valid, lvt = irt.VoxelBackboneMotionValidator(chk).check_motion_last_valid(a, b)
    for i in range(len(a)):
        w = orc.check_motion_discrete(orb, og, a[i], b[i], until_invalid=True)
        assert (d["valid"][i], d["last_valid_t"][i], d["n_fk"][i]) == (w["is_fully_valid"], w["last_valid_t"], w["n_fk"]), (i, w)
        u = orc.check_motion_until_invalid(orb, og, a[i], b[i])
        assert valid[i] == u["is_fully_valid"] and lvt[i] == u["last_valid_t"], (i, u)
    assert 0.05 < d["valid"].mean() < 0.98 and d["n_fk"].max() > 20


def test_one_point_backbones_and_repeatability(irt, orc, helpers):
    """s_start within dL/2 of L gives a one-point backbone without being the s_start == L early return
    (t_range pushes only `end`); such lanes take no RK4 interval at all.  Results must also be bit-identical
    from call to call (a stray store would show up as a changing neighbour)."""
    robot = _robot(irt, "helix")
    L, dL = robot.specs.L, robot.specs.dL
    st = irt.workloads.random_states(robot, 700, seed=62, tau_max=15.0)
    st[::7, -1] = np.linspace(L - 0.49 * dL, L - 1e-9, len(st[::7]))
    outs = [robot.shape_batch(st) for _ in range(6)]
    for o in outs[1:]:
        for k in ("L", "L_i", "converged", "n_points"):
            assert np.array_equal(outs[0][k], o[k]), k
        assert np.array_equal(np.nan_to_num(outs[0]["p"]), np.nan_to_num(o["p"]))
    got = outs[0]
    assert (got["n_points"][::7] == 1).all() and (got["L"][::7] == 0).all() and (got["L_i"][::7] == 0).all()
    want = helpers.oracle_robot(orc, robot).fk_batch(st)
    P = got["p"].shape[1]
    assert np.array_equal(np.isnan(got["p"]), np.isnan(want["p"][:, :P]))
    assert np.nanmax(np.abs(got["p"] - want["p"][:, :P])) <= TIP_TOL
    assert np.abs(got["L_i"] - want["L_i"]).max() <= 1e-10 and np.array_equal(got["converged"], want["converged"])


@pytest.mark.parametrize("n_tendons", [1, 2, 5, 7, 8])
def test_retraction_other_tendon_counts(irt, orc, helpers, n_tendons):
    """Every instantiated width of the retraction kernel (two waves per SIMD up to 3 tendons, one beyond), with
    rotation, general routing (numerically integrated home lengths) and dL not dividing L (two-step first
    interval of the shared grid)."""
    rng = np.random.default_rng(40 + n_tendons)
    tendons = [irt.TendonSpecs(C=[2 * np.pi * k / n_tendons, float(rng.uniform(-6, 6)), float(rng.uniform(-10, 10))],
                               D=[0.01, float(rng.uniform(-0.01, 0.01))], max_tension=12.0) for k in range(n_tendons)]
    robot = irt.TendonRobot(tendons=tendons, specs=irt.BackboneSpecs(dL=0.0035), enable_rotation=True, enable_retraction=True)
    st = irt.workloads.random_states(robot, 300, seed=50 + n_tendons, tau_max=12.0 / np.sqrt(n_tendons))
    L, dL = robot.specs.L, robot.specs.dL
    st[:8, -1] = [0.0, L, L - dL / 4, L - 0.75 * dL, L - 1.25 * dL, dL / 3, 0.1, 17.5 * dL]
    got = robot.shape_batch(st)
    want = helpers.oracle_robot(orc, robot).fk_batch(st)
    P = got["p"].shape[1]
    assert np.array_equal(np.isnan(got["p"]), np.isnan(want["p"][:, :P]))
    assert np.nanmax(np.abs(got["p"] - want["p"][:, :P])) <= TIP_TOL
    assert np.abs(got["L_i"] - want["L_i"]).max() <= 1e-10 and np.array_equal(got["converged"], want["converged"])
    # validity (home lengths per configuration feed the length limits)
    vox, _ = irt.workloads.reach_environment(seed=7, n_spheres=48, N=128)      # 3.9 mm voxels >= dL
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    det = chk.is_valid_detail(st)
    orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
    w = [orc.is_valid_state(orb, og, s) for s in st]
    assert np.array_equal(det["valid"], [x[0] for x in w]) and np.array_equal(det["flags"] & 15, [x[2] for x in w])


def test_device_resident_two_stage_path_with_retraction(irt):
    """tr_fk_batch_retraction_dev -> tr_validate_shapes_retraction_dev (K1r, K2 on caller-owned buffers: per-configuration
    point counts, home lengths, rows aligned at the tip) equals tr_validate_batch (the verdict-only kernel: same RK4 body
    inside another kernel, tips to rounding)."""
    import torch
    W = irt.workloads
    robot = _robot(irt, "quad")
    vox, _ = W.reach_environment(seed=7, n_spheres=48)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    eng = chk.engine
    n, ld = 1000, 1024
    st = _states(irt, robot, n, seed=66)
    want = chk.is_valid_detail(st)
    P, N = eng.num_points, eng.n_tendons
    dev = "cuda"
    d_st = torch.from_numpy(st).to(dev)
    px, py, pz = (torch.zeros(P * ld, dtype=torch.float64, device=dev) for _ in range(3))
    Li, hLi = torch.zeros(N * ld, dtype=torch.float64, device=dev), torch.zeros(N * ld, dtype=torch.float64, device=dev)
    conv = torch.zeros(n, dtype=torch.uint8, device=dev)
    npts = torch.zeros(n, dtype=torch.int32, device=dev)
    bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
    flags = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.fk_batch_retraction_dev(d_st, n, ld, px, py, pz, Li, conv, npts, hLi)
    eng.validate_shapes_retraction_dev(n, ld, px, py, pz, npts, Li, hLi, conv, bits, flags)
    torch.cuda.synchronize()
    got = irt.unpack_bits(bits.cpu().numpy().view(np.uint64), n)
    assert np.array_equal(got, want["valid"]) and np.array_equal(flags.cpu().numpy(), want["flags"])
    # the tip is always in the last row
    tips = np.stack([px.view(P, ld)[P - 1, :n].cpu().numpy(), py.view(P, ld)[P - 1, :n].cpu().numpy(), pz.view(P, ld)[P - 1, :n].cpu().numpy()], 1)
    assert np.abs(tips - want["tips"]).max() <= 1e-12
    with pytest.raises(irt.Unsupported):
        eng.validate_shapes_dev(n, ld, px, py, pz, Li, conv, bits)


def _with_env(env, fn):
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


@pytest.mark.parametrize("checker", ["backbone", "spheres"])
def test_verdict_only_kernel_with_retraction_all_branches(irt, checker):
    """fk_verdict_retract (K1r's body with the sweep in its point hook: tip-aligned rows, milestones laid out from the tip,
    per-lane point counts and home lengths; fallback pass fk_sweep_retract_list) against K1r -> K2 (-> K8) on stored points,
    where the rarer branches are busy: self collisions through a 64-column fallback workspace, length limits,
    non-converged solves, one-point and two-point backbones, negative retraction, rotation, a rotated environment, points
    outside the voxel domain, and the debug switches -- with the batch in arrival order and ordered by backbone length."""
    W = irt.workloads

    def ret(r, rot=False):
        r.specs.dL = 0.2 / 128
        r.enable_retraction = True
        r.enable_rotation = rot
        return r

    thin = ret(W.robot_config1())
    thin.r = 0.01
    for t in thin.tendons:
        t.max_tension, t.min_length, t.max_length = 100.0, -1.0, 0.08
    hard = ret(W.robot_config2())
    for t in hard.tendons:
        t.max_tension, t.max_length = 60.0, 0.02
    spin = ret(W.robot_config2(), rot=True)
    quad = ret(W.robot_config3(), rot=True)
    small, _ = W.sphere_environment(seed=5, n_spheres=40, radius=0.01, N=128, half=0.12, keepout=0.02)   # the robot reaches out of it
    a = 0.4
    rot = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    cases = [(thin, 100.0, None, None, 6000), (hard, 45.0, None, None, 4000), (spin, 12.0, rot, None, 5000), (quad, 20.0, rot, None, 4000),
             (ret(W.robot_config2()), 14.0, None, small, 5000)]
    cls = irt.VoxelBackboneValidityChecker if checker == "backbone" else irt.VoxelValidityChecker
    seen = np.zeros(32, int)
    for robot, tau_max, inv_rot, vox, n in cases:
        if vox is None:
            vox, _ = W.reach_environment(seed=3, n_spheres=48)
        env = irt.VoxelEnvironment()
        if inv_rot is not None:
            env.inv_rotation = inv_rot
        states = W.random_states(robot, n + 21, seed=23, tau_max=tau_max)
        L, dL = robot.specs.L, robot.specs.dL
        special = [0.0, L, L + 0.01, L - dL / 4, L - dL / 2, L - 0.75 * dL, L - 1.25 * dL, L - 1.5 * dL, 0.0975, 0.1, dL / 3,
                   L - 2.49 * dL, 17 * dL, 17.5 * dL, -0.01, L - 3.2 * dL]
        states[:len(special), -1] = special
        states[100:400, -1] *= 0.1                                   # long backbones: most of the self collisions

        def run(debug=0, detail=True):
            chk = cls(robot, env, vox)
            chk.engine.set_debug(debug)
            return chk.is_valid_detail(states) if detail else dict(valid=chk.is_valid(states))

        want = _with_env({"TENDON_HIP_FUSED": "0"}, run)
        for debug in ((0, 2, 3, 4) if checker == "backbone" else (0, 2, 3)):
            # in arrival order, and ordered by backbone length (what batches of 8192 or more get by default)
            for order in ("0", "64"):
                got = _with_env({"TENDON_HIP_FB_CAP": "64", "TENDON_HIP_RETRACT_SORT": order}, lambda: run(debug))
                for k in ("valid", "flags"):
                    assert np.array_equal(got[k], want[k]), (k, debug, order, np.flatnonzero(got[k] != want[k])[:8], got[k][got[k] != want[k]][:8],
                                                              want[k][got[k] != want[k]][:8])
                ok = want["flags"] & 1 > 0
                assert np.abs(got["tips"][ok] - want["tips"][ok]).max() <= 1e-12
        assert np.array_equal(run(0, detail=False)["valid"], want["valid"])
        seen += np.bincount(want["flags"], minlength=32)
    print("flag histogram", seen)
    assert seen[15] > 0 and seen[7] > 0 and seen[3] > 0 and seen[1] > 0 and seen[0] > 0


def test_edges_with_retraction_where_samples_need_the_fallback_pass(irt, orc, helpers):
    """Edge samples of retraction robots go through fk_verdict_retract, whose fallback pass (exact self-collision sweep)
    re-integrates its few configurations in the first columns of the point workspace.  The interval test's per-sample point
    counts must not live there (they did: the counts of the pool's first samples were overwritten and edge 0 subdivided
    21 times instead of 8).  A slender robot under high tension curls onto itself, so many samples take the fallback."""
    W = irt.workloads
    robot = W.robot_config1()
    robot.specs.dL = 0.2 / 128
    robot.r = 0.008
    robot.enable_retraction = True
    for t in robot.tendons:
        t.max_tension, t.min_length, t.max_length = 100.0, -1.0, 0.12
    vox, _ = W.reach_environment(seed=3, n_spheres=12)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    mv = irt.VoxelBackboneMotionValidator(chk)
    rng = np.random.default_rng(71)
    m = 400
    a = W.random_states(robot, m, seed=72, tau_max=100.0)
    a[:, -1] = rng.uniform(0.0, 0.1, m)
    b = a + rng.normal(0, 1.5, a.shape)
    b[:, :3] = np.clip(b[:, :3], 0, 100.0)
    b[:, -1] = np.clip(a[:, -1] + rng.normal(0, 0.01, m), 0, 0.2)
    det = chk.is_valid_detail(np.concatenate([a, b]))
    assert ((det["flags"] & 7) == 3).sum() > 60                      # end states in self collision: each one a fallback entry
    got = mv.check_motion_detail(a, b)
    want, nfk, _ = orc.check_motion_batch(helpers.oracle_robot(orc, robot, lib="omp"), helpers.oracle_grid(orc, vox), a, b,
                                          nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(got["valid"], want), np.flatnonzero(got["valid"] != want)[:10]
    assert np.array_equal(got["n_fk"][want], nfk[want]), np.flatnonzero((got["n_fk"] != nfk) & want)[:10]
    assert 0.3 < want.mean() < 0.95


@pytest.mark.parametrize("kind,rot", [("quad", True), ("helix", False)])
def test_stored_point_fk_in_length_order_equals_arrival_order(irt, orc, helpers, kind, rot):
    """tr_fk_batch / tr_voxelize_batch for a retraction robot integrate a batch of 8 192 or more in the order of its backbone lengths
    (fk_rk4_batch_retract with the order of retraction_order: a wave then holds backbones of one length) and scatter every stored
    output to its configuration's own column: points, R, L, L_i, converged, point counts -- bit for bit those of the arrival-order
    launch (TENDON_HIP_RETRACT_SORT=0), and the oracle's on a sample; the voxel sets built from those planes are the same too."""
    robot = _robot(irt, kind)
    robot.enable_rotation = rot
    n = 20000 + 37
    st = _states(irt, robot, n, seed=41)
    W = irt.workloads
    vox, _ = W.reach_environment(seed=3, n_spheres=48)

    def run():
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        fk = chk.engine.fk_batch(st, want_R=True)
        vc = chk.engine.voxelize_batch(st)
        return fk, vc

    fk0, vc0 = _with_env({"TENDON_HIP_RETRACT_SORT": "0"}, run)
    fk1, vc1 = run()                                                # the default: ordered from 8 192 configurations on
    for k in ("p", "R", "L", "L_i", "converged", "n_points"):
        a, b = fk0[k], fk1[k]
        assert a.shape == b.shape and np.array_equal(a, b, equal_nan=True), k
    for k in ("offsets", "block_ids", "masks", "shape_valid"):
        assert np.array_equal(vc0[k], vc1[k]), k
    assert len(set(fk1["n_points"].tolist())) > 100                 # every backbone length is there
    orb = helpers.oracle_robot(orc, robot)
    for i in list(range(16)) + list(np.random.default_rng(2).choice(n, 60, replace=False)):
        want = orb.shape(st[i])
        m = len(want["p"])
        if st[i, -1] < 0:
            continue
        assert fk1["n_points"][i] == m and bool(fk1["converged"][i]) == bool(want["converged"])
        if want["converged"]:
            assert np.abs(fk1["p"][i, :m] - want["p"]).max() <= TIP_TOL and np.abs(fk1["L_i"][i] - want["L_i"]).max() <= 1e-10
