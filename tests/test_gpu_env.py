"""Environment preparation on the resident grid (tr_grid_*: VoxelOctree::add_sphere, dilate_6/27neighbor,
dilate_sphere, remove_interior_6/27neighbor) -- bit-exact against the oracle's restatement, and the
edited obstacle set is the one K2 then tests against."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _checker(irt, N=256, lim=0.25):
    robot = irt.workloads.robot_config2()
    vox = irt.VoxelOctree(N)
    vox.set_xlim(-lim, lim); vox.set_ylim(-lim, lim); vox.set_zlim(-lim, lim)
    return robot, vox, irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)


def _oracle_grid(orc, vox):
    g = orc.Grid(vox.Nx(), vox.limits())
    g.blocks()[...] = vox.blocks
    return g


def test_add_capsules_matches_oracle(irt, orc):
    """tr_grid_add_capsules = VoxelOctree::add_capsule (VoxelOctree.cpp:471-515), bit for bit; with add_spheres it serves
    everything Environment::voxelize rasterises (motion-planning/Environment.cpp:62-100)."""
    robot, vox, chk = _checker(irt)
    rng = np.random.default_rng(8)
    cp = np.column_stack([rng.uniform(-0.3, 0.3, (150, 3)), rng.uniform(-0.3, 0.3, (150, 3)), rng.uniform(0.0005, 0.03, 150)])
    cp[0, 3:6] = cp[0, :3]                            # a == b: a sphere
    cp[1] = [0.0, 0.0, 0.0, 0.0, 0.0, 0.1, 0.0]       # radius 0: the end points' cells and the centres exactly on the axis
    cp[2] = [0.24, 0.0, 0.0, 0.6, 0.0, 0.0, 0.02]     # leaves the domain
    cp[3] = [1.0, 1.0, 1.0, 2.0, 1.0, 1.0, 0.1]       # entirely outside
    chk.add_capsules(cp)
    g = _oracle_grid(orc, vox)
    for row in cp:
        g.add_capsule(row[:3], row[3:6], row[6])
    got = chk.engine.get_grid()
    assert np.array_equal(got, np.asarray(g.blocks()).ravel())
    assert np.count_nonzero(got) > 1000
    chk.add_capsules(np.zeros((0, 7)))
    assert np.array_equal(chk.engine.get_grid(), got)
    # the host mirror builds the same set
    v = vox.empty_copy()
    v.blocks[...] = vox.blocks
    for row in cp[:20]:
        v.add_capsule(row[:3], row[3:6], row[6])
    g2 = _oracle_grid(orc, vox)
    for row in cp[:20]:
        g2.add_capsule(row[:3], row[3:6], row[6])
    assert np.array_equal(v.blocks.ravel(), np.asarray(g2.blocks()).ravel())


def test_add_spheres_matches_oracle(irt, orc):
    robot, vox, chk = _checker(irt)
    rng = np.random.default_rng(3)
    sp = np.column_stack([rng.uniform(-0.3, 0.3, (200, 3)), rng.uniform(0.0005, 0.04, 200)])
    sp[0] = [0.0, 0.0, 0.0, 0.0]                      # radius 0: only add_point(centre)
    sp[1] = [0.2499, -0.2499, 0.25, 0.01]             # on the closed upper face
    sp[2] = [0.4, 0.0, 0.0, 0.2]                      # centre outside, ball reaches in
    sp[3] = [1.0, 1.0, 1.0, 0.1]                      # entirely outside (clamped block range, no cell inside)
    chk.add_spheres(sp)
    g = _oracle_grid(orc, vox)
    for row in sp:
        g.add_sphere(row[:3], row[3])
    got = chk.engine.get_grid()
    assert np.array_equal(got, np.asarray(g.blocks()).ravel())
    assert np.count_nonzero(got) > 1000
    chk.add_spheres(np.zeros((0, 4)))                 # empty batch is a no-op
    assert np.array_equal(chk.engine.get_grid(), got)


@pytest.mark.parametrize("N", [64, 256])
def test_dilate_and_remove_interior_match_oracle(irt, orc, N):
    robot, vox, chk = _checker(irt, N)
    rng = np.random.default_rng(4)
    sp = np.column_stack([rng.uniform(-0.25, 0.25, (24, 3)), rng.uniform(0.01, 0.06, 24)])
    chk.add_spheres(sp)
    g = _oracle_grid(orc, vox)
    for row in sp:
        g.add_sphere(row[:3], row[3])
    for cell in ((0, 0, 0), (N - 1, N - 1, N - 1), (0, N // 2, N - 1)):       # grid corners / faces
        g.set_cell(*cell)
    vox2 = irt.VoxelOctree(N); vox2.set_xlim(-0.25, 0.25); vox2.set_ylim(-0.25, 0.25); vox2.set_zlim(-0.25, 0.25)
    vox2.blocks[...] = np.asarray(g.blocks()).reshape(vox2.blocks.shape)
    base = vox2.blocks.copy()
    for op, args in (("dilate", (1, False)), ("dilate", (3, False)), ("dilate", (6, False)), ("dilate", (1, True)),
                     ("dilate", (5, True)), ("remove_interior", (True,)), ("remove_interior", (False,)),
                     ("dilate_sphere", (0.015,))):
        chk.engine.set_grid(N, vox2.limits(), base, None)
        getattr(chk, op)(*args)
        h = orc.Grid(N, vox2.limits()); h.blocks()[...] = base.reshape(np.asarray(h.blocks()).shape)
        getattr(h, op)(*args)
        got = chk.engine.get_grid()
        assert np.array_equal(got, np.asarray(h.blocks()).ravel()), (op, args, int(np.count_nonzero(got != np.asarray(h.blocks()).ravel())))
        assert not np.array_equal(got, base.ravel())


def test_edited_obstacles_are_what_validity_sees(irt, orc, helpers):
    """prepare -> validate without the grid leaving the device: dilate the obstacles by the robot radius
    (dilate_sphere) and compare verdicts with the oracle on the same edited grid."""
    W = irt.workloads
    robot, vox, chk = _checker(irt)
    _, centres = W.reach_environment(seed=7, n_spheres=48)
    sp = np.column_stack([centres, np.full(len(centres), 0.006)])
    chk.add_spheres(sp)
    chk.dilate_sphere(robot.r)
    g = _oracle_grid(orc, vox)
    for row in sp:
        g.add_sphere(row[:3], row[3])
    g.dilate_sphere(robot.r)
    assert np.array_equal(chk.engine.get_grid(), np.asarray(g.blocks()).ravel())
    states = W.random_states(robot, 4096, seed=77, tau_max=12.0)
    got = chk.is_valid(states)
    want, _, _ = orc.validate_batch(helpers.oracle_robot(orc, robot), g, states, nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(got, want) and 0.05 < want.mean() < 0.95
    assert np.array_equal(chk.obstacles().blocks.ravel(), np.asarray(g.blocks()).ravel())


def test_environment_edits_on_an_anisotropic_grid(irt, orc):
    """dx != dy != dz: nearest_block_idx / voxel centres per axis, dilate_sphere's min(dx, dy, dz)."""
    robot = irt.workloads.robot_config1()              # dL = 5 mm: needs a voxel edge at least that long
    vox = irt.VoxelOctree(64)
    vox.set_xlim(-0.3, 0.3); vox.set_ylim(-0.2, 0.25); vox.set_zlim(-0.1, 0.35)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rng = np.random.default_rng(8)
    sp = np.column_stack([rng.uniform(-0.3, 0.3, (40, 3)), rng.uniform(0.005, 0.05, 40)])
    chk.add_spheres(sp)
    chk.dilate_sphere(0.02)
    chk.remove_interior(False)
    g = orc.Grid(64, vox.limits())
    for row in sp:
        g.add_sphere(row[:3], row[3])
    g.dilate_sphere(0.02)
    g.remove_interior(False)
    assert np.array_equal(chk.engine.get_grid(), np.asarray(g.blocks()).ravel()) and np.count_nonzero(chk.engine.get_grid()) > 200


def test_problem_environment_on_the_device(irt, orc, tmp_path):
    """A problem file's obstacle primitives rasterised straight into a checker's resident grid (Environment::voxelize,
    motion-planning/Environment.cpp:62-100) equal the host mirror's grid; Problem.voxel_backbone_checker wires the checker and
    the motion validator with the file's resolutions."""
    W = irt.workloads
    rng = np.random.default_rng(6)
    env = irt.Environment(points=rng.uniform(-0.2, 0.2, (20, 3)),
                          spheres=[(rng.uniform(-0.2, 0.2, 3), float(rng.uniform(0.005, 0.03))) for _ in range(15)],
                          capsules=[(rng.uniform(-0.2, 0.2, 3), rng.uniform(-0.2, 0.2, 3), float(rng.uniform(0.003, 0.02))) for _ in range(10)])
    vox = irt.VoxelOctree(256)
    vox.set_xlim(-0.25, 0.25); vox.set_ylim(-0.25, 0.25); vox.set_zlim(-0.25, 0.25)
    vfile = tmp_path / "empty.msgpack"
    vox.to_file(str(vfile))
    pr = irt.Problem(robot=W.robot_config2(), env=env, venv=irt.VoxelEnvironment(filename=str(vfile)), start=[1, 2, 3], goal=[3, 2, 1],
                     min_tension_change=0.05)
    pfile = tmp_path / "problem.toml"
    pfile.write_text(pr.to_toml())
    back = irt.Problem.from_toml(str(pfile))
    chk, mv = back.voxel_backbone_checker()
    assert mv.min_tension_change == 0.05 and chk.engine.get_grid().sum() == 0
    back.env.voxelize_into(chk)
    want = back.env.voxelize(vox)
    assert np.array_equal(chk.engine.get_grid(), want.blocks.ravel()) and want.ncells() > 2000
    g = _oracle_grid(orc, vox)
    for p in env.points:
        g.add_sphere(p, 0.0)
    for c, r in env.spheres:
        g.add_sphere(c, r)
    for a, b, r in env.capsules:
        g.add_capsule(a, b, r)
    assert np.array_equal(np.asarray(g.blocks()).ravel(), want.blocks.ravel())
    st = W.random_states(back.robot, 500, seed=3, tau_max=10.0)
    chk2 = irt.VoxelBackboneValidityChecker(back.robot, irt.VoxelEnvironment(), want)
    assert np.array_equal(chk.is_valid(st), chk2.is_valid(st)) and 0 < chk.is_valid(st).mean() < 1
    assert mv.checkMotion(back.start_state(), back.goal_state()) in (True, False)
