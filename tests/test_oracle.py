"""CPU tests of the oracle (oracle/tendon_oracle.c): since the reference ships no tests or golden
vectors, the oracle is pinned by known answers, invariants, an independent
formulation of the right-hand side (6x6 dense solve, the reference's `*_unopt` twin), an
independent high-order integrator, and hand-traced voxel walks.  Also regression-pins the
committed golden fixtures."""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PI = np.pi


def straight_robot(orc, **kw):
    return orc.Robot([[0.0], [2 * PI / 3], [4 * PI / 3]], [[0.01]] * 3, **kw)


def helix_robot(orc, **kw):
    kw.setdefault("dL", 0.2 / 128)
    return orc.Robot([[2 * PI * k / 3, 5.0] for k in range(3)], [[0.01]] * 3, **kw)


def quad_robot(orc, **kw):
    kw.setdefault("dL", 0.2 / 128)
    c1, c2, d1 = [3.0, -2.0, 4.0, -5.0], [10.0, 15.0, -12.0, 8.0], [-0.01, 0.005, 0.0, -0.005]
    return orc.Robot([[PI * k / 2, c1[k], c2[k]] for k in range(4)], [[0.01, d1[k], 0.0] for k in range(4)], **kw)


# ---- independent restatement of the right-hand side in numpy (dense 6x6 solve) ---------------------
def hat(u):
    return np.array([[0, -u[2], u[1]], [u[2], 0, -u[0]], [-u[1], u[0], 0.0]])


def np_rinfo(C, D, t):
    out = []
    for c, d in zip(C, D):
        pc, pd = np.polynomial.Polynomial(c), np.polynomial.Polynomial(d)
        th, th1, th2 = pc(t), pc.deriv(1)(t), pc.deriv(2)(t)
        rh, rh1, rh2 = pd(t), pd.deriv(1)(t), pd.deriv(2)(t)
        e = np.array([np.sin(th), np.cos(th), 0.0])
        e1 = np.array([np.cos(th), -np.sin(th), 0.0]) * th1
        e2 = np.array([-np.sin(th), -np.cos(th), 0.0]) * th1 ** 2 + np.array([np.cos(th), -np.sin(th), 0.0]) * th2
        out.append((rh * e, rh1 * e + rh * e1, rh2 * e + 2 * rh1 * e1 + rh * e2))
    return out


def np_stiffness(ro=0.01, ri=0.0, E=2.1e6, nu=0.3):
    I = PI / 4 * (ro ** 4 - ri ** 4)
    Ar = PI * (ro ** 2 - ri ** 2)
    G = E / (2 * (1 + nu))
    return np.diag([G * Ar, G * Ar, E * Ar]), np.diag([E * I, E * I, 2 * I * G])


def np_deriv(C, D, tau, x, t):
    """tendon_deriv_unopt formulation (tendon/tendon_deriv.cpp:180-260): one dense 6x6 solve."""
    Kse, Kbt = np_stiffness()
    R = x[3:12].reshape(3, 3).T
    v, u = x[12:15], x[15:18]
    A = np.zeros((3, 3)); B = np.zeros((3, 3)); G = np.zeros((3, 3)); H = np.zeros((3, 3))
    a = np.zeros(3); b = np.zeros(3)
    sd = []
    for (r, rd, rdd), ta in zip(np_rinfo(C, D, t), tau):
        pd = np.cross(u, r) + rd + v
        s = np.linalg.norm(pd)
        Ai = -ta * hat(pd) @ hat(pd) / s ** 3
        Bi = hat(r) @ Ai
        A += Ai; B += Bi; G += -Ai @ hat(r); H += -Bi @ hat(r)
        ai = Ai @ (np.cross(u, pd) + np.cross(u, rd) + rdd)
        a += ai; b += np.cross(r, ai)
        sd.append(s)
    e3 = np.array([0, 0, 1.0])
    c = -np.cross(u, Kbt @ u) - np.cross(v, Kse @ (v - e3)) - b
    d = -np.cross(u, Kse @ (v - e3)) - a
    M = np.block([[Kse + A, G], [B, Kbt + H]])
    xi = np.linalg.solve(M, np.concatenate([d, c]))
    out = np.zeros_like(x)
    out[0:3] = R @ v
    out[3:12] = (R @ hat(u)).T.reshape(9)
    out[12:15], out[15:18] = xi[:3], xi[3:]
    out[18] = np.linalg.norm(v)
    out[19:] = sd
    return out


HELIX_C = [[2 * PI * k / 3, 5.0] for k in range(3)]
HELIX_D = [[0.01]] * 3


def test_t_range_grid(orc):
    for dL, P in ((0.005, 41), (0.2 / 128, 129)):
        t = straight_robot(orc, dL=dL).t_range()
        assert len(t) == P and t[0] == 0.0 and t[-1] == 0.2
        d = np.diff(t)
        assert d[0] >= dL / 2 - 1e-15 and d[0] < 1.5 * dL          # util/vector_ops.h:67-75
        assert np.allclose(d[1:], dL, rtol=0, atol=1e-15)            # uniform from the tip end
    t = straight_robot(orc, dL=0.003).t_range()                      # L not a multiple of dL
    assert abs((t[1] - t[0]) - 0.002) < 1e-12 and np.allclose(np.diff(t)[1:], 0.003, atol=1e-15)
    t = orc.Robot([[0.0]], [[0.01]], enable_retraction=True).t_range(0.0731)
    assert abs(t[0] - 0.0731) < 1e-15 and t[-1] == 0.2 and np.allclose(np.diff(t)[1:], 0.005, atol=1e-15)


def test_home_shape_known_answers(orc):
    h = straight_robot(orc).home_shape()
    assert h["converged"] and np.array_equal(h["L_i"], [0.2] * 3) and h["L"] == 0.2
    assert np.array_equal(h["p"][:, :2], np.zeros((41, 2))) and np.array_equal(h["p"][:, 2], h["t"])
    h = helix_robot(orc).home_shape()
    assert np.allclose(h["L_i"], 0.2 * np.sqrt(1 + 0.01 ** 2 * 25.0), rtol=0, atol=1e-16)   # TendonRobot.cpp:287-292
    # general routing: defined Simpson rule vs adaptive quadrature of sqrt(rho'^2 + rho^2 theta'^2 + 1)
    from scipy.integrate import quad
    rb = quad_robot(orc)
    h = rb.home_shape()
    c1, c2, d1 = [3.0, -2.0, 4.0, -5.0], [10.0, 15.0, -12.0, 8.0], [-0.01, 0.005, 0.0, -0.005]
    for k in range(4):
        f = lambda t: np.sqrt(d1[k] ** 2 + (0.01 + d1[k] * t) ** 2 * (c1[k] + 2 * c2[k] * t) ** 2 + 1)
        assert abs(h["L_i"][k] - quad(f, 0, 0.2, epsabs=1e-14)[0]) < 1e-10


def test_zero_tension_is_home(orc):
    for rb in (straight_robot(orc), helix_robot(orc), quad_robot(orc)):
        s = rb.shape(np.zeros(rb.n_tendons))
        h = rb.home_shape()
        assert s["converged"] and s["fp_iters"] == 0
        assert np.abs(s["p"] - h["p"]).max() < 1e-13
        assert np.abs(s["L_i"] - h["L_i"]).max() < 1e-9      # RK4 quadrature of the tendon length vs closed form


def test_rhs_matches_dense_formulation(orc):
    """Block-inverse solve (tendon_deriv.cpp:60-87) == dense 6x6 solve (the `_unopt` twin)."""
    rb = helix_robot(orc)
    rng = np.random.default_rng(0)
    for _ in range(20):
        tau = rng.uniform(0, 20, 3)
        x = np.zeros(22)
        q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
        x[3:12] = q.T.reshape(9)
        x[12:15] = np.array([0, 0, 1.0]) + rng.normal(scale=0.02, size=3)
        x[15:18] = rng.normal(scale=5.0, size=3)
        t = rng.uniform(0, 0.2)
        got, want = rb.deriv(tau, x, t), np_deriv(HELIX_C, HELIX_D, tau, x, t)
        assert np.abs(got - want).max() <= 1e-9 * max(1.0, np.abs(want).max())


def test_rinfo_matches_polynomial_calculus(orc):
    rb = quad_robot(orc)
    c1, c2, d1 = [3.0, -2.0, 4.0, -5.0], [10.0, 15.0, -12.0, 8.0], [-0.01, 0.005, 0.0, -0.005]
    C = [[PI * k / 2, c1[k], c2[k]] for k in range(4)]
    D = [[0.01, d1[k], 0.0] for k in range(4)]
    for t in (0.0, 0.03, 0.2):
        r, rd, rdd = rb.r_info(t)
        for k, (wr, wrd, wrdd) in enumerate(np_rinfo(C, D, t)):
            assert np.abs(r[k] - wr).max() < 1e-15 and np.abs(rd[k] - wrd).max() < 1e-14 and np.abs(rdd[k] - wrdd).max() < 1e-12


def test_initial_bending_residual_invariant(orc):
    """converged <=> base residual <= threshold, and the fixed point solves n(0)=F_t, m(0)=L_t."""
    rb = helix_robot(orc)
    rng = np.random.default_rng(1)
    for _ in range(30):
        tau = rng.uniform(0, 15, 3)
        v0, u0, it = rb.solve_initial_bending(tau)
        res = rb.base_residual(tau, 0.0, v0, u0)
        s = rb.shape(tau)
        assert s["converged"] == (res <= 5e-6)
        assert it < 1000 and res <= 5e-6
        assert np.array_equal(s["v_i"], v0) and np.array_equal(s["u_i"], u0)


def _np_rk4_tip(C, D, tau, v0, u0, tgrid, dL):
    x = np.zeros(22)
    x[3] = x[7] = x[11] = 1
    x[12:15], x[15:18] = v0, u0
    f = lambda xx, tt: np_deriv(C, D, tau, xx, tt)
    for j in range(len(tgrid) - 1):
        cur = tgrid[j]
        while tgrid[j + 1] - cur > np.finfo(float).eps:
            h = min(dL, tgrid[j + 1] - cur)
            k1 = f(x, cur); k2 = f(x + h / 2 * k1, cur + h / 2); k3 = f(x + h / 2 * k2, cur + h / 2); k4 = f(x + h * k3, cur + h)
            x = x + h / 6 * k1 + h / 3 * k2 + h / 3 * k3 + h / 6 * k4
            cur += h
    return x


def test_rk4_stepping_against_independent_rk4(orc):
    rb = helix_robot(orc, dL=0.005)
    tau = np.array([7.0, 2.0, 11.0])
    s = rb.shape(tau)
    x = _np_rk4_tip(HELIX_C, HELIX_D, tau, s["v_i"], s["u_i"], rb.t_range(), 0.005)
    assert np.abs(x[0:3] - s["p"][-1]).max() < 1e-12
    assert abs(x[18] - s["L"]) < 1e-13 and np.abs(x[19:] - s["L_i"]).max() < 1e-13


def test_against_high_order_integrator(orc):
    """DOP853 (rtol 1e-12) on the independent RHS: RK4 converges to it with 4th order."""
    from scipy.integrate import solve_ivp
    tau = np.array([9.0, 1.5, 4.0])
    errs = []
    for dL in (0.01, 0.005, 0.0025):
        rb = helix_robot(orc, dL=dL)
        s = rb.shape(tau)
        x0 = np.zeros(22)
        x0[3] = x0[7] = x0[11] = 1
        x0[12:15], x0[15:18] = s["v_i"], s["u_i"]
        ref = solve_ivp(lambda t, x: np_deriv(HELIX_C, HELIX_D, tau, x, t), (0, 0.2), x0, method="DOP853",
                        rtol=1e-12, atol=1e-14)
        errs.append(np.abs(ref.y[0:3, -1] - s["p"][-1]).max())
    assert errs[-1] < 2e-9
    assert 10 < errs[0] / errs[1] < 24 and 10 < errs[1] / errs[2] < 24       # ~16 = 2^4


def test_rotate_z(orc):
    rb = orc.Robot([[0.0], [2 * PI / 3], [4 * PI / 3]], [[0.01]] * 3, enable_rotation=True)
    tau = [5.0, 1.0, 2.0]
    base = straight_robot(orc).shape(tau)
    th = 0.7
    rot = rb.shape(tau + [th])
    c, s = np.cos(th), np.sin(th)
    Rz = np.array([[c, -s, 0], [s, c, 0], [0, 0, (1 - c) + c]])
    assert np.abs(rot["p"] - base["p"] @ Rz.T).max() < 1e-16
    Rm = base["R"].reshape(-1, 3, 3).transpose(0, 2, 1)
    assert np.abs(rot["R"].reshape(-1, 3, 3).transpose(0, 2, 1) - Rz @ Rm).max() < 1e-15
    assert np.array_equal(rot["L_i"], base["L_i"])


def test_retraction_single_point_and_two_step_interval(orc):
    rb = orc.Robot([[0.0]], [[0.01]], enable_retraction=True)
    s = rb.shape([3.0, 0.2])                     # s_start == L -> single point (TendonRobot.cpp:361-372)
    assert len(s["p"]) == 1 and s["L"] == 0 and s["converged"]
    s = rb.shape([3.0, 0.25])                    # truncated to L
    assert len(s["p"]) == 1
    s = rb.shape([3.0, 0.0731])
    assert abs(s["L"] - (0.2 - 0.0731) * np.linalg.norm(s["v_i"])) < 1e-3 and len(s["p"]) == len(rb.t_range(0.0731))


# ---- voxels ------------------------------------------------------------------------------------------
def test_bitmask_and_cells(orc):
    lib = orc._load()
    assert lib.orc_bitmask(0, 0, 0) == 1 and lib.orc_bitmask(0, 0, 1) == 2 and lib.orc_bitmask(0, 1, 0) == 16
    assert lib.orc_bitmask(1, 0, 0) == 1 << 16 and lib.orc_bitmask(3, 3, 3) == 1 << 63
    g = orc.Grid(8, (0, 1, 0, 1, 0, 1))
    assert not g.set_cell(5, 2, 7) and g.set_cell(5, 2, 7) and g.cell(5, 2, 7) and not g.cell(5, 2, 6)
    b = g.blocks()
    assert b[1, 0, 1] == np.uint64(1) << np.uint64(1 * 16 + 2 * 4 + 3) and np.count_nonzero(b) == 1
    with pytest.raises(ValueError):
        orc.Grid(12)
    with pytest.raises(ValueError):
        orc.Grid(8, (0, 0, 0, 1, 0, 1))


def test_domain_and_cell_lookup(orc):
    g = orc.Grid(16, (-0.8, 0.8, -0.8, 0.8, -0.8, 0.8))
    assert g.is_in_domain(-0.8, 0.8, 0.0) and not g.is_in_domain(0.8000001, 0, 0)     # closed on both ends
    assert g.nearest_cell(-5, 0.05, 5) == (0, 8, 15)
    assert g.find_cell(-0.8, 0.0, 0.79) == (0, 8, 15)
    assert g.find_cell(0.8, 0, 0)[0] == 16                                             # no clamp, as the reference
    with pytest.raises(ValueError):
        g.find_cell(0.81, 0, 0)
    assert not g.collides((0.05, 0.05, 0.05))
    g.add_point((0.05, 0.05, 0.05))
    assert g.collides((0.09, 0.01, 0.099)) and not g.collides((0.11, 0.05, 0.05)) and not g.collides((5, 5, 5))


def test_add_line_hand_traced(orc):
    """Voxel size 0.1.  Expected sequences follow VoxelOctree::add_line statement by statement."""
    g = orc.Grid(16, (0, 1.6, 0, 1.6, 0, 1.6))
    # +x ray through cells 0..4, plus ONE cell beyond B's cell (the walk continues while i == B)
    g.add_line([0.05, 0.05, 0.05], [0.45, 0.05, 0.05])
    assert g.cells() == [(i, 0, 0) for i in range(6)]
    # zero-length segment: its cell and one neighbour in +z (all t equal -> "else" branch = z)
    g.clear(); g.add_line([0.55, 0.55, 0.55], [0.55, 0.55, 0.55])
    assert g.cells() == [(5, 5, 5), (5, 5, 6)]
    # segment entirely outside
    g.clear(); g.add_line([-0.5, -0.5, -0.5], [-0.1, -0.6, -0.2])
    assert g.cells() == []
    # entering from outside: cells are set only once inside; again one cell past B
    g.clear(); g.add_line([-0.25, 0.45, 0.45], [0.25, 0.45, 0.45])
    assert g.cells() == [(0, 4, 4), (1, 4, 4), (2, 4, 4), (3, 4, 4)]
    # leaving the domain stops the walk
    g.clear(); g.add_line([1.45, 1.45, 1.45], [1.85, 1.55, 1.45])
    assert g.cells() == [(14, 14, 14), (15, 14, 14)]
    # quirk 1: the initial boundary distances mix voxel and metre units (VoxelOctree.cpp:371-373), so
    # a long oblique segment is NOT the Amanatides-Woo set: z is exhausted before y ever steps.
    g.clear(); g.add_line([0.85, 0.95, 0.25], [0.55, 0.75, 0.15])
    assert g.cells() == [(5, 7, 1), (5, 9, 0), (5, 9, 1), (6, 9, 1), (7, 9, 1), (8, 9, 1), (8, 9, 2)]


def test_add_line_cells_golden(orc):
    d = np.load(os.path.join(GOLD, "add_line_cells.npz"))
    g = orc.Grid(16, (0, 1.6, 0, 1.6, 0, 1.6))
    for name in [k[:-6] for k in d.files if k.endswith("_cells")]:
        g.clear()
        g.add_line(d[name + "_a"], d[name + "_b"])
        assert np.array_equal(np.array(g.cells(), dtype=np.int32).reshape(-1, 3), d[name + "_cells"]), name


def test_short_segments_cover_both_end_cells(orc):
    """With dL <= voxel (the checked precondition) every segment sets A's and B's cells."""
    g = orc.Grid(64, (-0.25, 0.25) * 3)
    rng = np.random.default_rng(3)
    for _ in range(200):
        a = rng.uniform(-0.2, 0.2, 3)
        b = a + rng.normal(size=3) * 0.002
        g.clear(); g.add_line(a, b)
        assert g.cell(*g.find_cell(*a)) and g.cell(*g.find_cell(*b))
        assert 2 <= g.ncells() <= 6


def test_segment_aabox(orc):
    lib = orc._load()
    f = lambda a, b: bool(lib.orc_segment_aabox_intersect(*[orc._dp(orc._f64(v)) for v in (a, b, [0, 0, 0], [1, 1, 1])]))
    assert f([0.5, 0.5, 0.5], [0.6, 0.5, 0.5]) and f([-1, 0.5, 0.5], [2, 0.5, 0.5]) and f([-0.5, -0.5, 0.5], [0.6, 0.6, 0.5])
    assert not f([-1, -1, -1], [-0.1, -0.2, -0.3]) and not f([1.5, -1, 0.5], [3, 0.9, 0.5])
    assert f([0.5, 0.5, 0.5], [0.5, 0.5, 0.5])     # degenerate: NaN direction -> all comparisons false -> intersects


def test_add_sphere_voxel_centres(orc):
    g = orc.Grid(32, (0, 1, 0, 1, 0, 1))
    c, r = np.array([0.52, 0.47, 0.5]), 0.11
    g.add_sphere(c, r)
    idx = (np.arange(32) + 0.5) / 32
    X, Y, Z = np.meshgrid(idx, idx, idx, indexing="ij")
    want = ((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) <= r * r
    want[g.nearest_cell(*c)] = True
    got = np.zeros((32, 32, 32), bool)
    for cell in g.cells():
        got[cell] = True
    assert np.array_equal(got, want)


# ---- self collision --------------------------------------------------------------------------------------
def _brute_seg_dist(A, B, C, D, n=400):
    s = np.linspace(0, 1, n)
    P = A[None] + (B - A)[None] * s[:, None]
    Q = C[None] + (D - C)[None] * s[:, None]
    return np.sqrt(((P[:, None, :] - Q[None, :, :]) ** 2).sum(-1)).min()


def test_closest_st_segment(orc):
    """Follows collision_primitives.cpp:10-102 INCLUDING its end-point branches, which are not the
    textbook ones: for t < 0 it returns s = bound(-c/a) with c = |CD|^2 (always 0), for t > 1
    s = bound((b - c)/a).  The restatement must reproduce that, not "fix" it."""
    lib = orc._load()
    rng = np.random.default_rng(4)
    import ctypes as Ct
    n_interior = n_lo = n_hi = 0
    for k in range(200):
        A, B, C, D = rng.normal(size=(4, 3))
        s, t = Ct.c_double(), Ct.c_double()
        lib.orc_closest_st_segment(orc._dp(A), orc._dp(B), orc._dp(C), orc._dp(D), Ct.byref(s), Ct.byref(t))
        assert 0 <= s.value <= 1 and 0 <= t.value <= 1
        AB, CD, AC = B - A, D - C, C - A
        a, c, b, d, e = AB @ AB, CD @ CD, AB @ CD, AC @ AB, AC @ CD
        den = a * c - b * b
        tt = (b * d - a * e) / den
        if 0 <= tt <= 1:
            n_interior += 1
            ss = min(1.0, max(0.0, (c * d - b * e) / den))
            assert abs(s.value - ss) < 1e-12 and abs(t.value - tt) < 1e-12
            if 0 < ss < 1:                                   # true interior optimum
                dist = np.linalg.norm((A + AB * s.value) - (C + CD * t.value))
                assert abs(dist - _brute_seg_dist(A, B, C, D, 600)) < 5e-3
        elif tt < 0:
            n_lo += 1
            assert t.value == 0.0 and s.value == 0.0
        else:
            n_hi += 1
            assert t.value == 1.0 and abs(s.value - min(1.0, max(0.0, (b - c) / a))) < 1e-14
    assert n_interior > 20 and n_lo > 20 and n_hi > 20
    # degenerate and parallel branches
    A, B = np.array([0.0, 0, 0]), np.array([1.0, 0, 0])
    for C, D, want in (((0.2, 1, 0), (0.7, 1, 0), (0.0, 0.0)),      # parallel, overlapping: closest_CD_t(A) < 0 -> B? -> ...
                       ((2.0, 1, 0), (3.0, 1, 0), (1.0, 0.0)),      # parallel, disjoint: best end-point pair
                       ((0.5, 1, 0), (0.5, 1, 0), (0.5, 0.0))):     # C == D
        s, t = Ct.c_double(), Ct.c_double()
        lib.orc_closest_st_segment(orc._dp(A), orc._dp(B), orc._dp(np.array(C)), orc._dp(np.array(D)), Ct.byref(s), Ct.byref(t))
        if C == (0.2, 1, 0):
            assert (s.value, t.value) == (1.0, 0.0) or 0 <= s.value <= 1   # first feasible candidate in the cascade
        else:
            assert (s.value, t.value) == want


def test_collides_self_cases(orc):
    rb = straight_robot(orc)                     # r = 0.015
    t = np.linspace(0, 0.2, 41)
    rod = np.stack([0 * t, 0 * t, t], 1)
    assert not rb.collides_self(rod)
    assert not rb.collides_self(rod[:2])         # N <= 2
    # a loop of circumference 0.16: points meet again -> collision
    ang = np.linspace(0, 2 * PI * 0.97, 41)
    R = 0.16 / (2 * PI)
    loop = np.stack([R * np.cos(ang), R * np.sin(ang), 0 * ang], 1)
    assert rb.collides_self(loop)
    # the 3r arc gate: a hairpin whose legs are 0.02 apart (< 2r) but only 0.03 of arc away is ignored
    hair = np.array([[0, 0, 0], [0.01, 0, 0], [0.02, 0, 0], [0.02, 0.01, 0], [0.01, 0.01, 0], [0, 0.01, 0.0]])
    assert not rb.collides_self(hair)
    wide = orc.Robot([[0.0]], [[0.01]], r=0.004)  # same shape, thin robot: 3r = 0.012 <= arc, 2r = 0.008 < 0.01
    assert not wide.collides_self(hair)
    thick = orc.Robot([[0.0]], [[0.01]], r=0.0055)  # 3r = 0.0165, 2r = 0.011 >= 0.01 and far capsules pass the gate
    assert thick.collides_self(hair)


def test_length_limits(orc):
    rb = straight_robot(orc, min_length=-0.125, max_length=0.25)      # binary fractions: exact arithmetic
    home = [1.0, 1.0, 1.0]
    assert rb.is_within_length_limits(home, [1.0, 1.0, 1.0])
    assert rb.is_within_length_limits(home, [0.75, 1.125, 1.0])                          # inclusive bounds
    assert not rb.is_within_length_limits(home, [0.75 - 2 ** -20, 1.0, 1.0])
    assert not rb.is_within_length_limits(home, [1.0, 1.125 + 2 ** -20, 1.0])


# ---- golden regression -----------------------------------------------------------------------------------
def test_golden_config1(orc):
    d = np.load(os.path.join(GOLD, "config1_fk.npz"))
    fk = straight_robot(orc).fk_batch(d["states"])
    assert np.array_equal(fk["p"][:, -1], d["tips"]) and np.array_equal(fk["L_i"], d["L_i"])
    assert np.array_equal(fk["converged"], d["converged"]) and np.array_equal(fk["p"][:8], d["p_first8"])
    assert d["converged"].all()


def test_golden_config2_validity(orc, irt, helpers):
    d = np.load(os.path.join(GOLD, "config2_validity.npz"))
    vox = irt.VoxelOctree.from_sparse(256, (-0.25, 0.25) * 3, d["grid_ids"], d["grid_masks"])
    vox2, centres = irt.workloads.reach_environment(seed=7, n_spheres=64)
    assert np.array_equal(centres, d["sphere_centres"]) and vox == vox2
    og = helpers.oracle_grid(orc, vox)
    n = 1024
    # the OpenMP build (-O3 -march=native: FMA contraction) agrees in every verdict and to rounding in the tips; the
    # strict build (-ffp-contract=off) reproduces the fixture bit for bit
    valid, tips, _ = orc.validate_batch(helix_robot(orc), og, d["states"][:n], nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(valid, d["valid"][:n]) and np.abs(tips - d["tips"][:n]).max() <= 1e-13
    valid, tips, _ = orc.validate_batch(helix_robot(orc), og, d["states"][:96])
    assert np.array_equal(valid, d["valid"][:96]) and np.array_equal(tips, d["tips"][:96])
    fl = np.array([orc.is_valid_state(helix_robot(orc), og, s)[2] for s in d["states"][:64]], dtype=np.uint8)
    assert np.array_equal(fl, d["flags512"][:64])
    assert 0.3 < d["valid"].mean() < 0.9


def test_golden_config3(orc):
    d = np.load(os.path.join(GOLD, "config3_fk.npz"))
    rb = quad_robot(orc)
    fk = rb.fk_batch(d["states"][:64])
    assert np.array_equal(fk["p"][:, -1], d["tips"][:64]) and np.array_equal(fk["L_i"], d["L_i"][:64])
    assert np.array_equal(rb.home_shape()["L_i"], d["home_L_i"])


def test_edge_oracles_invariants(orc):
    """The two edge checks (adaptive bisection, VoxelEnvironment.cpp:207-444; discrete sampling,
    VoxelBackboneDiscreteMotionValidator.cpp:9-79) against facts that hold by construction."""
    rb = straight_robot(orc, dL=0.2 / 64)
    g = orc.Grid(128, (-0.25, 0.25) * 3)
    sp = orc.space_params(min_tension_change=0.5)
    a, b = np.array([0.5, 0.0, 0.2]), np.array([5.8, 0.9, 0.2])
    nd = int(np.ceil(np.linalg.norm(a - b) / 0.5))                   # 10.75 -> 11
    # free space: both valid; the discrete check takes a, nd - 1 interior samples and b
    d = orc.check_motion_discrete(rb, g, a, b, sp)
    m = orc.check_motion(rb, g, a, b, sp)
    assert d["valid"] and m["valid"] and d["n_fk"] == nd + 1 and d["last_valid_t"] == 1.0 and m["last_valid_t"] == 1.0
    assert 3 <= m["n_fk"] <= 2 * nd + 1
    # a == b: two samples either way
    assert orc.check_motion_discrete(rb, g, a, a, sp)["n_fk"] == 2 and orc.check_motion(rb, g, a, a, sp)["n_fk"] == 2
    # an obstacle on the tip of an interior sample: both invalid; the discrete loop stops right there
    k = nd // 2
    mid = a + (b - a) * (k / nd)
    _, tip, _ = orc.is_valid_state(rb, g, mid)
    g.add_sphere(tip, 0.006)
    assert not orc.is_valid_state(rb, g, mid)[0] and orc.is_valid_state(rb, g, a)[0] and orc.is_valid_state(rb, g, b)[0]
    d0 = orc.check_motion_discrete(rb, g, a, b, sp)
    d1 = orc.check_motion_discrete(rb, g, a, b, sp, until_invalid=True)
    assert not d0["valid"] and d0["is_fully_valid"] and d0["n_fk"] == nd + 1      # shapes are fine, the union collides
    assert not d1["is_fully_valid"] and d1["n_fk"] <= k + 1 and d1["last_valid_t"] == (d1["n_fk"] - 2) / nd
    m0 = orc.check_motion(rb, g, a, b, sp)
    m1 = orc.check_motion_until_invalid(rb, g, a, b, sp)
    assert not m0["valid"] and not m1["is_fully_valid"] and 0.0 <= m1["last_valid_t"] < k / nd + 1e-12


def _cells(g):
    """bool[N,N,N] view of an oracle grid (block ((bx*Nb)+by)*Nb+bz, bit x*16+y*4+z)."""
    Nb = g.N // 4
    blk = np.asarray(g.blocks()).reshape(Nb, Nb, Nb)
    out = np.zeros((g.N,) * 3, dtype=bool)
    for x in range(4):
        for y in range(4):
            for z in range(4):
                out[x::4, y::4, z::4] = (blk >> np.uint64(x * 16 + y * 4 + z)) & np.uint64(1)
    return out


def _np_dilate(c, n, moves):
    for _ in range(n):
        p = np.pad(c, 1)
        out = c.copy()
        for dx, dy, dz in moves:
            out |= p[1 - dx:1 - dx + c.shape[0], 1 - dy:1 - dy + c.shape[1], 1 - dz:1 - dz + c.shape[2]]
        c = out
    return c


def test_environment_edits_against_numpy_morphology(orc):
    """remove_interior / dilate (VoxelOctree.cpp:533-818) against a plain numpy statement of what they
    compute: erosion-style shelling with the outside counted as occupied, and n-step dilation by the
    reference's move lists (the 27-neighbour list lacks (-1,+1,+1))."""
    rng = np.random.default_rng(5)
    g = orc.Grid(32, (0, 1, 0, 1, 0, 1))
    for c in rng.uniform(0.1, 0.9, (6, 3)):
        g.add_sphere(c, rng.uniform(0.05, 0.2))
    g.set_cell(0, 0, 0); g.set_cell(31, 31, 31); g.set_cell(0, 17, 31)
    base = _cells(g)
    m6 = [(-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
    m27 = [(i, j, k) for i in (-1, 0, 1) for j in (-1, 0, 1) for k in (-1, 0, 1) if (i, j, k) != (-1, 1, 1)]
    for num, diag, moves in ((1, False, m6), (3, False, m6), (6, False, m6), (1, True, m27), (5, True, m27)):
        h = orc.Grid(32, (0, 1, 0, 1, 0, 1)); h.blocks()[...] = g.blocks()
        h.dilate(num, diag)
        assert np.array_equal(_cells(h), _np_dilate(base, num, moves)), (num, diag)
    for diag in (True, False):
        h = orc.Grid(32, (0, 1, 0, 1, 0, 1)); h.blocks()[...] = g.blocks()
        h.remove_interior(diag)
        p = np.pad(base, 1, constant_values=True)
        full = np.ones_like(base)
        for i in (-1, 0, 1):
            for j in (-1, 0, 1):
                for k in (-1, 0, 1):
                    if diag or abs(i) + abs(j) + abs(k) <= 1:
                        full &= p[1 + i:33 + i, 1 + j:33 + j, 1 + k:33 + k]
        assert np.array_equal(_cells(h), base & ~full), diag
    h = orc.Grid(32, (0, 1, 0, 1, 0, 1)); h.blocks()[...] = g.blocks()
    h.dilate_sphere(0.07)                                   # round(0.07 / (1/32)) = 2 six-neighbour steps
    assert np.array_equal(_cells(h), _np_dilate(base, 2, m6))


def test_add_capsule_voxel_centres(orc):
    """orc_grid_add_capsule (VoxelOctree.cpp:471-515) against a brute-force mask: voxel centres whose distance to the
    segment is <= r (collides(Capsule, Point), collision.hxx:83-87), plus the two end points' cells."""
    rng = np.random.default_rng(12)
    for trial in range(6):
        g = orc.Grid(32, (-1, 1, -0.5, 0.5, 0, 2))
        a, b = rng.uniform([-1, -0.5, 0], [1, 0.5, 2]), rng.uniform([-1.2, -0.6, -0.2], [1.2, 0.6, 2.2])
        r = rng.uniform(0.03, 0.3)
        if trial == 0:
            b = a.copy()                                     # degenerate capsule = sphere
        g.add_capsule(a, b, r)
        dx, dy, dz = g.cell_size
        X = (-1 + dx * (np.arange(32) + 0.5))[:, None, None]
        Y = (-0.5 + dy * (np.arange(32) + 0.5))[None, :, None]
        Z = (0 + dz * (np.arange(32) + 0.5))[None, None, :]
        d = b - a
        dsq = d @ d
        t = np.zeros((32, 32, 32)) if dsq <= 1e-30 else np.clip((d[0] * (X - a[0]) + d[1] * (Y - a[1]) + d[2] * (Z - a[2])) / dsq, 0, 1)
        dist2 = (a[0] + d[0] * t - X) ** 2 + (a[1] + d[1] * t - Y) ** 2 + (a[2] + d[2] * t - Z) ** 2
        want = dist2 <= r * r
        for p in (a, b):
            if g.is_in_domain(*p):
                want[tuple(g.nearest_cell(*p))] = True
        got = np.zeros((32, 32, 32), bool)
        for c in g.cells():
            got[tuple(c)] = True
        near = np.abs(dist2 - r * r) < 1e-12                 # centres on the surface to rounding: either answer
        assert np.array_equal(got | near, want | near) and want.sum() > 20
        if trial == 0:
            s = orc.Grid(32, (-1, 1, -0.5, 0.5, 0, 2))
            s.add_sphere(a, r)
            assert np.array_equal(s.blocks(), g.blocks())


def test_lazy_prm_query_loop_hand_traced(orc):
    """orc_roadmap_query on a 6-vertex graph where the lazy loop can be followed by hand (constructSolution,
    VoxelCachedLazyPRM.cpp:2689-2771): the cheapest path runs through an invalid vertex (ALL invalid interior vertices
    of the candidate path are removed, :2711-2733), the next one over an invalid edge (only the FIRST invalid edge from
    the goal side is removed, :2745-2757), the third is clean."""
    rb = orc.Robot([[0.0], [2.0]], [[0.01], [0.01]])                       # 2 tendons: states are points of the plane
    st = np.array([[0, 0], [1, 0.2], [2, 0], [1, -1], [1, 1.5], [0.5, 3.0]], float)
    edges = np.array([[0, 1], [1, 2], [0, 3], [3, 2], [0, 4], [4, 2], [4, 5]])
    g = orc.Grid(16, (-1, 1) * 3)
    g.set_cell(3, 3, 3)                                                     # block 0, bit of cell (3, 3, 3)
    bm = lambda x, y, z: np.uint64(int(orc._load().orc_bitmask(x, y, z)))
    hitmask, free = bm(3, 3, 3), bm(0, 0, 0)
    # vertex 1 collides; edge 3-2 (index 3) collides; everything else is free
    vc = dict(offsets=np.arange(7), block_ids=np.zeros(6, np.uint32), masks=np.array([free, hitmask, free, free, free, free]))
    ec = dict(offsets=np.arange(8), block_ids=np.zeros(7, np.uint32), masks=np.array([free, free, free, hitmask, free, free, free]))
    rm = orc.Roadmap(rb, st, edges, None, vc, ec)
    q = rm.query(g, 0, 2)
    assert q["n"] == 3 and list(q["path"]) == [0, 4, 2] and q["iterations"] == 3
    assert abs(q["cost"] - 2 * np.hypot(1, 1.5)) < 1e-15
    vs, es = rm.validity()
    # 0-1, 1-2 never looked at (vertex 1 went first); 0-3 neither: the edges are walked from the goal and 3-2 failed first
    assert list(vs) == [1, 2, 1, 1, 1, 0] and list(es) == [0, 0, 0, 2, 1, 1, 0]
    assert q["checked"] == 2 + 1 + 1 + 1 + 1 + 2      # start, goal | v1 | v3 | e(3,2) | v4 | e(4,2), e(0,4)
    # known validity is kept: the second query searches once and checks nothing on that path
    q2 = rm.query(g, 2, 0)
    assert list(q2["path"]) == [2, 4, 0] and q2["iterations"] == 1 and q2["checked"] == 0
    assert rm.query(g, 0, 0)["n"] == 1 and rm.query(g, 1, 2)["n"] == -2 and rm.query(g, 2, 1)["n"] == -3
    # an empty grid after clearValidity: the direct route again
    rm.clear_validity()
    g.clear()
    q3 = rm.query(g, 0, 2)
    assert list(q3["path"]) == [0, 1, 2] and q3["iterations"] == 1
    # disconnected: vertex 5 hangs on one edge, and that edge collides
    ec2 = dict(ec, masks=np.array([free, free, free, free, free, free, hitmask]))
    rm2 = orc.Roadmap(rb, st, edges, None, vc, ec2)
    g.set_cell(3, 3, 3)
    assert rm2.query(g, 0, 5)["n"] == 0
