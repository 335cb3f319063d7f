"""ctypes front end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
PARITY UNPINNED (no reference tests/fixtures exist and the reference cannot be built here):
see oracle/tendon_oracle.h and DESIGN.md.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
MAX_TENDONS = 8
MAX_COEF = 8

c_double_p = C.POINTER(C.c_double)


class OrcRobot(C.Structure):
    _fields_ = [
        ("r", C.c_double),
        ("L", C.c_double), ("dL", C.c_double), ("ro", C.c_double), ("ri", C.c_double),
        ("E", C.c_double), ("nu", C.c_double),
        ("n_tendons", C.c_int), ("n_a", C.c_int), ("n_m", C.c_int),
        ("C", (C.c_double * MAX_COEF) * MAX_TENDONS),
        ("D", (C.c_double * MAX_COEF) * MAX_TENDONS),
        ("max_tension", C.c_double * MAX_TENDONS),
        ("min_length", C.c_double * MAX_TENDONS),
        ("max_length", C.c_double * MAX_TENDONS),
        ("enable_rotation", C.c_int), ("enable_retraction", C.c_int),
        ("residual_threshold", C.c_double),
    ]


class OrcResult(C.Structure):
    _fields_ = [
        ("cap", C.c_int), ("n", C.c_int),
        ("t", c_double_p), ("p", c_double_p), ("R", c_double_p),
        ("L", C.c_double), ("L_i", C.c_double * MAX_TENDONS),
        ("u_i", C.c_double * 3), ("u_f", C.c_double * 3),
        ("v_i", C.c_double * 3), ("v_f", C.c_double * 3),
        ("converged", C.c_int), ("fp_iters", C.c_int),
    ]


class OrcGrid(C.Structure):
    _fields_ = [
        ("N", C.c_int), ("Nb", C.c_int),
        ("xmin", C.c_double), ("xmax", C.c_double), ("ymin", C.c_double), ("ymax", C.c_double),
        ("zmin", C.c_double), ("zmax", C.c_double),
        ("dx", C.c_double), ("dy", C.c_double), ("dz", C.c_double),
        ("blocks", C.POINTER(C.c_uint64)),
    ]


class OrcSpaceParams(C.Structure):
    _fields_ = [("min_tension_change", C.c_double), ("min_rotation_change", C.c_double),
                ("min_retraction_change", C.c_double)]


def _cpu_stamp():
    """Identity of the host CPU the -march=native build is valid for (model name + ISA flags)."""
    model, flags = "", ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and not model:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("flags") and not flags:
                flags = " ".join(sorted(line.split(":", 1)[1].split()))
            if model and flags:
                break
    except OSError:
        pass
    import hashlib
    return model + " " + hashlib.sha1(flags.encode()).hexdigest()[:16]


def build(force=False):
    """Compile oracle/_build/liboracle{,_omp}.so with gcc (seconds).  The OpenMP library is built with
    -march=native, so it is rebuilt when this host's CPU is not the one it was built on
    (the prebuilt file travels from the dev container to the GPU box)."""
    out = os.path.join(_HERE, "_build", "liboracle.so")
    omp = os.path.join(_HERE, "_build", "liboracle_omp.so")
    stamp = os.path.join(_HERE, "_build", "omp.cpu")
    src = os.path.join(_HERE, "tendon_oracle.c")
    newest = max(os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "Makefile")))
    here = _cpu_stamp()
    same_cpu = os.path.exists(stamp) and open(stamp).read() == here
    if force or not os.path.exists(out) or os.path.getmtime(out) < newest:
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/liboracle.so"])
    if force or not same_cpu or not os.path.exists(omp) or os.path.getmtime(omp) < newest:
        if os.path.exists(omp):
            os.remove(omp)
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/liboracle_omp.so"])
        with open(stamp, "w") as f:
            f.write(here)
    return out


_libs = {}


def _load(kind="strict"):
    if kind in _libs:
        return _libs[kind]
    name = {"strict": "liboracle.so", "omp": "liboracle_omp.so"}[kind]
    path = os.path.join(_HERE, "_build", name)
    build()                      # no-op when up to date; rebuilds the -march=native library on a different host CPU
    lib = C.CDLL(path)
    P = C.POINTER
    lib.orc_state_size.argtypes = [P(OrcRobot)]
    lib.orc_t_range.argtypes = [C.c_double, C.c_double, C.c_double, c_double_p, C.c_int]
    lib.orc_get_r_info.argtypes = [P(OrcRobot), C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_solve_initial_bending.argtypes = [P(OrcRobot), c_double_p, C.c_double, c_double_p, c_double_p]
    lib.orc_tendon_deriv.argtypes = [P(OrcRobot), c_double_p, c_double_p, c_double_p, C.c_double]
    lib.orc_tension_shape.argtypes = [P(OrcRobot), c_double_p, C.c_double, P(OrcResult)]
    lib.orc_home_shape.argtypes = [P(OrcRobot), C.c_double, P(OrcResult)]
    lib.orc_shape.argtypes = [P(OrcRobot), c_double_p, P(OrcResult)]
    lib.orc_home_shape_state.argtypes = [P(OrcRobot), c_double_p, P(OrcResult)]
    lib.orc_base_residual.argtypes = [P(OrcRobot), c_double_p, C.c_double, c_double_p, c_double_p]
    lib.orc_base_residual.restype = C.c_double
    lib.orc_collides_self.argtypes = [c_double_p, C.c_int, C.c_double]
    lib.orc_is_within_length_limits.argtypes = [P(OrcRobot), c_double_p, c_double_p]
    lib.orc_is_valid_shape.argtypes = [P(OrcRobot), P(OrcResult), P(OrcResult)]
    lib.orc_closest_st_segment.argtypes = [c_double_p] * 4 + [c_double_p, c_double_p]
    lib.orc_grid_create.argtypes = [C.c_int]
    lib.orc_grid_create.restype = P(OrcGrid)
    lib.orc_grid_empty_copy.argtypes = [P(OrcGrid)]
    lib.orc_grid_empty_copy.restype = P(OrcGrid)
    lib.orc_grid_free.argtypes = [P(OrcGrid)]
    lib.orc_grid_clear.argtypes = [P(OrcGrid)]
    lib.orc_grid_set_limits.argtypes = [P(OrcGrid)] + [C.c_double] * 6
    lib.orc_bitmask.argtypes = [C.c_int] * 3
    lib.orc_bitmask.restype = C.c_uint64
    lib.orc_grid_set_cell.argtypes = [P(OrcGrid)] + [C.c_int] * 3
    lib.orc_grid_cell.argtypes = [P(OrcGrid)] + [C.c_int] * 3
    lib.orc_grid_is_in_domain.argtypes = [P(OrcGrid)] + [C.c_double] * 3
    lib.orc_grid_nearest_cell.argtypes = [P(OrcGrid)] + [C.c_double] * 3 + [P(C.c_int)]
    lib.orc_grid_find_cell.argtypes = [P(OrcGrid)] + [C.c_double] * 3 + [P(C.c_int)]
    lib.orc_grid_add_point.argtypes = [P(OrcGrid)] + [C.c_double] * 3
    lib.orc_grid_add_line.argtypes = [P(OrcGrid), c_double_p, c_double_p]
    lib.orc_grid_add_piecewise_line.argtypes = [P(OrcGrid), c_double_p, C.c_int]
    lib.orc_grid_add_sphere.argtypes = [P(OrcGrid), c_double_p, C.c_double]
    lib.orc_grid_add_capsule.argtypes = [P(OrcGrid), c_double_p, c_double_p, C.c_double]
    lib.orc_grid_add_capsule.restype = None
    lib.orc_capsule_contains.argtypes = [c_double_p, c_double_p, C.c_double, c_double_p]
    lib.orc_grid_remove_interior.argtypes = [P(OrcGrid), C.c_int]
    lib.orc_grid_remove_interior.restype = None
    lib.orc_grid_dilate.argtypes = [P(OrcGrid), C.c_int, C.c_int]
    lib.orc_grid_dilate.restype = None
    lib.orc_grid_dilate_sphere.argtypes = [P(OrcGrid), C.c_double]
    lib.orc_grid_dilate_sphere.restype = None
    lib.orc_grid_collides.argtypes = [P(OrcGrid), P(OrcGrid)]
    lib.orc_grid_collides_point.argtypes = [P(OrcGrid)] + [C.c_double] * 3
    lib.orc_grid_nblocks.argtypes = [P(OrcGrid)]
    lib.orc_grid_nblocks.restype = C.c_size_t
    lib.orc_grid_ncells.argtypes = [P(OrcGrid)]
    lib.orc_grid_ncells.restype = C.c_size_t
    lib.orc_segment_aabox_intersect.argtypes = [c_double_p] * 4
    lib.orc_rotate_points.argtypes = [c_double_p, c_double_p, C.c_int]
    lib.orc_is_valid_state.argtypes = [P(OrcRobot), P(OrcGrid), c_double_p, c_double_p, c_double_p, P(C.c_int)]
    lib.orc_is_valid_state_spheres.argtypes = [P(OrcRobot), P(OrcGrid), c_double_p, c_double_p, c_double_p, P(C.c_int)]
    lib.orc_validate_batch.argtypes = [P(OrcRobot), P(OrcGrid), c_double_p, c_double_p, C.c_long,
                                       P(C.c_uint8), c_double_p, C.c_int]
    lib.orc_fk_batch.argtypes = [P(OrcRobot), c_double_p, C.c_long, c_double_p, c_double_p, c_double_p,
                                 P(C.c_uint8), C.c_int, C.c_int]
    lib.orc_valid_segment_count.argtypes = [P(OrcRobot), P(OrcSpaceParams), c_double_p, c_double_p]
    lib.orc_valid_segment_count.restype = C.c_uint
    lib.orc_interpolate_state.argtypes = [P(OrcRobot), c_double_p, c_double_p, C.c_double, c_double_p]
    lib.orc_state_distance.argtypes = [P(OrcRobot), c_double_p, c_double_p]
    lib.orc_state_distance.restype = C.c_double
    lib.orc_check_motion.argtypes = [P(OrcRobot), P(OrcSpaceParams), P(OrcGrid), c_double_p, c_double_p,
                                     c_double_p, P(OrcGrid), P(C.c_int), P(C.c_int), c_double_p]
    lib.orc_check_motion_until_invalid.argtypes = [P(OrcRobot), P(OrcSpaceParams), P(OrcGrid), c_double_p, c_double_p,
                                                   c_double_p, P(C.c_int), c_double_p]
    lib.orc_check_motion_discrete.argtypes = [P(OrcRobot), P(OrcSpaceParams), P(OrcGrid), c_double_p, c_double_p,
                                              c_double_p, C.c_int, P(C.c_int), P(C.c_int), c_double_p]
    lib.orc_check_motion_until_invalid_vc.argtypes = [P(OrcRobot), P(OrcSpaceParams), P(OrcGrid), c_double_p, c_double_p,
                                                      c_double_p, C.c_int, P(C.c_int), c_double_p]
    lib.orc_check_motion_discrete_vc.argtypes = [P(OrcRobot), P(OrcSpaceParams), P(OrcGrid), c_double_p, c_double_p,
                                                 c_double_p, C.c_int, C.c_int, P(C.c_int), P(C.c_int), c_double_p]
    lib.orc_check_motion_batch.argtypes = [P(OrcRobot), P(OrcSpaceParams), P(OrcGrid), c_double_p,
                                           c_double_p, c_double_p, C.c_long, P(C.c_uint8),
                                           P(C.c_int32), C.c_int]
    lib.orc_roadmap_create.argtypes = [P(OrcRobot), c_double_p, C.c_long, P(C.c_int32), c_double_p, C.c_long,
                                       P(C.c_int64), P(C.c_uint32), P(C.c_uint64), P(C.c_uint8),
                                       P(C.c_int64), P(C.c_uint32), P(C.c_uint64), P(C.c_uint8)]
    lib.orc_roadmap_create.restype = C.c_void_p
    lib.orc_roadmap_free.argtypes = [C.c_void_p]
    lib.orc_roadmap_free.restype = None
    lib.orc_roadmap_clear_validity.argtypes = [C.c_void_p]
    lib.orc_roadmap_clear_validity.restype = None
    lib.orc_roadmap_get_validity.argtypes = [C.c_void_p, P(C.c_uint8), P(C.c_uint8)]
    lib.orc_roadmap_get_validity.restype = None
    lib.orc_roadmap_query.argtypes = [C.c_void_p, P(OrcGrid), C.c_int, C.c_int, P(C.c_int32), C.c_int, c_double_p,
                                      P(C.c_int), P(C.c_long)]
    lib.orc_check_cached.argtypes = [P(OrcGrid), P(C.c_uint32), P(C.c_uint64), P(C.c_int64), C.c_long,
                                     P(C.c_uint8)]
    lib.orc_grid_export_blocks.argtypes = [P(OrcGrid), P(C.c_uint32), P(C.c_uint64), C.c_long]
    lib.orc_grid_export_blocks.restype = C.c_long
    lib.orc_grid_export_blocks_leaf_order.argtypes = [P(OrcGrid), P(C.c_uint32), P(C.c_uint64), C.c_long]
    lib.orc_grid_export_blocks_leaf_order.restype = C.c_long
    lib.orc_grid_block.argtypes = [P(OrcGrid), C.c_int, C.c_int, C.c_int]
    lib.orc_grid_block.restype = C.c_uint64
    lib.orc_grid_set_block.argtypes = [P(OrcGrid), C.c_int, C.c_int, C.c_int, C.c_uint64]
    lib.orc_grid_set_block.restype = None
    lib.orc_grid_union_block.argtypes = [P(OrcGrid), C.c_int, C.c_int, C.c_int, C.c_uint64]
    lib.orc_grid_union_block.restype = C.c_uint64
    lib.orc_max_threads.restype = C.c_int
    _libs[kind] = lib
    return lib


def _dp(a):
    return a.ctypes.data_as(c_double_p)


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


class Robot:
    """Mirror of tendon::TendonRobot (tendon/TendonRobot.h:52-58) for the oracle."""

    def __init__(self, C_coef, D_coef, r=0.015, L=0.2, dL=0.005, ro=0.01, ri=0.0, E=2.1e6, nu=0.3,
                 max_tension=20.0, min_length=-0.015, max_length=0.035,
                 enable_rotation=False, enable_retraction=False, residual_threshold=5e-6, lib="strict"):
        self.lib = _load(lib)
        C_coef = [list(map(float, c)) for c in C_coef]
        D_coef = [list(map(float, d)) for d in D_coef]
        n = len(C_coef)
        assert n == len(D_coef) and 1 <= n <= MAX_TENDONS
        rb = OrcRobot()
        rb.r, rb.L, rb.dL, rb.ro, rb.ri, rb.E, rb.nu = r, L, dL, ro, ri, E, nu
        rb.n_tendons = n
        rb.n_a, rb.n_m = len(C_coef[0]), len(D_coef[0])
        for j in range(n):
            assert len(C_coef[j]) == rb.n_a and len(D_coef[j]) == rb.n_m
            for i, v in enumerate(C_coef[j]):
                rb.C[j][i] = v
            for i, v in enumerate(D_coef[j]):
                rb.D[j][i] = v
        bc = lambda v: [float(v)] * n if np.isscalar(v) else list(map(float, v))
        for j, (a, b, c) in enumerate(zip(bc(max_tension), bc(min_length), bc(max_length))):
            rb.max_tension[j], rb.min_length[j], rb.max_length[j] = a, b, c
        rb.enable_rotation, rb.enable_retraction = int(enable_rotation), int(enable_retraction)
        rb.residual_threshold = residual_threshold
        self.c = rb
        self.n_tendons = n

    @property
    def state_size(self):
        return self.lib.orc_state_size(C.byref(self.c))

    def t_range(self, s_start=0.0):
        n = self.lib.orc_t_range(s_start, self.c.L, self.c.dL, None, 0)
        out = np.empty(n)
        self.lib.orc_t_range(s_start, self.c.L, self.c.dL, _dp(out), n)
        return out

    def max_points(self):
        return int(self.lib.orc_t_range(0.0, self.c.L, self.c.dL, None, 0)) + 2

    def _new_result(self):
        cap = self.max_points()
        t, p, R = np.zeros(cap), np.zeros((cap, 3)), np.zeros((cap, 9))
        res = OrcResult()
        res.cap = cap
        res.t, res.p, res.R = _dp(t), _dp(p), _dp(R)
        return res, (t, p, R)

    @staticmethod
    def _unpack(res, bufs, n_t):
        t, p, R = bufs
        n = res.n
        return dict(t=t[:n].copy(), p=p[:n].copy(), R=R[:n].copy(), L=res.L,
                    L_i=np.array(res.L_i[:n_t]), u_i=np.array(res.u_i[:]), u_f=np.array(res.u_f[:]),
                    v_i=np.array(res.v_i[:]), v_f=np.array(res.v_f[:]),
                    converged=bool(res.converged), fp_iters=res.fp_iters)

    def shape(self, state):
        state = _f64(state)
        assert state.size == self.state_size
        res, bufs = self._new_result()
        rc = self.lib.orc_shape(C.byref(self.c), _dp(state), C.byref(res))
        assert rc == 0
        return self._unpack(res, bufs, self.n_tendons)

    def home_shape(self, s_start=0.0):
        res, bufs = self._new_result()
        rc = self.lib.orc_home_shape(C.byref(self.c), float(s_start), C.byref(res))
        assert rc == 0
        return self._unpack(res, bufs, self.n_tendons)

    def r_info(self, t):
        n = self.n_tendons
        r, rd, rdd = np.zeros((n, 3)), np.zeros((n, 3)), np.zeros((n, 3))
        self.lib.orc_get_r_info(C.byref(self.c), float(t), r.ctypes.data, rd.ctypes.data, rdd.ctypes.data)
        return r, rd, rdd

    def deriv(self, tau, x, t):
        tau, x = _f64(tau), _f64(x)
        out = np.zeros_like(x)
        self.lib.orc_tendon_deriv(C.byref(self.c), _dp(tau), _dp(x), _dp(out), float(t))
        return out

    def solve_initial_bending(self, tau, s_start=0.0):
        tau = _f64(tau)
        v, u = np.zeros(3), np.zeros(3)
        it = self.lib.orc_solve_initial_bending(C.byref(self.c), _dp(tau), float(s_start), _dp(v), _dp(u))
        return v, u, it

    def base_residual(self, tau, s_start, v0, u0):
        tau, v0, u0 = _f64(tau), _f64(v0), _f64(u0)
        return self.lib.orc_base_residual(C.byref(self.c), _dp(tau), float(s_start), _dp(v0), _dp(u0))

    def collides_self(self, pts):
        pts = _f64(pts)
        return bool(self.lib.orc_collides_self(_dp(pts), pts.shape[0], self.c.r))

    def is_within_length_limits(self, home_Li, fk_Li):
        a, b = _f64(home_Li), _f64(fk_Li)
        return bool(self.lib.orc_is_within_length_limits(C.byref(self.c), _dp(a), _dp(b)))

    def fk_batch(self, states, nthreads=1):
        states = _f64(states)
        n = states.shape[0]
        P = self.max_points() - 2 if not self.c.enable_retraction else self.max_points()
        p = np.empty((n, P, 3))
        L, Li = np.empty(n), np.empty((n, self.n_tendons))
        conv = np.empty(n, dtype=np.uint8)
        self.lib.orc_fk_batch(C.byref(self.c), _dp(states), n, _dp(p), _dp(L), _dp(Li),
                              conv.ctypes.data_as(C.POINTER(C.c_uint8)), P, nthreads)
        return dict(p=p, L=L, L_i=Li, converged=conv.astype(bool))


class Grid:
    """Dense stand-in for collision::VoxelOctree (collision/VoxelOctree.h:68-330)."""

    def __init__(self, N, limits=None, lib="strict", _ptr=None):
        self.lib = _load(lib)
        self.ptr = _ptr if _ptr is not None else self.lib.orc_grid_create(int(N))
        if not self.ptr:
            raise ValueError("unsupported voxel dimension: %r" % N)
        if limits is not None:
            if self.lib.orc_grid_set_limits(self.ptr, *map(float, limits)) != 0:
                raise ValueError("limits must be positive in size")

    def __del__(self):
        try:
            self.lib.orc_grid_free(self.ptr)
        except Exception:
            pass

    N = property(lambda s: s.ptr.contents.N)
    Nb = property(lambda s: s.ptr.contents.Nb)
    limits = property(lambda s: (s.ptr.contents.xmin, s.ptr.contents.xmax, s.ptr.contents.ymin,
                                 s.ptr.contents.ymax, s.ptr.contents.zmin, s.ptr.contents.zmax))
    cell_size = property(lambda s: (s.ptr.contents.dx, s.ptr.contents.dy, s.ptr.contents.dz))

    def empty_copy(self):
        g = Grid.__new__(Grid)
        g.lib = self.lib
        g.ptr = self.lib.orc_grid_empty_copy(self.ptr)
        return g

    def blocks(self):
        """numpy view (Nb,Nb,Nb) uint64 of the dense block array (no copy)."""
        nb = self.Nb
        arr = np.ctypeslib.as_array(self.ptr.contents.blocks, shape=(nb * nb * nb,))
        return arr.reshape(nb, nb, nb)

    def clear(self):
        self.lib.orc_grid_clear(self.ptr)

    def set_cell(self, ix, iy, iz):
        return bool(self.lib.orc_grid_set_cell(self.ptr, ix, iy, iz))

    def cell(self, ix, iy, iz):
        return bool(self.lib.orc_grid_cell(self.ptr, ix, iy, iz))

    def cells(self):
        """sorted list of occupied (ix,iy,iz)."""
        b = self.blocks()
        out = []
        for bx, by, bz in zip(*np.nonzero(b)):
            v = int(b[bx, by, bz])
            for bit in range(64):
                if v >> bit & 1:
                    out.append((4 * bx + bit // 16, 4 * by + (bit // 4) % 4, 4 * bz + bit % 4))
        return sorted((int(a), int(b_), int(c)) for a, b_, c in out)

    def is_in_domain(self, x, y, z):
        return bool(self.lib.orc_grid_is_in_domain(self.ptr, x, y, z))

    def nearest_cell(self, x, y, z):
        out = (C.c_int * 3)()
        self.lib.orc_grid_nearest_cell(self.ptr, x, y, z, out)
        return tuple(out)

    def find_cell(self, x, y, z):
        out = (C.c_int * 3)()
        if self.lib.orc_grid_find_cell(self.ptr, x, y, z, out) != 0:
            raise ValueError("point is out of the voxel dimensions")   # std::domain_error
        return tuple(out)

    def add_point(self, p):
        self.lib.orc_grid_add_point(self.ptr, *map(float, p))

    def add_line(self, a, b):
        a, b = _f64(a), _f64(b)
        self.lib.orc_grid_add_line(self.ptr, _dp(a), _dp(b))

    def add_piecewise_line(self, pts):
        pts = _f64(pts)
        self.lib.orc_grid_add_piecewise_line(self.ptr, _dp(pts), pts.shape[0])

    def add_sphere(self, c, r):
        c = _f64(c)
        self.lib.orc_grid_add_sphere(self.ptr, _dp(c), float(r))

    def add_capsule(self, a, b, r):
        a, b = _f64(a), _f64(b)
        self.lib.orc_grid_add_capsule(self.ptr, _dp(a), _dp(b), float(r))

    def remove_interior(self, keep_diagonal=True):
        self.lib.orc_grid_remove_interior(self.ptr, int(bool(keep_diagonal)))

    def dilate(self, num=1, use_diagonal=False):
        self.lib.orc_grid_dilate(self.ptr, int(num), int(bool(use_diagonal)))

    def dilate_sphere(self, r):
        self.lib.orc_grid_dilate_sphere(self.ptr, float(r))

    def collides(self, other):
        if isinstance(other, Grid):
            rc = self.lib.orc_grid_collides(self.ptr, other.ptr)
            if rc < 0:
                raise ValueError("voxel dimension mismatch")           # std::invalid_argument
            return bool(rc)
        return bool(self.lib.orc_grid_collides_point(self.ptr, *map(float, other)))

    def nblocks(self):
        return int(self.lib.orc_grid_nblocks(self.ptr))

    def ncells(self):
        return int(self.lib.orc_grid_ncells(self.ptr))

    def export_blocks(self, leaf_order=False):
        """Non-zero blocks as (dense block id, mask); leaf_order=True gives them in the reference's visit_leaves
        order (the order its files are written in) instead of ascending block id."""
        f = self.lib.orc_grid_export_blocks_leaf_order if leaf_order else self.lib.orc_grid_export_blocks
        n = f(self.ptr, None, None, 0)
        ids, masks = np.empty(n, dtype=np.uint32), np.empty(n, dtype=np.uint64)
        f(self.ptr, ids.ctypes.data_as(C.POINTER(C.c_uint32)), masks.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        return ids, masks

    def block(self, bx, by, bz):
        return int(self.lib.orc_grid_block(self.ptr, int(bx), int(by), int(bz)))

    def set_block(self, bx, by, bz, value):
        self.lib.orc_grid_set_block(self.ptr, int(bx), int(by), int(bz), int(value))

    def union_block(self, bx, by, bz, value):
        return int(self.lib.orc_grid_union_block(self.ptr, int(bx), int(by), int(bz), int(value)))


IDENTITY = np.eye(3).reshape(9)


def is_valid_state(robot, grid, state, inv_rot=IDENTITY):
    state, inv_rot = _f64(state), _f64(inv_rot).reshape(9)
    tip = np.zeros(3)
    flags = C.c_int(0)
    v = robot.lib.orc_is_valid_state(C.byref(robot.c), grid.ptr, _dp(inv_rot), _dp(state), _dp(tip),
                                     C.byref(flags))
    return bool(v), tip, flags.value


def is_valid_state_spheres(robot, grid, state, inv_rot=IDENTITY):
    """VoxelValidityChecker: the robot voxelised as spheres of its radius at every backbone point."""
    state, inv_rot = _f64(state), _f64(inv_rot).reshape(9)
    tip = np.zeros(3)
    flags = C.c_int(0)
    v = robot.lib.orc_is_valid_state_spheres(C.byref(robot.c), grid.ptr, _dp(inv_rot), _dp(state), _dp(tip), C.byref(flags))
    return bool(v), tip, flags.value


def validate_batch(robot, grid, states, inv_rot=IDENTITY, nthreads=1, lib=None):
    lib = lib or robot.lib
    states, inv_rot = _f64(states), _f64(inv_rot).reshape(9)
    n = states.shape[0]
    valid = np.empty(n, dtype=np.uint8)
    tips = np.empty((n, 3))
    used = lib.orc_validate_batch(C.byref(robot.c), grid.ptr, _dp(inv_rot), _dp(states), n,
                                  valid.ctypes.data_as(C.POINTER(C.c_uint8)), _dp(tips), nthreads)
    return valid.astype(bool), tips, used


def space_params(min_tension_change=0.02, min_rotation_change=0.01, min_retraction_change=0.0001):
    return OrcSpaceParams(min_tension_change, min_rotation_change, min_retraction_change)


def check_motion(robot, grid, a, b, sp=None, inv_rot=IDENTITY, want_swept=False):
    sp = sp or space_params()
    a, b, inv_rot = _f64(a), _f64(b), _f64(inv_rot).reshape(9)
    nfk, fully, lvt = C.c_int(0), C.c_int(0), C.c_double(0)
    swept = grid.empty_copy() if want_swept else None
    v = robot.lib.orc_check_motion(C.byref(robot.c), C.byref(sp), grid.ptr, _dp(inv_rot), _dp(a), _dp(b),
                                   swept.ptr if swept else None, C.byref(nfk), C.byref(fully), C.byref(lvt))
    return dict(valid=(v == 1), domain_error=(v < 0), n_fk=nfk.value, is_fully_valid=bool(fully.value),
                last_valid_t=lvt.value, swept=swept)


def check_motion_until_invalid(robot, grid, a, b, sp=None, inv_rot=IDENTITY, vc_spheres=False):
    """vc_spheres: the installed state checker is VoxelValidityChecker (sphere-swept samples)."""
    sp = sp or space_params()
    a, b, inv_rot = _f64(a), _f64(b), _f64(inv_rot).reshape(9)
    nfk, lvt = C.c_int(0), C.c_double(0)
    fully = robot.lib.orc_check_motion_until_invalid_vc(C.byref(robot.c), C.byref(sp), grid.ptr, _dp(inv_rot), _dp(a), _dp(b),
                                                        int(vc_spheres), C.byref(nfk), C.byref(lvt))
    return dict(is_fully_valid=bool(fully), last_valid_t=lvt.value, n_fk=nfk.value)


def check_motion_discrete(robot, grid, a, b, sp=None, inv_rot=IDENTITY, until_invalid=False, vc_spheres=False):
    sp = sp or space_params()
    a, b, inv_rot = _f64(a), _f64(b), _f64(inv_rot).reshape(9)
    nfk, fully, lvt = C.c_int(0), C.c_int(0), C.c_double(0)
    valid = robot.lib.orc_check_motion_discrete_vc(C.byref(robot.c), C.byref(sp), grid.ptr, _dp(inv_rot), _dp(a), _dp(b),
                                                   int(until_invalid), int(vc_spheres), C.byref(nfk), C.byref(fully), C.byref(lvt))
    return dict(valid=bool(valid), is_fully_valid=bool(fully), last_valid_t=lvt.value, n_fk=nfk.value)


def check_motion_batch(robot, grid, a, b, sp=None, inv_rot=IDENTITY, nthreads=1, lib=None):
    lib = lib or robot.lib
    sp = sp or space_params()
    a, b, inv_rot = _f64(a), _f64(b), _f64(inv_rot).reshape(9)
    n = a.shape[0]
    valid = np.empty(n, dtype=np.uint8)
    nfk = np.empty(n, dtype=np.int32)
    used = lib.orc_check_motion_batch(C.byref(robot.c), C.byref(sp), grid.ptr, _dp(inv_rot), _dp(a), _dp(b),
                                      n, valid.ctypes.data_as(C.POINTER(C.c_uint8)),
                                      nfk.ctypes.data_as(C.POINTER(C.c_int32)), nthreads)
    return valid.astype(bool), nfk, used


def check_cached(grid, block_ids, masks, offsets):
    block_ids = np.ascontiguousarray(block_ids, dtype=np.uint32)
    masks = np.ascontiguousarray(masks, dtype=np.uint64)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    n = offsets.size - 1
    hit = np.empty(n, dtype=np.uint8)
    grid.lib.orc_check_cached(grid.ptr, block_ids.ctypes.data_as(C.POINTER(C.c_uint32)),
                              masks.ctypes.data_as(C.POINTER(C.c_uint64)),
                              offsets.ctypes.data_as(C.POINTER(C.c_int64)), n,
                              hit.ctypes.data_as(C.POINTER(C.c_uint8)))
    return hit.astype(bool)


def omp_lib():
    return _load("omp")


def max_threads():
    return int(_load("omp").orc_max_threads())


class Roadmap:
    """Sequential restatement of VoxelCachedLazyPRM's query loop on a cached roadmap (orc_roadmap_*)."""

    def __init__(self, robot, states, edges, weights, vertex_caches, edge_caches, lib=None):
        self.lib = lib or robot.lib
        self.robot = robot
        self.states = _f64(states)
        self.edges = np.ascontiguousarray(np.asarray(edges).reshape(-1, 2), dtype=np.int32)
        self.w = None if weights is None else _f64(weights)

        def arrs(c, keys):
            present = next((c[k] for k in keys if k in c and c[k] is not None), None)
            return (np.ascontiguousarray(c["offsets"], dtype=np.int64), np.ascontiguousarray(c["block_ids"], dtype=np.uint32),
                    np.ascontiguousarray(c["masks"], dtype=np.uint64),
                    None if present is None else np.ascontiguousarray(present, dtype=np.uint8))
        self._v = arrs(vertex_caches, ("present", "shape_valid"))
        self._e = arrs(edge_caches, ("present", "fully_valid"))
        pa = lambda a, t: a.ctypes.data_as(C.POINTER(t)) if a is not None else None
        self.ptr = self.lib.orc_roadmap_create(
            C.byref(robot.c), _dp(self.states), len(self.states), self.edges.ctypes.data_as(C.POINTER(C.c_int32)),
            _dp(self.w) if self.w is not None else None, len(self.edges),
            pa(self._v[0], C.c_int64), pa(self._v[1], C.c_uint32), pa(self._v[2], C.c_uint64), pa(self._v[3], C.c_uint8),
            pa(self._e[0], C.c_int64), pa(self._e[1], C.c_uint32), pa(self._e[2], C.c_uint64), pa(self._e[3], C.c_uint8))

    def __del__(self):
        try:
            if self.ptr:
                self.lib.orc_roadmap_free(self.ptr)
                self.ptr = None
        except Exception:
            pass

    def clear_validity(self):
        self.lib.orc_roadmap_clear_validity(self.ptr)

    def validity(self):
        v, e = np.zeros(len(self.states), dtype=np.uint8), np.zeros(len(self.edges), dtype=np.uint8)
        self.lib.orc_roadmap_get_validity(self.ptr, v.ctypes.data_as(C.POINTER(C.c_uint8)), e.ctypes.data_as(C.POINTER(C.c_uint8)))
        return v, e

    def query(self, grid, start, goal):
        """-> dict(n, path, cost, iterations, checked): n > 0 solved, 0 disconnected, -2 / -3 invalid start / goal."""
        cap = len(self.states)
        path = np.zeros(cap, dtype=np.int32)
        cost, it, chk = C.c_double(0), C.c_int(0), C.c_long(0)
        n = self.lib.orc_roadmap_query(self.ptr, grid.ptr, int(start), int(goal), path.ctypes.data_as(C.POINTER(C.c_int32)), cap,
                                       C.byref(cost), C.byref(it), C.byref(chk))
        return dict(n=n, path=path[:max(n, 0)].copy(), cost=cost.value, iterations=it.value, checked=chk.value)
