"""Host-side mirror of the reference's validity plug-ins for the hot path
(cpp/src/motion-planning/{VoxelEnvironment, AbstractValidityChecker, AbstractVoxelValidityChecker,
VoxelBackboneValidityChecker, AbstractVoxelMotionValidator, VoxelBackboneMotionValidator}):
same names and semantics, operating on robot states (`std::vector<double>` in the reference,
AbstractValidityChecker.cpp:50-78) -- one or a batch.  OMPL itself is not a dependency here; the
OMPL-facing virtuals (isValid(const State*), checkMotion(s1, s2)) are the C++ shim shown in
INTEGRATION.md.
"""
import time
from dataclasses import dataclass, field

import numpy as np

from . import _lib as L
from .collision import VoxelOctree
from .tendon import TendonRobot


class FunctionTimer:
    """util::FunctionTimer (util/FunctionTimer.h:14-91): one wall-time sample per timed call.
    Here a call is a whole batch."""

    def __init__(self):
        self.times = []

    def time(self, f, *a, **k):
        t0 = time.perf_counter()
        try:
            return f(*a, **k)
        finally:
            self.times.append(time.perf_counter() - t0)

    def get_times(self): return self.times
    def clear(self): self.times = []


@dataclass
class VoxelEnvironment:                    # motion-planning/VoxelEnvironment.h:31-49
    filename: str = ""
    scaling: float = 1.0
    translation: np.ndarray = field(default_factory=lambda: np.zeros(3))
    inv_rotation: np.ndarray = field(default_factory=lambda: np.eye(3))
    interior_fname: str = ""
    _obstacle_cache: VoxelOctree = None

    def set_obstacle_cache(self, obstacles):
        self._obstacle_cache = obstacles

    def get_obstacles(self):
        """VoxelEnvironment::get_obstacles: the cached set, else the file named by `filename` (the reference's JSON /
        msgpack / TOML layouts, collision/VoxelOctree.cpp:1357-1497; .nrrd images need ITK and are not read here)."""
        if self._obstacle_cache is None:
            if not self.filename:
                raise L.InvalidArgument("VoxelEnvironment has neither an obstacle cache nor a filename")
            self._obstacle_cache = VoxelOctree.from_file(self.filename)
        return self._obstacle_cache

    # ---- the [voxel_environment] table of the reference's problem files (VoxelEnvironment.cpp:17-103): filename (or the voxel
    # set inline), interior_filename, scaling, translation, rotation_quat = (w, x, y, z) of inv_rotation ----------------------
    @staticmethod
    def _quat_from_matrix(R):
        """Unit quaternion (w, x, y, z) of a rotation matrix, largest component first (the branch structure every
        matrix-to-quaternion conversion uses; q and -q are the same rotation)."""
        R = np.asarray(R, float)
        t = R[0, 0] + R[1, 1] + R[2, 2]
        if t > 0:
            s = 0.5 / np.sqrt(t + 1.0)
            q = [0.25 / s, (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s]
        else:
            i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
            j, k = (i + 1) % 3, (i + 2) % 3
            s = 2.0 * np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
            v = [0.0, 0.0, 0.0]
            v[i] = 0.25 * s; v[j] = (R[j, i] + R[i, j]) / s; v[k] = (R[k, i] + R[i, k]) / s
            q = [(R[k, j] - R[j, k]) / s] + v
        return np.array(q)

    @staticmethod
    def _matrix_from_quat(q):
        w, x, y, z = (float(c) for c in q)
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def to_toml(self):
        f = lambda x: repr(float(x))
        arr = lambda a: "[" + ", ".join(f(x) for x in a) + "]"
        out = ["[voxel_environment]"]
        inline = ""
        if not self.filename and self._obstacle_cache is not None:
            inline = self._obstacle_cache.to_toml()              # the voxel set stored in the file itself (:27-33)
        else:
            out.append('filename = "%s"' % self.filename)
        out += ['interior_filename = "%s"' % self.interior_fname, "scaling = " + f(self.scaling), "translation = " + arr(self.translation),
                "rotation_quat = " + arr(self._quat_from_matrix(self.inv_rotation)), ""]
        return "\n".join(out) + ("\n" + inline if inline else "")

    @classmethod
    def from_toml(cls, tbl):
        """`tbl`: the parsed table (tomli) or a path; either the top-level table holding [voxel_environment] or that table."""
        if isinstance(tbl, str):
            import tomli
            with open(tbl, "rb") as fh:
                tbl = tomli.load(fh)
        top = tbl
        if "voxel_environment" in tbl:
            tbl = tbl["voxel_environment"]
        env = cls()
        if "filename" in tbl:
            env.filename = str(tbl["filename"])
        else:
            env.set_obstacle_cache(VoxelOctree.from_toml(top))
        env.scaling = float(tbl["scaling"])
        env.translation = np.array([float(x) for x in tbl["translation"]])
        env.inv_rotation = cls._matrix_from_quat(tbl["rotation_quat"])
        env.interior_fname = str(tbl.get("interior_filename", ""))
        return env

    def rotate_point(self, p):             # VoxelEnvironment.cpp:125-127
        return np.asarray(self.inv_rotation, float) @ np.asarray(p, float)

    def rotate_points(self, pts):          # VoxelEnvironment.cpp:129-131
        return (np.asarray(self.inv_rotation, float) @ np.asarray(pts, float).T).T


class VoxelBackboneValidityChecker:
    """motion-planning/VoxelBackboneValidityChecker.h:28-58 over AbstractVoxelValidityChecker /
    AbstractValidityChecker.  `is_valid(states)` is AbstractValidityChecker::isValid
    (AbstractValidityChecker.cpp:124-133) for a batch: one K1 + one K2 launch.
    Every checker owns its engine context -- robot constants as they are now, its own obstacle grid and
    inv_rotation -- like the reference's checkers own their `_voxels` (AbstractVoxelValidityChecker.h:63-64): a
    second checker on the same robot cannot replace this one's environment."""

    def __init__(self, robot: TendonRobot, venv: VoxelEnvironment, voxels: VoxelOctree, device=0):
        from .engine import Engine
        self._robot, self._venv, self._voxels = robot, venv, voxels
        self._timers = {k: FunctionTimer() for k in
                        ("fk", "collision", "self_collision", "voxelize", "collision-without-voxelizing", "is_valid")}
        self.engine = Engine(robot, device)
        # constructor check of VoxelBackboneValidityChecker.h:37-45 is enforced by tr_set_grid
        self.engine.set_grid(voxels.Nx(), voxels.limits(), voxels.blocks, venv.inv_rotation)

    def robot(self): return self._robot
    def timers(self): return self._timers
    def timer(self, name): return self._timers[name]
    def calls(self, name): return len(self._timers[name].get_times())

    def clear_timing(self):
        for t in self._timers.values():
            t.clear()

    def fk(self, robot_states):
        """AbstractValidityChecker::fk (AbstractValidityChecker.cpp:80-97) for a batch."""
        return self._timers["fk"].time(self.engine.fk_batch, robot_states)

    def is_valid(self, robot_states):
        """bool[n]: full state validity (FK, converged, lengths, self collision, voxel collision)."""
        return self._timers["is_valid"].time(self.engine.validate_batch, robot_states, False, False)["valid"]

    def is_valid_detail(self, robot_states):
        """valid, tips (fk_shape.p.back(), VoxelCachedLazyPRM.cpp:1438) and TR_FLAG_* bits."""
        return self._timers["is_valid"].time(self.engine.validate_batch, robot_states, True, True)

    # ---- obstacle-set edits on the device (collision::VoxelOctree's add_sphere / dilate / remove_interior,
    # as apps/prepare_voxel_env.cpp:247-315 applies them); the resident grid is the checker's obstacle set ----
    def add_spheres(self, spheres):
        """VoxelOctree::add_sphere for rows (cx, cy, cz, r)."""
        self.engine.grid_add_spheres(spheres)

    def add_capsules(self, capsules):
        """VoxelOctree::add_capsule for rows (ax, ay, az, bx, by, bz, r)."""
        self.engine.grid_add_capsules(capsules)

    def dilate(self, num=1, use_diagonal=False):
        self.engine.grid_dilate(num, use_diagonal)

    def dilate_sphere(self, r):
        self.engine.grid_dilate_sphere(r)

    def remove_interior(self, keep_diagonal=True):
        self.engine.grid_remove_interior(keep_diagonal)

    def obstacles(self):
        """The obstacle set as it is on the device now (a VoxelOctree with the same limits)."""
        v = self._voxels.empty_copy()
        v.blocks[...] = self.engine.get_grid().reshape(v.blocks.shape)
        return v

    def isValid(self, robot_state):
        """Single state, reference spelling."""
        return bool(self.is_valid(np.asarray(robot_state, float).reshape(1, -1))[0])


class VoxelValidityChecker(VoxelBackboneValidityChecker):
    """motion-planning/VoxelValidityChecker.h:18-26: the same chain, but the robot is voxelised as a sphere of
    its radius at every backbone point (add_sphere) and tested against the raw, un-dilated environment."""

    def __init__(self, robot: TendonRobot, venv: VoxelEnvironment, voxels: VoxelOctree, device=0):
        from .engine import Engine
        self._robot, self._venv, self._voxels = robot, venv, voxels
        self._timers = {k: FunctionTimer() for k in
                        ("fk", "collision", "self_collision", "voxelize", "collision-without-voxelizing", "is_valid")}
        self.engine = Engine(robot, device)
        self.engine.set_checker(True)            # before the grid: the dL <= voxel check belongs to the backbone checker only
        self.engine.set_grid(voxels.Nx(), voxels.limits(), voxels.blocks, venv.inv_rotation)


class VoxelBackboneMotionValidator:
    """motion-planning/VoxelBackboneMotionValidator.{h,cpp} over AbstractVoxelMotionValidator:
    `check_motion(a, b)` is checkMotion(s1, s2) (AbstractVoxelMotionValidator.h:143-151) for a
    batch of edges.  As in the reference (Problem.h:175-210 installs any state checker next to this validator),
    check_motion sweeps the BACKBONE against the checker's voxels whichever checker it is, while
    check_motion_last_valid asks the checker itself about every sample (`_vc->collides`,
    VoxelBackboneMotionValidator.cpp:83-91) -- the sphere-swept robot for a VoxelValidityChecker."""

    def __init__(self, checker: VoxelBackboneValidityChecker, min_tension_change=0.02,
                 min_rotation_change=0.01, min_retraction_change=0.0001):
        self._vc = checker
        self.engine = checker.engine
        self.min_tension_change = min_tension_change          # Problem.h:59
        self.min_rotation_change = min_rotation_change        # Problem.h:61
        self.min_retraction_change = min_retraction_change    # Problem.h:62
        self._timers = {"voxelize-swept-volume": FunctionTimer(), "collision-swept-volume": FunctionTimer()}
        self._num_voxelize_errors = 0

    def validity_checker(self): return self._vc
    def timers(self): return self._timers
    def num_voxelize_errors(self): return self._num_voxelize_errors

    def check_motion_detail(self, a, b):
        return self._timers["voxelize-swept-volume"].time(
            self.engine.validate_edges, a, b, self.min_tension_change, self.min_rotation_change,
            self.min_retraction_change)

    def check_motion(self, a, b):
        return self.check_motion_detail(a, b)["valid"]

    def check_motion_indexed(self, states, edges):
        """checkMotion for roadmap edges given as index pairs into one vertex array: same verdicts and n_fk, but
        every vertex is evaluated once for all of its edges (tr_validate_edges_indexed)."""
        return self._timers["voxelize-swept-volume"].time(
            self.engine.validate_edges_indexed, states, edges, self.min_tension_change, self.min_rotation_change,
            self.min_retraction_change)

    def check_motion_last_valid(self, a, b):
        """checkMotion(s1, s2, last_valid) for a batch: (valid, last_valid_t); last_valid.first is
        interpolate(s1, s2, last_valid_t) in the caller's state space."""
        d = self.engine.validate_edges_last_valid(a, b, self.min_tension_change, self.min_rotation_change,
                                                  self.min_retraction_change)
        return d["valid"], d["last_valid_t"]

    def checkMotion(self, s1, s2):
        return bool(self.check_motion(np.asarray(s1, float).reshape(1, -1), np.asarray(s2, float).reshape(1, -1))[0])


class VoxelBackboneDiscreteMotionValidator(VoxelBackboneMotionValidator):
    """motion-planning/VoxelBackboneDiscreteMotionValidator.{h,cpp}: the same interface, but an edge is
    sampled at a, interpolate(i / validSegmentCount), b instead of bisected adaptively."""

    def check_motion_detail(self, a, b, last_valid=False):
        return self._timers["voxelize-swept-volume"].time(
            self.engine.validate_edges_discrete, a, b, self.min_tension_change, self.min_rotation_change,
            self.min_retraction_change, last_valid)

    def check_motion_last_valid(self, a, b):
        d = self.check_motion_detail(a, b, last_valid=True)
        return d["valid"], d["last_valid_t"]


class Environment:
    """motion_planning::Environment (motion-planning/Environment.h): the obstacle primitives of a problem file -- points, spheres
    (centre, radius), capsules (a, b, radius); meshes are kept as their tables (rasterising them needs FCL / ITK and is out of
    scope).  `voxelize` is Environment::voxelize (Environment.cpp:62-100): an empty copy of the reference grid with every
    primitive rasterised -- on the host mirror, or straight into a checker's resident grid."""

    def __init__(self, points=(), spheres=(), capsules=(), meshes=()):
        self.points = [np.asarray(p, float) for p in points]
        self.spheres = [(np.asarray(c, float), float(r)) for c, r in spheres]
        self.capsules = [(np.asarray(a, float), np.asarray(b, float), float(r)) for a, b, r in capsules]
        self.meshes = list(meshes)

    def to_toml(self):
        f = lambda x: repr(float(x))
        arr = lambda a: "[" + ", ".join(f(x) for x in a) + "]"
        out = ["[environment]", ""]
        for p in self.points:
            out += ["[[environment.points]]", "point = " + arr(p), ""]
        for c, r in self.spheres:
            out += ["[[environment.spheres]]", "[environment.spheres.sphere]", "radius = " + f(r), "center = " + arr(c), ""]
        for a, b, r in self.capsules:
            out += ["[[environment.capsules]]", "[environment.capsules.capsule]", "radius = " + f(r), "a = " + arr(a), "b = " + arr(b), ""]
        return "\n".join(out)

    @classmethod
    def from_toml(cls, tbl):
        if isinstance(tbl, str):
            import tomli
            with open(tbl, "rb") as fh:
                tbl = tomli.load(fh)
        tbl = tbl.get("environment", tbl)
        inner = lambda t, key: t.get(key, t)
        return cls(points=[t["point"] for t in tbl.get("points", [])],
                   spheres=[(inner(t, "sphere")["center"], inner(t, "sphere")["radius"]) for t in tbl.get("spheres", [])],
                   capsules=[(inner(t, "capsule")["a"], inner(t, "capsule")["b"], inner(t, "capsule")["radius"]) for t in tbl.get("capsules", [])],
                   meshes=tbl.get("meshes", []))

    def _rows(self):
        sph = [list(p) + [0.0] for p in self.points] + [list(c) + [r] for c, r in self.spheres]       # a point is add_point = a sphere of radius 0
        cap = [list(a) + list(b) + [r] for a, b, r in self.capsules]
        return np.array(sph, float).reshape(-1, 4), np.array(cap, float).reshape(-1, 7)

    def voxelize(self, reference: VoxelOctree):
        if self.meshes:
            raise L.Unsupported("mesh obstacles are not rasterised here")
        v = reference.empty_copy()
        for p in self.points:
            v.add_point(p)
        for c, r in self.spheres:
            v.add_sphere(c, r)
        for a, b, r in self.capsules:
            v.add_capsule(a, b, r)
        return v

    def voxelize_into(self, checker, clear=True):
        """The same on the checker's resident grid (tr_grid_add_spheres / tr_grid_add_capsules)."""
        if self.meshes:
            raise L.Unsupported("mesh obstacles are not rasterised here")
        if clear:
            vox = checker._voxels
            checker.engine.set_grid(vox.Nx(), vox.limits(), np.zeros_like(vox.blocks), checker._venv.inv_rotation)
        sph, cap = self._rows()
        if len(sph):
            checker.add_spheres(sph)
        if len(cap):
            checker.add_capsules(cap)


class Problem:
    """motion_planning::Problem (motion-planning/Problem.h:41-95, Problem.cpp:420-560): the planner's input file -- robot,
    obstacle primitives, voxel environment, start and goal, the state-space resolutions of the motion validator."""

    def __init__(self, robot=None, env=None, venv=None, start=(), goal=(), min_tension_change=0.02, min_rotation_change=0.01,
                 min_retraction_change=0.0001, start_rotation=0.0, start_retraction=0.0, goal_rotation=0.0, goal_retraction=0.0,
                 sample_like_sphere=True):
        self.robot = robot if robot is not None else TendonRobot()
        self.env = env if env is not None else Environment()
        self.venv = venv if venv is not None else VoxelEnvironment()
        self.start, self.goal = [float(x) for x in start], [float(x) for x in goal]
        self.min_tension_change, self.min_rotation_change, self.min_retraction_change = min_tension_change, min_rotation_change, min_retraction_change
        self.start_rotation, self.start_retraction, self.goal_rotation, self.goal_retraction = start_rotation, start_retraction, goal_rotation, goal_retraction
        self.sample_like_sphere = sample_like_sphere

    def _state(self, tau, rot, ret):                      # Problem.h:70-83
        st = list(tau)
        if self.robot.enable_rotation:
            st.append(rot)
        if self.robot.enable_retraction:
            st.append(ret)
        return np.array(st, float)

    def start_state(self): return self._state(self.start, self.start_rotation, self.start_retraction)
    def goal_state(self): return self._state(self.goal, self.goal_rotation, self.goal_retraction)

    def to_toml(self):
        f = lambda x: repr(float(x))
        arr = lambda a: "[" + ", ".join(f(x) for x in a) + "]"
        out = ["[problem]", "start = " + arr(self.start), "goal = " + arr(self.goal), "min_tension_change = " + f(self.min_tension_change),
               "min_rotation_change = " + f(self.min_rotation_change), "min_retraction_change = " + f(self.min_retraction_change),
               "start_rotation = " + f(self.start_rotation), "start_retraction = " + f(self.start_retraction),
               "goal_rotation = " + f(self.goal_rotation), "goal_retraction = " + f(self.goal_retraction),
               "sample_like_sphere = " + str(bool(self.sample_like_sphere)).lower(), ""]
        return "\n".join(out) + "\n" + self.robot.to_toml() + "\n" + self.env.to_toml() + "\n" + self.venv.to_toml()

    @classmethod
    def from_toml(cls, tbl):
        if isinstance(tbl, str):
            import tomli
            with open(tbl, "rb") as fh:
                tbl = tomli.load(fh)
        pt = tbl["problem"]
        robot = TendonRobot.from_toml(tbl)
        for key, flag in (("start_rotation", robot.enable_rotation), ("goal_rotation", robot.enable_rotation),
                          ("start_retraction", robot.enable_retraction), ("goal_retraction", robot.enable_retraction)):
            if flag and key not in pt:
                raise L.OutOfRange("Must specify %s if %s is enabled" % (key, "rotation" if "rotation" in key else "retraction"))
        return cls(robot=robot, env=Environment.from_toml(tbl), venv=VoxelEnvironment.from_toml(tbl) if "voxel_environment" in tbl else None,
                   start=pt["start"], goal=pt["goal"], min_tension_change=float(pt["min_tension_change"]),
                   min_rotation_change=float(pt.get("min_rotation_change", 0.01)), min_retraction_change=float(pt.get("min_retraction_change", 0.0001)),
                   start_rotation=float(pt.get("start_rotation", 0.0)), start_retraction=float(pt.get("start_retraction", 0.0)),
                   goal_rotation=float(pt.get("goal_rotation", 0.0)), goal_retraction=float(pt.get("goal_retraction", 0.0)),
                   sample_like_sphere=bool(pt.get("sample_like_sphere", True)))

    # ---- plans: the planners' output (Problem.cpp:350-418): CSV with the columns i, tau_1 .. tau_N [, theta] [, s_start] ----------
    def write_plan(self, stream, plan):
        names = ["tau_%d" % (i + 1) for i in range(len(self.robot.tendons))]
        if self.robot.enable_rotation:
            names.append("theta")
        if self.robot.enable_retraction:
            names.append("s_start")
        plan = np.asarray(plan, float).reshape(-1, len(names))
        stream.write(",".join(["i"] + names) + "\n")
        for i, row in enumerate(plan):
            stream.write(",".join([str(i + 1)] + [repr(float(x)) for x in row]) + "\n")

    def save_plan(self, path, plan):
        with open(path, "w", newline="") as f:
            self.write_plan(f, plan)

    @staticmethod
    def read_plan(stream):
        """Columns tau_1, tau_2, ... as far as they go, then theta and s_start when present (Problem::read_plan)."""
        import csv
        rd = csv.reader(stream)
        header = [h.strip() for h in next(rd)]
        idx = []
        i = 1
        while "tau_%d" % i in header:
            idx.append(header.index("tau_%d" % i)); i += 1
        for nm in ("theta", "s_start"):
            if nm in header:
                idx.append(header.index(nm))
        rows = [[float(r[j]) for j in idx] for r in rd if r]
        return np.array(rows, float).reshape(len(rows), len(idx))

    @classmethod
    def load_plan(cls, path):
        with open(path, newline="") as f:
            return cls.read_plan(f)

    def plan_from_path(self, states, path_vertices):
        """The states along a roadmap path (vertex indices, as VoxelCachedLazyPRM.solveWithRoadmap returns them) as a plan."""
        return np.asarray(states, float)[np.asarray(path_vertices, dtype=np.int64)]

    def voxel_backbone_checker(self, voxels=None, device=0, spheres=False):
        """Problem::set_voxel_backbone_state_checker / set_voxel_state_checker (Problem.h:175-216): the state checker over the voxel
        environment's obstacles and the motion validator with this problem's resolutions -> (checker, motion validator)."""
        vox = voxels if voxels is not None else self.venv.get_obstacles()
        chk = (VoxelValidityChecker if spheres else VoxelBackboneValidityChecker)(self.robot, self.venv, vox, device)
        mv = VoxelBackboneMotionValidator(chk, self.min_tension_change, self.min_rotation_change, self.min_retraction_change)
        return chk, mv
