"""Host-side mirror of the reference's `tendon` namespace for the hot path
(cpp/src/tendon/{BackboneSpecs,TendonSpecs,TendonResult,TendonRobot}.h): same names, argument
meaning and error behaviour; all arithmetic runs in libtendon_hip.so on the GPU.
"""
from dataclasses import dataclass, field
from typing import List

import numpy as np

from . import _lib as L
from .engine import Engine


@dataclass
class BackboneSpecs:                       # tendon/BackboneSpecs.h:14-20
    L: float = 0.2
    dL: float = 0.005
    ro: float = 0.01
    ri: float = 0.0
    E: float = 2.1e6
    nu: float = 0.3


@dataclass
class TendonSpecs:                         # tendon/TendonSpecs.h:25-30
    C: List[float] = field(default_factory=lambda: [0.0])
    D: List[float] = field(default_factory=lambda: [0.01])
    max_tension: float = 20.0
    min_length: float = -0.015
    max_length: float = 0.035

    @staticmethod
    def _degree(c, eps):
        for i in range(len(c) - 1, 0, -1):
            if abs(c[i]) > eps:
                return i
        return 0

    def r_degree(self, eps=0.0):
        return self._degree(self.D, eps)

    def theta_degree(self, eps=0.0):
        return self._degree(self.C, eps)

    def is_straight(self, eps=0.0):        # TendonSpecs.h:36-38
        return self.r_degree(eps) == 0 and self.theta_degree(eps) == 0

    def is_helix(self, eps=0.0):           # TendonSpecs.h:40-42
        return self.r_degree(eps) == 0 and self.theta_degree(eps) == 1


@dataclass
class TendonResult:                        # tendon/TendonResult.h:17-40
    t: np.ndarray
    p: np.ndarray                          # (P, 3)
    R: np.ndarray                          # (P, 3, 3) or None
    L: float
    L_i: np.ndarray
    converged: bool = True


class TendonRobot:
    """tendon::TendonRobot (tendon/TendonRobot.h:52-355), batched on the GPU."""

    def __init__(self, tendons=None, specs=None, r=0.015, enable_rotation=False, enable_retraction=False,
                 residual_threshold=5e-6):
        self.r = r
        self.specs = specs or BackboneSpecs()
        self.tendons = list(tendons or [])
        self.enable_rotation = enable_rotation
        self.enable_retraction = enable_retraction
        self.residual_threshold = residual_threshold
        self._engines = {}

    # ---- the reference's robot description files (tendon/TendonRobot.cpp:1012-1089, BackboneSpecs.cpp:7-40,
    # TendonSpecs.cpp:32-54): [tendon_robot], [backbone_specs], [[tendons]] ----------------------------------------------
    def to_toml(self):
        f = lambda x: repr(float(x))
        arr = lambda a: "[" + ", ".join(f(x) for x in a) + "]"
        s = self.specs
        out = ["[tendon_robot]", "radius = " + f(self.r), "enable_rotation = " + str(bool(self.enable_rotation)).lower(),
               "enable_retraction = " + str(bool(self.enable_retraction)).lower(), "residual_threshold = " + f(self.residual_threshold), "",
               "[backbone_specs]", "length = " + f(s.L), "length_discretization = " + f(s.dL), "ro = " + f(s.ro), "ri = " + f(s.ri),
               "E = " + f(s.E), "nu = " + f(s.nu), ""]
        for t in self.tendons:
            out += ["[[tendons]]", "C = " + arr(t.C), "D = " + arr(t.D), "max_tension = " + f(t.max_tension),
                    "min_length = " + f(t.min_length), "max_length = " + f(t.max_length), ""]
        return "\n".join(out)

    @classmethod
    def from_toml(cls, tbl):
        """`tbl`: the parsed table (tomli) or a path to a .toml file.  Missing optional keys keep the reference's defaults
        (TendonRobot.h:53-58); a missing [tendon_robot] / [backbone_specs] entry raises KeyError as cpptoml would throw."""
        if isinstance(tbl, str):
            import tomli
            with open(tbl, "rb") as fh:
                tbl = tomli.load(fh)
        rt, bs = tbl["tendon_robot"], tbl["backbone_specs"]
        specs = BackboneSpecs(L=float(bs["length"]), dL=float(bs["length_discretization"]), ro=float(bs["ro"]), ri=float(bs["ri"]),
                              E=float(bs["E"]), nu=float(bs["nu"]))
        tendons = [TendonSpecs(C=[float(x) for x in t["C"]], D=[float(x) for x in t["D"]], max_tension=float(t["max_tension"]),
                               min_length=float(t["min_length"]), max_length=float(t["max_length"])) for t in tbl.get("tendons", [])]
        return cls(tendons=tendons, specs=specs, r=float(rt["radius"]), enable_rotation=bool(rt.get("enable_rotation", False)),
                   enable_retraction=bool(rt.get("enable_retraction", False)),
                   residual_threshold=float(rt.get("residual_threshold", 5e-6)))

    def state_size(self):                  # TendonRobot.h:60-64
        return len(self.tendons) + int(self.enable_rotation) + int(self.enable_retraction)

    def engine(self, device=0) -> Engine:
        """The GPU context holding this robot's constants (created on first use)."""
        if device not in self._engines:
            self._engines[device] = Engine(self, device)
        return self._engines[device]

    def _t(self, s_start=0.0):
        """t_range(s_start, L, dL) (TendonRobot.cpp:69-84): only to fill TendonResult.t on the host."""
        s = self.specs
        start = min(float(s_start), s.L)
        vals, p = [], start
        while p <= s.L - s.dL / 2:
            vals.append(p)
            p += s.dL
        vals.append(s.L)
        return np.array([s.L - (v - start) for v in vals])[::-1].copy()

    # ---- single-configuration API (reference signatures) -----------------------------------------
    def shape(self, state, device=0) -> TendonResult:          # TendonRobot.h:105-115
        state = np.asarray(state, dtype=np.float64).reshape(-1)
        if state.size != self.state_size():
            raise L.InvalidArgument("State is not the right size")
        out = self.engine(device).fk_batch(state.reshape(1, -1), want_R=True)
        n = int(out["n_points"][0])
        R = out["R"][0, :n].reshape(n, 3, 3).transpose(0, 2, 1)    # column-major storage -> R[j][r][c]
        s_start = float(state[-1]) if self.enable_retraction else 0.0
        t = self._t(s_start) if n > 1 else np.array([min(s_start, self.specs.L)])
        return TendonResult(t=t[:n], p=out["p"][0, :n], R=R, L=float(out["L"][0]),
                            L_i=out["L_i"][0], converged=bool(out["converged"][0]))

    def forward_kinematics(self, state, device=0):              # TendonRobot.h:68-72
        return self.shape(state, device).p

    def home_shape(self, s_start=0.0, device=0) -> TendonResult:  # TendonRobot.cpp:249-314
        """Zero-tension shape.  Pure host arithmetic (closed forms; the defined Simpson rule of DESIGN.md
        for general routing); s_start = 0 lengths come from the engine so both agree bit for bit."""
        L = self.specs.L
        s_start = min(max(float(s_start), 0.0), L)
        n_t = len(self.tendons)
        if s_start == L:
            return TendonResult(t=np.array([L]), p=np.zeros((1, 3)), R=np.eye(3)[None], L=0.0, L_i=np.zeros(n_t))
        t = self._t(s_start)
        p = np.zeros((t.size, 3))
        p[:, 2] = t - s_start
        R = np.tile(np.eye(3), (t.size, 1, 1))
        if s_start == 0.0:
            L_i = self.engine(device).home_lengths()
        else:
            L_i = np.empty(n_t)
            for i, td in enumerate(self.tendons):
                if td.is_straight():
                    L_i[i] = L - s_start
                elif td.is_helix():
                    L_i[i] = (L - s_start) * np.sqrt(1 + td.D[0] * td.D[0] * td.C[1] * td.C[1])
                else:
                    Cd = np.polynomial.Polynomial(td.C).deriv()
                    D, Dd = np.polynomial.Polynomial(td.D), np.polynomial.Polynomial(td.D).deriv()
                    vals = np.sqrt(Dd(t) ** 2 + D(t) ** 2 * Cd(t) ** 2 + 1)
                    n, dx = t.size, self.specs.dL
                    nint, odd = n - 1, 0.0
                    if nint % 2:
                        odd = 0.5 * dx * (vals[n - 2] + vals[n - 1])
                        nint -= 1
                    if nint == 0:
                        L_i[i] = odd
                    else:
                        w = np.where(np.arange(1, nint) % 2 == 1, 4.0, 2.0)
                        L_i[i] = odd + (vals[0] + vals[nint] + (w * vals[1:nint]).sum()) * dx / 3.0
        return TendonResult(t=t, p=p, R=R, L=L - s_start, L_i=L_i, converged=True)

    def calc_dl(self, home_l, other_l):                        # TendonRobot.h:247-259
        home_l, other_l = np.asarray(home_l, float), np.asarray(other_l, float)
        if home_l.shape != other_l.shape:
            raise L.OutOfRange("vector size mismatch")
        return home_l - other_l

    def is_within_length_limits(self, dl):                     # TendonRobot.h:268-278
        dl = np.asarray(dl, float)
        if dl.shape[-1] != len(self.tendons):
            raise L.OutOfRange("length mismatch")
        lo = np.array([t.min_length for t in self.tendons])
        hi = np.array([t.max_length for t in self.tendons])
        return bool(np.all(~((dl < lo) | (hi < dl))))

    def shape_and_lengths(self, state, home_shape=None, device=0):     # TendonRobot.h:231-247
        """(shape(state), dl = home L_i - shape L_i): what is_valid compares with the tendons' length limits."""
        st = np.asarray(state, float).reshape(-1)
        home = home_shape if home_shape is not None else self.home_shape(float(st[-1]) if self.enable_retraction else 0.0, device)
        sh = self.shape(st, device)
        return sh, self.calc_dl(home.L_i, sh.L_i)

    def collides_self(self, shape, device=0):                          # TendonRobot.h:229, collision/collision.cpp:6-46
        """collision::collides_self of one shape (a TendonResult or a (P, 3) point array with this robot's point count): the
        exact capsule sweep of the validity predicate, run on the GPU over the given points (tr_validate_shapes_dev)."""
        import torch
        if self.enable_retraction:
            raise L.Unsupported("collides_self of a given shape: robots with retraction have per-shape point counts (use is_valid)")
        eng = self.engine(device)
        P, N = eng.num_points, eng.n_tendons
        pts = np.asarray(shape.p if hasattr(shape, "p") else shape, dtype=np.float64)
        if pts.shape != (P, 3):
            raise L.InvalidArgument("the shape must have this robot's %d backbone points" % P)
        dev = "cuda:%d" % eng.device
        planes = torch.zeros((3, P, 64), dtype=torch.float64, device=dev)
        planes[:, :, 0] = torch.from_numpy(np.ascontiguousarray(pts.T)).to(dev)
        # tendon lengths that pass the length limits whatever they are, so that the flags report the self-collision test
        lo = np.array([t.min_length for t in self.tendons]); hi = np.array([t.max_length for t in self.tendons])
        Li = torch.zeros((N, 64), dtype=torch.float64, device=dev)
        Li[:, 0] = torch.from_numpy(eng.home_lengths() - np.clip(0.0, lo, hi)).to(dev)
        conv = torch.ones(64, dtype=torch.uint8, device=dev)
        bits = torch.zeros(1, dtype=torch.int64, device=dev)
        flags = torch.zeros(64, dtype=torch.uint8, device=dev)
        eng.validate_shapes_dev(1, 64, planes[0], planes[1], planes[2], Li, conv, bits, flags, check_voxels=False)
        torch.cuda.synchronize(eng.device)
        fl = int(flags[0].item())
        if not (fl & 2):
            raise L.TendonHipError("internal: the length limits rejected the stand-in lengths")
        return not (fl & 4)

    def random_state(self, rng=None):                                  # TendonRobot.cpp:219-246
        """Uniform in the state space: tensions in [0, max_tension], rotation in [-pi, pi], retraction in [0, L]."""
        rng = np.random.default_rng() if rng is None else rng
        st = [rng.uniform(0.0, t.max_tension) for t in self.tendons]
        if self.enable_rotation:
            st.append(rng.uniform(-np.pi, np.pi))
        if self.enable_retraction:
            st.append(rng.uniform(0.0, self.specs.L))
        return np.array(st)

    def read_config_csv(self, stream):                                 # TendonRobot.cpp:976-1002
        """Robot states from a CSV with a header: columns tau_1 .. tau_N (+ theta with rotation, + s_start with retraction),
        in any order and among other columns -- the input of the reference's batch tools; returns (n, state_size)."""
        import csv
        rd = csv.reader(stream)
        header = [h.strip() for h in next(rd)]
        names = ["tau_%d" % (i + 1) for i in range(len(self.tendons))]
        if self.enable_rotation:
            names.append("theta")
        if self.enable_retraction:
            names.append("s_start")
        try:
            idx = [header.index(nm) for nm in names]
        except ValueError as e:
            raise L.OutOfRange("missing CSV column: %s" % e)
        rows = [[float(r[i]) for i in idx] for r in rd if r]
        return np.array(rows, dtype=np.float64).reshape(len(rows), len(names))

    def load_config_csv(self, path):                                   # TendonRobot.cpp:1004-1010
        with open(path, newline="") as f:
            return self.read_config_csv(f)

    def __eq__(self, other):                                           # TendonRobot.h:280-286
        return (isinstance(other, TendonRobot) and self.r == other.r and self.specs == other.specs and self.tendons == other.tendons
                and self.enable_rotation == other.enable_rotation and self.enable_retraction == other.enable_retraction
                and self.residual_threshold == other.residual_threshold)

    __hash__ = object.__hash__

    # ---- batched API ---------------------------------------------------------------------------
    def shape_batch(self, states, want_R=False, device=0):
        """Batched shape(): dict(p (n,P,3), R, L, L_i, converged, n_points)."""
        return self.engine(device).fk_batch(states, want_R=want_R)

    def forward_kinematics_batch(self, states, device=0):
        """The omp loop of apps/estimate_length_discretization.cpp:62-71 as one launch."""
        return self.engine(device).fk_batch(states)["p"]

    def tip_jacobian_batch(self, states, delta=1e-6, device=0):
        """Central-difference tip Jacobians d tip / d state for a batch of states: (n, 3, S).

        This is the Jacobian levmar forms inside tip_control::inverse_kinematics
        (tip-control/tip_control.cpp:35-140 passes opts[4] = -finite_difference_delta, i.e. central
        differences; 3rdparty/levmar-2.6/misc_core.c:175-211): per column j, d = max(1e-4 |p_j|, delta),
        J[:, j] = (tip(p + d e_j) - tip(p - d e_j)) * (0.5 / d) -- but all 2*S*n perturbed states go
        through ONE K1 launch instead of 2*S sequential FK calls per state."""
        st = np.ascontiguousarray(np.asarray(states, dtype=np.float64))
        if st.ndim == 1:
            st = st.reshape(1, -1)
        n, S = st.shape
        if S != self.state_size():
            raise L.InvalidArgument("State is not the right size")
        d = np.maximum(np.abs(1e-4 * st), delta)                         # (n, S)
        pert = np.repeat(st[:, None, None, :], S, axis=1).repeat(2, axis=2)   # (n, S, 2, S)
        j = np.arange(S)
        pert[:, j, 0, j] = st[:, j] - d[:, j]
        pert[:, j, 1, j] = st[:, j] + d[:, j]
        out = self.engine(device).fk_batch(pert.reshape(-1, S))
        npts = out["n_points"]
        tips = out["p"][np.arange(len(npts)), npts - 1].reshape(n, S, 2, 3)
        if self.enable_retraction:
            # tip_control's FK wrapper: s_start beyond L returns (0, 0, L - s_start)  (tip_control.cpp:96-104)
            s = pert[..., -1]
            over = s > self.specs.L
            tips[over] = 0.0
            tips[over, 2] = (self.specs.L - s)[over]
        J = (tips[:, :, 1, :] - tips[:, :, 0, :]) * (0.5 / d)[:, :, None]   # (n, S, 3)
        return np.transpose(J, (0, 2, 1))
