"""Batched tip-position inverse kinematics over the GPU forward kinematics.

The reference's `tip_control::inverse_kinematics` (tip-control/tip_control.cpp:35-140,341-396) hands an FK
callback to the vendored levmar (`dlevmar_bc_dif`, 3rdparty/levmar-2.6): box-constrained Levenberg-Marquardt
with a central-difference Jacobian, 2 S + 1 sequential FK calls per iteration, one start state at a time;
`VoxelCachedLazyPRM::roadmapIk` (motion-planning/VoxelCachedLazyPRM.cpp:3095-3577) runs it from the k
roadmap vertices whose tips are nearest to the goal.

Here every iteration of every start state is ONE K1 launch: the trial points of all still-active problems
together with their 2 S perturbations.  The iteration is the damped Gauss-Newton scheme levmar documents
(Madsen, Nielsen, Tingleff: "Methods for non-linear least squares problems" -- mu = mu_init * max diag(J^T J),
gain ratio rho, mu *= max(1/3, 1 - (2 rho - 1)^3) on acceptance, mu *= nu, nu *= 2 on rejection) with the box
constraints handled by projection; it is NOT levmar's code path (projected-gradient fallbacks, line search),
so iterates are not comparable step by step -- the contract is the result: the same stopping thresholds, the
same bounds (`Bounds.from_robot`), the same FK wrapper for retraction beyond L, the same canonical rotation.
Names, argument order, defaults and result fields follow tip_control.h:27-107.
"""
from dataclasses import dataclass

import numpy as np

from . import _lib as L


@dataclass
class IKResult:                               # tip_control.h:40-46
    state: np.ndarray
    tip: np.ndarray
    error: float
    iters: int
    num_fk_calls: int


@dataclass
class Bounds:                                 # tip_control.h:48-58, tip_control.cpp:160-185
    lower: np.ndarray
    upper: np.ndarray

    @staticmethod
    def from_robot(robot):
        n, m = len(robot.tendons), robot.state_size()
        lo, hi = np.zeros(m), np.zeros(m)
        hi[:n] = [t.max_tension for t in robot.tendons]
        if robot.enable_rotation:
            lo[n], hi[n] = np.finfo(float).min, np.finfo(float).max
        if robot.enable_retraction:
            hi[-1] = robot.specs.L
        return Bounds(lo, hi)

    def center_about_state(self, state):
        self.lower = self.lower - np.asarray(state, float)
        self.upper = self.upper - np.asarray(state, float)


def canonical_angle(theta):                   # util/angles.h:13-34: [-pi, pi)
    two_pi = 2 * np.pi
    return np.fmod(np.fmod(theta + np.pi, two_pi) + two_pi, two_pi) - np.pi


def clamped_v_times_dt(measured_tip, desired_tip, max_speed_times_dt):     # tip_control.cpp:398-411
    e = np.asarray(desired_tip, float) - np.asarray(measured_tip, float)
    n = np.linalg.norm(e)
    return e * max_speed_times_dt / n if n > max_speed_times_dt else e


def _tips(robot, states, device):
    """Tip positions through tip_control's FK wrapper: a retraction beyond L returns (0, 0, L - s)
    (tip_control.cpp:96-104)."""
    out = robot.engine(device).fk_batch(states)
    npts = out["n_points"]
    tips = out["p"][np.arange(len(npts)), npts - 1].copy()
    if robot.enable_retraction:
        over = states[:, -1] > robot.specs.L
        tips[over] = 0.0
        tips[over, 2] = robot.specs.L - states[over, -1]
    return tips


def _eval(robot, p, delta, device):
    """f(p) and the central-difference Jacobian (levmar's: d_j = max(1e-4 |p_j|, delta), misc_core.c:175-211)
    of every row of p in ONE FK batch: (n, 3), (n, 3, S)."""
    n, S = p.shape
    d = np.maximum(np.abs(1e-4 * p), delta)
    batch = np.repeat(p[:, None, :], 2 * S + 1, axis=1)                # (n, 2S+1, S): [p, p - d e_j, p + d e_j ...]
    j = np.arange(S)
    batch[:, 1 + 2 * j, j] -= d
    batch[:, 2 + 2 * j, j] += d
    tips = _tips(robot, batch.reshape(-1, S), device).reshape(n, 2 * S + 1, 3)
    J = (tips[:, 2::2, :] - tips[:, 1::2, :]) * (0.5 / d)[:, :, None]  # (n, S, 3)
    return tips[:, 0, :], np.transpose(J, (0, 2, 1))


def inverse_kinematics_batch(robot, initial_states, des, max_iters=100, mu_init=0.1, stop_threshold_JT_err_inf=1e-9,
                             stop_threshold_Dp=1e-4, stop_threshold_err=1e-4, finite_difference_delta=1e-6, device=0):
    """Solve tip(state) = des from every row of initial_states at once (des: (3,) or one row per start).
    Returns dict(state (n, S), tip (n, 3), error (n,), iters (n,), num_fk_calls (n,), launches)."""
    p = np.ascontiguousarray(np.asarray(initial_states, dtype=np.float64))
    if p.ndim == 1:
        p = p.reshape(1, -1)
    n, S = p.shape
    if S != robot.state_size():
        raise L.InvalidArgument("State is not the right size")
    des = np.broadcast_to(np.asarray(des, dtype=np.float64), (n, 3)).copy()
    b = Bounds.from_robot(robot)
    p = np.clip(p, b.lower, b.upper)
    eps1, eps2_sq, eps3_sq = stop_threshold_JT_err_inf, stop_threshold_Dp ** 2, stop_threshold_err ** 2
    f, J = _eval(robot, p, finite_difference_delta, device)
    fk_calls = np.full(n, 2 * S + 1)
    launches = 1
    e = des - f
    err2 = (e * e).sum(1)
    A = np.einsum("nki,nkj->nij", J, J)
    g = np.einsum("nki,nk->ni", J, e)
    mu = mu_init * np.max(np.diagonal(A, axis1=1, axis2=2), axis=1)
    mu = np.where(mu > 0, mu, mu_init)
    nu = np.full(n, 2.0)
    iters = np.zeros(n, dtype=int)
    eye = np.eye(S)

    def free_gradient(pp, gg):
        """components of J^T e that can still move the state inside the box"""
        blocked = ((pp <= b.lower) & (gg < 0)) | ((pp >= b.upper) & (gg > 0))
        return np.where(blocked, 0.0, gg)

    active = (err2 > eps3_sq) & (np.abs(free_gradient(p, g)).max(1) > eps1)
    while active.any() and (iters[active] < max_iters).any():
        idx = np.flatnonzero(active & (iters < max_iters))
        if idx.size == 0:
            break
        dp = np.linalg.solve(A[idx] + mu[idx, None, None] * eye, g[idx][..., None])[..., 0]
        pn = np.clip(p[idx] + dp, b.lower, b.upper)
        dpe = pn - p[idx]
        small = (dpe * dpe).sum(1) <= eps2_sq * (p[idx] * p[idx]).sum(1)
        iters[idx] += 1
        active[idx[small]] = False
        go = ~small
        if not go.any():
            continue
        k = idx[go]
        fn, Jn = _eval(robot, pn[go], finite_difference_delta, device)
        launches += 1
        fk_calls[k] += 2 * S + 1
        en = des[k] - fn
        err2n = (en * en).sum(1)
        pred = (dpe[go] * (mu[k, None] * dpe[go] + g[k])).sum(1)
        rho = np.where(pred > 0, (err2[k] - err2n) / np.where(pred > 0, pred, 1.0), -1.0)
        acc = rho > 0
        ka = k[acc]
        p[ka], f[ka], J[ka], e[ka], err2[ka] = pn[go][acc], fn[acc], Jn[acc], en[acc], err2n[acc]
        A[ka] = np.einsum("nki,nkj->nij", J[ka], J[ka])
        g[ka] = np.einsum("nki,nk->ni", J[ka], e[ka])
        mu[ka] *= np.maximum(1.0 / 3.0, 1.0 - (2.0 * rho[acc] - 1.0) ** 3)
        nu[ka] = 2.0
        kr = k[~acc]
        mu[kr] *= nu[kr]
        nu[kr] *= 2.0
        active[ka] = (err2[ka] > eps3_sq) & (np.abs(free_gradient(p[ka], g[ka])).max(1) > eps1)
        active[kr] &= np.isfinite(mu[kr]) & (mu[kr] < 1e300)
    state = p.copy()
    if robot.enable_rotation:
        state[:, len(robot.tendons)] = canonical_angle(state[:, len(robot.tendons)])
    return dict(state=state, tip=f, error=np.sqrt(err2), iters=iters, num_fk_calls=fk_calls, launches=launches)


def inverse_kinematics(robot, initial_state, des, max_iters=100, mu_init=0.1, stop_threshold_JT_err_inf=1e-9,
                       stop_threshold_Dp=1e-4, stop_threshold_err=1e-4, finite_difference_delta=1e-6, device=0):
    """tip_control::inverse_kinematics for one start state (tip_control.h:88-100)."""
    r = inverse_kinematics_batch(robot, np.asarray(initial_state, float).reshape(1, -1), des, max_iters, mu_init,
                                 stop_threshold_JT_err_inf, stop_threshold_Dp, stop_threshold_err, finite_difference_delta, device)
    return IKResult(r["state"][0], r["tip"][0], float(r["error"][0]), int(r["iters"][0]), int(r["num_fk_calls"][0]))


def roadmap_ik(robot, goal_tip, vertex_states, vertex_tips, k=10, tolerance=1e-4, **lm):
    """The IK leg of VoxelCachedLazyPRM::roadmapIk (:3164-3205): start from the k roadmap vertices whose tips are
    nearest to the goal, all k solves in the same launches; returns the batch result sorted by error plus the
    indices of the vertices used.  (Validity of the solutions and their connection to the roadmap are the
    caller's: VoxelBackboneValidityChecker.is_valid, VoxelBackboneMotionValidator.check_motion.)"""
    tips = np.asarray(vertex_tips, float)
    d = np.linalg.norm(tips - np.asarray(goal_tip, float), axis=1)
    near = np.argsort(d, kind="stable")[:k]
    lm.setdefault("stop_threshold_err", tolerance)
    r = inverse_kinematics_batch(robot, np.asarray(vertex_states, float)[near], goal_tip, **lm)
    order = np.argsort(r["error"], kind="stable")
    out = {key: (val[order] if isinstance(val, np.ndarray) else val) for key, val in r.items()}
    out["vertices"] = near[order]
    return out
