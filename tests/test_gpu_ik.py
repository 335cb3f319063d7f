"""Batched inverse kinematics (tip_control.py: every LM iteration of every start state is one FK launch).
The optimiser is this repository's own projected Levenberg-Marquardt, not levmar's code path, so what is
checked is the contract: targets known to be reachable are reached to the requested tolerance inside the
reference's bounds, reported tips are the oracle's FK of the reported states, and the result fields add up."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_tip(orc, helpers, robot, state):
    return helpers.oracle_robot(orc, robot).shape(state)["p"][-1]


@pytest.mark.parametrize("kind", ["config2", "config3", "rot_ret"])
def test_reachable_targets_are_reached(irt, orc, helpers, kind):
    W, T = irt.workloads, irt.tip_control
    robot = {"config2": W.robot_config2, "config3": W.robot_config3, "rot_ret": W.robot_config2}[kind]()
    if kind == "rot_ret":
        robot.enable_rotation = True
        robot.enable_retraction = True
    n = 48
    goal_states = W.random_states(robot, n, seed=11, tau_max=12.0)
    if robot.enable_retraction:
        goal_states[:, -1] = np.random.default_rng(1).uniform(0.0, 0.08, n)
    goals = np.array([_oracle_tip(orc, helpers, robot, s) for s in goal_states])
    rng = np.random.default_rng(12)
    start = goal_states + rng.normal(size=goal_states.shape) * (np.array([1.5] * len(robot.tendons) + ([0.3] if robot.enable_rotation else [])
                                                                         + ([0.01] if robot.enable_retraction else [])))
    b = T.Bounds.from_robot(robot)
    start = np.clip(start, np.maximum(b.lower, -10), np.minimum(b.upper, 25))
    r = T.inverse_kinematics_batch(robot, start, goals, stop_threshold_err=1e-6, stop_threshold_Dp=1e-12, stop_threshold_JT_err_inf=1e-16,
                                   max_iters=60)
    assert (r["error"] <= 1e-6).mean() > 0.9, np.sort(r["error"])[-8:]
    assert (r["state"] >= b.lower - 1e-15).all() and (r["state"] <= b.upper + 1e-15).all()
    if robot.enable_rotation:
        rot = r["state"][:, len(robot.tendons)]
        assert (rot >= -np.pi).all() and (rot < np.pi).all()
    for i in (0, 7, 23, 40):
        tip = _oracle_tip(orc, helpers, robot, r["state"][i])
        assert np.abs(tip - r["tip"][i]).max() <= 1e-9
        assert abs(np.linalg.norm(goals[i] - tip) - r["error"][i]) <= 1e-9
    assert (r["num_fk_calls"] % (2 * robot.state_size() + 1) == 0).all() and (r["iters"] <= 60).all()
    # all starts share launches: far fewer launches than total LM iterations
    assert r["launches"] <= r["iters"].max() + 1 and r["launches"] < r["iters"].sum()


def test_single_start_unreachable_target_and_bounds(irt):
    W, T = irt.workloads, irt.tip_control
    robot = W.robot_config2()
    res = T.inverse_kinematics(robot, [2.0, 2.0, 2.0], [0.5, 0.0, 0.0], max_iters=25)       # 0.5 m away: out of reach
    assert isinstance(res, T.IKResult) and res.iters <= 25 and np.isfinite(res.error) and res.error > 0.25
    b = T.Bounds.from_robot(robot)
    assert (res.state >= b.lower).all() and (res.state <= b.upper).all()
    assert np.array_equal(b.upper, [20.0, 20.0, 20.0]) and np.array_equal(b.lower, [0.0, 0.0, 0.0])
    robot.enable_rotation = True; robot.enable_retraction = True
    b = T.Bounds.from_robot(robot)
    assert b.upper[-1] == robot.specs.L and b.lower[3] == np.finfo(float).min and b.upper[3] == np.finfo(float).max
    assert abs(T.canonical_angle(3 * np.pi + 0.25) - (-np.pi + 0.25)) < 1e-12 and T.canonical_angle(np.pi) == -np.pi
    assert np.allclose(T.clamped_v_times_dt([0, 0, 0], [3, 4, 0], 1.0), [0.6, 0.8, 0]) and np.allclose(T.clamped_v_times_dt([0, 0, 0], [0.3, 0, 0], 1.0), [0.3, 0, 0])
    with pytest.raises(irt.InvalidArgument):
        T.inverse_kinematics(robot, [1.0, 2.0], [0, 0, 0.2])


def test_roadmap_ik_uses_the_nearest_tips(irt):
    W, T = irt.workloads, irt.tip_control
    robot = W.robot_config3()
    states = W.random_states(robot, 2000, seed=21, tau_max=15.0)
    tips = robot.engine(0).fk_batch(states)
    tips = tips["p"][np.arange(len(states)), tips["n_points"] - 1]
    goal = tips[123] + np.array([0.002, -0.001, 0.0015])
    r = T.roadmap_ik(robot, goal, states, tips, k=8, tolerance=1e-5, stop_threshold_Dp=1e-12, stop_threshold_JT_err_inf=1e-16)
    assert len(r["vertices"]) == 8 and (np.diff(r["error"]) >= 0).all() and r["error"][0] <= 1e-5
    d = np.linalg.norm(tips - goal, axis=1)
    assert set(r["vertices"]) == set(np.argsort(d, kind="stable")[:8])
