"""The HIP path against the committed golden fixtures (tests/golden/*.npz, produced by
tests/golden/make_golden.py with the strict oracle build): no oracle code runs here."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_golden_config1_fk(irt):
    d = np.load(os.path.join(GOLD, "config1_fk.npz"))
    out = irt.workloads.robot_config1().shape_batch(d["states"])
    assert np.abs(out["p"][:, -1] - d["tips"]).max() <= 1e-9
    assert np.abs(out["p"][:8] - d["p_first8"]).max() <= 1e-9
    assert np.abs(out["L"] - d["L"]).max() <= 1e-10 and np.abs(out["L_i"] - d["L_i"]).max() <= 1e-10
    assert np.array_equal(out["converged"], d["converged"])


def test_golden_config2_validity(irt):
    d = np.load(os.path.join(GOLD, "config2_validity.npz"))
    vox = irt.VoxelOctree.from_sparse(256, (-0.25, 0.25) * 3, d["grid_ids"], d["grid_masks"])
    chk = irt.VoxelBackboneValidityChecker(irt.workloads.robot_config2(), irt.VoxelEnvironment(), vox)
    out = chk.is_valid_detail(d["states"])
    assert np.array_equal(out["valid"], d["valid"])                       # bit-exact verdicts
    assert np.abs(out["tips"] - d["tips"]).max() <= 1e-9
    assert np.array_equal(out["flags"][:512], d["flags512"])


def test_golden_config3_fk(irt):
    d = np.load(os.path.join(GOLD, "config3_fk.npz"))
    robot = irt.workloads.robot_config3()
    out = robot.shape_batch(d["states"])
    assert np.abs(out["p"][:, -1] - d["tips"]).max() <= 1e-9 and np.abs(out["L_i"] - d["L_i"]).max() <= 1e-10
    assert np.array_equal(out["converged"], d["converged"])
    assert np.abs(robot.home_shape().L_i - d["home_L_i"]).max() == 0.0    # same Simpson rule, same order on the host
