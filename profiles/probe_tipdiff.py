"""Tips of the three schedules of tr_validate_batch against each other, and of one schedule against itself (run-to-run)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
for robot in (W.robot_config2(), W.robot_config3()):
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    states = W.random_states(robot, 20037, seed=91, tau_max=14.0)
    res = {}
    for f in ("2", "1", "0", "0b", "1b"):
        os.environ["TENDON_HIP_FUSED"] = f[0]
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        res[f] = chk.is_valid_detail(states)
        fk = chk.engine.fk_batch(states[:4096]) if hasattr(chk.engine, "fk_batch") else None
        if fk is not None:
            res[f]["p"] = fk["p"]
    for a, b in (("1", "0"), ("2", "0"), ("0", "0b"), ("1", "1b")):
        d = np.abs(res[a]["tips"] - res[b]["tips"])
        print(len(robot.tendons), a, b, "max tip diff", d.max(), "n differing", int((d > 0).any(axis=1).sum()), "valid equal", np.array_equal(res[a]["valid"], res[b]["valid"]), flush=True)
