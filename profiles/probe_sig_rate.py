"""The edge samples' kernel on full batches (fk_verdict<4, .., SIG> through tr_validate_candidates_sig_dev): seconds per resident round of
131 072 samples on THIS box -- the yardstick for the edge phase's efficiency."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W, D = irt.workloads, irt.distributed
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
eng = chk.engine
box = D.sampling_box(robot)
for M in (1 << 17, 1 << 20, 1 << 21):
    sw = eng.signature_words()
    d_mask = torch.zeros((M + 63) // 64, dtype=torch.int64, device="cuda")
    d_sig = torch.empty((M, sw), dtype=torch.int32, device="cuda")
    eng.validate_candidates_sig_dev(11, 0, M, d_mask, d_sig, box=box)
    best = 1e9
    for _ in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.validate_candidates_sig_dev(11, 0, M, d_mask, d_sig, box=box)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    print("%d candidates with signatures: %.3f ms = %.3g samples/s, %.3f ms per round of 131072" % (M, 1e3 * best, M / best, 1e3 * best / (M / 131072)), flush=True)
