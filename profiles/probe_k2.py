"""K2 `backbone_voxel_sweep` on stored shapes (tr_validate_shapes_dev): 2^20 backbones of config 2 / 2^19 of config 3, best of seven
launches, the verdict words' checksum (so that two builds can be compared bit for bit)."""
import importlib, os, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
for mk, n, tmax in ((W.robot_config2, 1 << 20, 10.0), (W.robot_config3, 1 << 19, 20.0)):
    robot = mk()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    eng = chk.engine
    P, N = eng.num_points, eng.n_tendons
    st = torch.from_numpy(W.random_states(robot, n, seed=5, tau_max=tmax)).cuda()
    px, py, pz = (torch.empty(P * n, dtype=torch.float64, device="cuda") for _ in range(3))
    Li = torch.empty(N * n, dtype=torch.float64, device="cuda")
    conv = torch.empty(n, dtype=torch.uint8, device="cuda")
    eng.fk_batch_dev(st.reshape(-1), n, n, px, py, pz, d_Li=Li, d_conv=conv)
    bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
    best = 1e9
    for _ in range(7):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.validate_shapes_dev(n, n, px, py, pz, Li, conv, bits)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    b = bits.cpu().numpy()
    print("%d tendons, %d shapes: %.3f ms = %.3g shapes/s, %.0f GB/s algorithmic; valid %d, crc %08x"
          % (N, n, 1e3 * best, n / best, n * (24 * P + 8 * N + 1) / best / 1e9, int(np.unpackbits(b.view(np.uint8)).sum()), zlib.crc32(b.tobytes())), flush=True)
