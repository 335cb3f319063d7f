"""K4 alone on a working set beyond the Infinity Cache: synthetic CSR of n items x ~51 blocks."""
import importlib, os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=72)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rng = np.random.default_rng(1)
n = 700000
cnt = rng.integers(30, 72, n)
off = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int64)
nnz = int(off[-1])
# consecutive-ish block ids per item like a backbone's path
start = rng.integers(0, 64 ** 3 - 100, n)
ids = (np.repeat(start, cnt) + (np.arange(nnz) - np.repeat(off[:-1], cnt))).astype(np.uint32)
masks = rng.integers(1, 2 ** 62, nnz, dtype=np.uint64)
want = chk.engine.check_cached(ids[: off[20000]], masks[: off[20000]], off[:20001])
g = vox.blocks.reshape(-1)
ref = np.array([bool((g[ids[off[i]:off[i + 1]]] & masks[off[i]:off[i + 1]]).any()) for i in range(20000)])
assert np.array_equal(want, ref), "K4 mismatch"
d_ids = torch.from_numpy(ids.view(np.int32)).cuda(); d_m = torch.from_numpy(masks.view(np.int64)).cuda(); d_o = torch.from_numpy(off).cuda()
bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
for _ in range(3):
    chk.engine.check_cached_dev(d_ids, d_m, d_o, n, bits)
torch.cuda.synchronize()
chk.engine.profile_begin()
for _ in range(20):
    chk.engine.check_cached_dev(d_ids, d_m, d_o, n, bits)
torch.cuda.synchronize()
p = chk.engine.profile_read()["cached_blocks_vs_grid"]
ms = p["total_ms"] / p["launches"]
print("K4: %d items, %.1f MiB, %.4f ms per launch, %.0f GB/s algorithmic (%.3f of 8 TB/s), hit fraction %.3f"
      % (n, 12.0 * nnz / 2 ** 20, ms, (12.0 * nnz + 8.0 * n) / (ms * 1e-3) / 1e9, (12.0 * nnz + 8.0 * n) / (ms * 1e-3) / 8e12, float(ref.mean())))
