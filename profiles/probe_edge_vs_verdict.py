"""Same box, same process: fk_verdict<4, SIG> on full batches (tr_validate_candidates_sig_dev, 2^20 candidates) against the edge queue on a
1/8 shard of config 4's edges (tr_validate_edges_indexed_sig_dev): samples/s of both, so that the queue's per-round cost can be told from
the box's clock."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W, D = irt.workloads, irt.distributed
M = 1 << 20
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
eng, mv = chk.engine, irt.VoxelBackboneMotionValidator(chk)
box = D.sampling_box(robot)
k, seed, S, sw = 10, 11, eng.state_size, eng.signature_words()
d_mask = torch.zeros((M + 63) // 64, dtype=torch.int64, device="cuda")
d_sig = torch.empty((M, sw), dtype=torch.int32, device="cuda")
for _ in range(4):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.validate_candidates_sig_dev(seed, 0, M, d_mask, d_sig, box=box)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("fk_verdict<4, SIG> on %d candidates: %.3f ms = %.4g samples/s" % (M, 1e3 * dt, M / dt), flush=True)
cand = torch.empty(M * S, dtype=torch.float64, device="cuda")
eng.candidate_states_dev(seed, 0, M, cand, box=box)
d_v = torch.empty(M * S, dtype=torch.float64, device="cuda")
nv = eng.compact_rows_dev(d_mask, M, cand, S, d_v, M)
d_vsig = torch.empty((M, sw), dtype=torch.int32, device="cuda")
eng.compact_rows_dev(d_mask, M, d_sig.view(torch.float64).reshape(-1), sw // 2, d_vsig.view(torch.float64).reshape(-1), M)
d_v, d_vsig = d_v[: nv * S], d_vsig[:nv].contiguous()
# the accepted vertices alone (valid states, as the edges' samples mostly are)
d_m2 = torch.zeros((nv + 63) // 64, dtype=torch.int64, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng.validate_batch_dev(d_v, nv, d_m2) if hasattr(eng, "validate_batch_dev") else None
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if hasattr(eng, "validate_batch_dev"):
        print("fk_verdict<4> (no rows) on the %d accepted: %.3f ms = %.4g samples/s" % (nv, 1e3 * dt, nv / dt), flush=True)
del d_sig, cand
d_e = torch.empty((nv * (k + 1), 2), dtype=torch.int32, device="cuda")
ne = eng.knn_edges_dev(d_v, nv, k + 1, d_e)
d_e = d_e[:ne].contiguous()
eng.reserve_edges(ne)
for G in (8, 1):
    per = (-(-ne // G) + 63) // 64 * 64
    sh = d_e[:per].contiguous()
    n = len(sh)
    d_bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    d_nfk = torch.zeros(n, dtype=torch.int32, device="cuda")
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.validate_edges_indexed_dev(d_v, nv, sh, n, d_bits, d_nfk, mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change, d_vertex_sig=d_vsig)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        own = int(d_nfk.sum().item()) - 2 * n
        print("edge queue, 1/%d of the edges: %d edges, %d samples: %.3f ms = %.4g samples/s" % (G, n, own, 1e3 * dt, own / dt), flush=True)
