"""The edge queue's claim order, A/B in one process: pushed samples first + seeds by depth class (the default) against push order
(TENDON_HIP_EDGE_QUEUE_ORDER=fifo, round 4's schedule) and pushed-first without classes, alternating, on 1/8 shards and the whole
edge list of config 4; every timed call is preceded by a full-size fk_verdict launch so that the clocks are up.  Verdicts and FK counts
are compared."""
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
eng.validate_candidates_sig_dev(seed, 0, M, d_mask, d_sig, box=box)
cand = torch.empty(M * S, dtype=torch.float64, device="cuda")
eng.candidate_states_dev(seed, 0, M, cand, box=box)
d_v = torch.empty(M * S, dtype=torch.float64, device="cuda")
nv = eng.compact_rows_dev(d_mask, M, cand, S, d_v, M)
d_vsig = torch.empty((M, sw), dtype=torch.int32, device="cuda")
eng.compact_rows_dev(d_mask, M, d_sig.view(torch.float64).reshape(-1), sw // 2, d_vsig.view(torch.float64).reshape(-1), M)
d_v, d_vsig = d_v[: nv * S], d_vsig[:nv].contiguous()
d_e = torch.empty((nv * (k + 1), 2), dtype=torch.int32, device="cuda")
ne = eng.knn_edges_dev(d_v, nv, k + 1, d_e)
d_e = d_e[:ne].contiguous()
eng.reserve_edges(ne)
d_m2 = torch.zeros_like(d_mask)
MODES = {"default": {}, "fifo": {"TENDON_HIP_EDGE_QUEUE_ORDER": "fifo"}, "pushed first, no classes": {"TENDON_HIP_EDGE_SEED_CLASSES": "0"}}
for G, ranks in ((8, (0, 3, 7)), (1, (0,))):
    per = (-(-ne // G) + 63) // 64 * 64
    for r in ranks:
        sh = d_e[r * per:min(ne, (r + 1) * per)].contiguous()
        n = len(sh)
        ref = None
        times = {m: [] for m in MODES}
        for rep in range(5):
            for m, env in MODES.items():
                for kk in ("TENDON_HIP_EDGE_QUEUE_ORDER", "TENDON_HIP_EDGE_SEED_CLASSES"):
                    os.environ.pop(kk, None)
                os.environ.update(env)
                d_bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
                d_nfk = torch.zeros(n, dtype=torch.int32, device="cuda")
                eng.validate_candidates_sig_dev(seed, 0, M, d_m2, d_sig, box=box)        # (clocks up; not timed)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                eng.validate_edges_indexed_dev(d_v, nv, sh, n, d_bits, d_nfk, mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change, d_vertex_sig=d_vsig)
                torch.cuda.synchronize(); times[m].append(1e3 * (time.perf_counter() - t0))
                if ref is None:
                    ref = (d_bits.clone(), d_nfk.clone())
                assert torch.equal(ref[0], d_bits) and torch.equal(ref[1], d_nfk), (m, "results differ")
        own = int(ref[1].sum().item()) - 2 * n
        print("1/%d rank %d: %d edges, %d samples: " % (G, r, n, own) + "; ".join("%s %.2f ms (median %.2f)" % (m, min(t), float(np.median(t))) for m, t in times.items()), flush=True)
print("verdicts and FK counts equal in every order")
