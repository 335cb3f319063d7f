"""Config 4's edge phase per rank: the 2^20-candidate roadmap's edge list cut into G contiguous shards (as distributed.sharded_edge_verdicts_dev
cuts it), each validated alone through tr_validate_edges_indexed_sig_dev (vertex signatures handed over): edge queue against level-synchronous
lanes, best of four calls per shard, shards 0 and G - 1.  usage: probe_edge_shards.py [G ...] (default 1 8)"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W, D = irt.workloads, irt.distributed
Gs = [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 8]
M = int(os.environ.get("PROBE_CANDIDATES", 1 << 20))
out = {}
for mode in ("0", "1"):
    os.environ["TENDON_HIP_EDGE_QUEUE"] = mode
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
    del d_sig, cand
    d_e = torch.empty((nv * (k + 1), 2), dtype=torch.int32, device="cuda")
    ne = eng.knn_edges_dev(d_v, nv, k + 1, d_e)
    d_e = d_e[:ne].contiguous()
    eng.reserve_edges(ne)
    for G in Gs:
        per = -(-ne // G)
        per = (per + 63) // 64 * 64
        for r in sorted({0, G - 1}):
            lo, hi = min(ne, r * per), min(ne, (r + 1) * per)
            sh = d_e[lo:hi].contiguous()
            d_bits = torch.zeros((hi - lo + 63) // 64, dtype=torch.int64, device="cuda")
            d_nfk = torch.zeros(hi - lo, dtype=torch.int32, device="cuda")
            best = 1e9
            for _ in range(4):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                eng.validate_edges_indexed_dev(d_v, nv, sh, hi - lo, d_bits, d_nfk, mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change, d_vertex_sig=d_vsig)
                torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
            own = int(d_nfk.sum().item()) - 2 * (hi - lo)
            out[(mode, G, r)] = (best, d_bits.clone(), d_nfk.clone())
            print("queue=%s  world %d rank %d: %d edges, %d own samples: %.2f ms = %.3g edges/s, %.3g samples/s" % (mode, G, r, hi - lo, own, 1e3 * best, (hi - lo) / best, own / best), flush=True)
for (mode, G, r), v in out.items():
    if mode == "1":
        o = out[("0", G, r)]
        assert torch.equal(o[1], v[1]) and torch.equal(o[2], v[2]), ("results differ", G, r)
print("verdicts and FK counts equal in both modes")
