"""Config 3's edge call (10^5 valid vertices, 10-NN edges) through the device-resident forms, with and without the vertices' signatures
handed over from the vertex phase (tr_validate_candidates_sig_dev -> tr_validate_edges_indexed_sig_dev): fastest of six calls each."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W, D = irt.workloads, irt.distributed
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
eng, mv = chk.engine, irt.VoxelBackboneMotionValidator(chk)
box = D.sampling_box(robot)
M, k, seed, S, sw = 174720, 10, 11, eng.state_size, eng.signature_words()
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
eng.reserve_edges(ne)
bits = {}
for name, sig in (("integrating the vertices", None), ("signatures handed over", d_vsig), ("integrating the vertices", None), ("signatures handed over", d_vsig)):
    d_bits = torch.zeros((ne + 63) // 64, dtype=torch.int64, device="cuda")
    best = 1e9
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.validate_edges_indexed_dev(d_v, nv, d_e, ne, d_bits, None, mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change, d_vertex_sig=sig)
        torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    bits[name] = d_bits.clone()
    print("%d vertices, %d edges, %s: %.2f ms = %.3g edges/s" % (nv, ne, name, 1e3 * best, ne / best), flush=True)
assert torch.equal(bits["integrating the vertices"], bits["signatures handed over"])
