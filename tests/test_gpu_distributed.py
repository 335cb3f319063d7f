"""The multi-GPU leg on hardware: RCCL (`nccl` backend) process group, the validity bitmask all-gathered as DEVICE
tensors (distributed.allgather_mask -> dist.all_gather_into_tensor), in fresh rank processes -- a one-GPU box can
run world size 1 of it (RCCL refuses two ranks on one device); world size 2 of the sharding logic is the gloo test
in tests/test_host.py, and `bench.py --gpus 2` is rehearsed here with both ranks on cuda:0 (gloo through host
memory for the collective).  Replaces the exchange the reference gets from shared memory
(motion-planning/VoxelCachedLazyPRM.cpp:1446-1483)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import importlib, os, sys
import numpy as np
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
assert dist.get_backend() == "nccl"
irt = importlib.import_module("interactive-rate-tendons_amd")
W, D = irt.workloads, irt.distributed
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
seen = {}
def validate_local(states):
    d = torch.from_numpy(states).cuda()
    bits = torch.zeros((len(states) + 63) // 64, dtype=torch.int64, device="cuda")
    chk.engine.validate_batch_dev(d, len(states), bits)
    torch.cuda.synchronize()
    return bits.cpu().numpy()
M = int(sys.argv[1])
# the collective itself, on device tensors
local = torch.arange(5, dtype=torch.int64, device="cuda") + 100 * dist.get_rank()
full = D.allgather_mask(local)
assert full.is_cuda and full.numel() == 5 * dist.get_world_size() and torch.equal(full[:5].cpu(), torch.arange(5))
mask = irt.unpack_bits(D.ShardedVertexValidator(robot, validate_local, seed=3, device="cuda").run(M), M)
cand = D.candidate_states(robot, 3, 0, M)
mv = irt.VoxelBackboneMotionValidator(chk)
idx = np.flatnonzero(mask)[:261]
a, b = cand[idx[:-1]], cand[idx[1:]]
emask = irt.unpack_bits(D.ShardedEdgeValidator(lambda ea, eb: D.pack_bits(mv.check_motion(ea, eb)), device="cuda").run(a, b), len(a))
# the neighbour phase between them: this rank's rows of the k-nearest table (tr_knn_range), gathered, edges from the table
rb = irt.RoadmapBuilder(chk, mv, seed=1)
verts = cand[mask][:3000]
e_sharded = rb.knn_edges_sharded(verts, 7)
# the edge phase in its roadmap form: this rank's shard of the index pairs through tr_validate_edges_indexed, verdict words gathered
ev_sharded = rb.validate_edges_sharded(verts, e_sharded)
ev_single, _ = rb.validate_edges(verts, e_sharded)
assert np.array_equal(ev_sharded, ev_single) and 0 < ev_single.sum() < len(ev_single)
if dist.get_rank() == 0:
    np.savez(sys.argv[2], mask=mask, emask=emask, direct=chk.is_valid(cand), edirect=mv.check_motion(a, b),
             e_sharded=e_sharded, e_single=rb.knn_edges_gpu(verts, 7))
dist.barrier()
dist.destroy_process_group()
print("nccl-ok")
'''


def _env(**kw):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29650 + os.getpid() % 300), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(kw)
    return env


def test_nccl_world1_allgather_on_device_tensors(tmp_path, orc, irt, helpers):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    out = str(tmp_path / "res.npz")
    M = 5000
    p = subprocess.run([sys.executable, str(script), str(M), out], env=_env(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"),
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "nccl-ok" in p.stdout, p.stderr[-3000:]
    r = np.load(out)
    assert np.array_equal(r["mask"], r["direct"]) and np.array_equal(r["emask"], r["edirect"])
    assert len(r["e_single"]) > 9000 and np.array_equal(r["e_sharded"], r["e_single"])
    robot = irt.workloads.robot_config3()
    vox, _ = irt.workloads.reach_environment(seed=7, n_spheres=64)
    cand = irt.distributed.candidate_states(robot, 3, 0, M)
    want, _, _ = orc.validate_batch(helpers.oracle_robot(orc, robot, lib="omp"), helpers.oracle_grid(orc, vox), cand,
                                    nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(r["mask"], want) and 0 < want.sum() < M


def _bench(args, **env):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=_env(**env), capture_output=True,
                       text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_rccl_path_with_one_rank():
    """bench.py under a launcher's environment with one rank: init_process_group("nccl") + all_gather_into_tensor of
    the device bitmask + the all_reduce of the timing, exactly the statements the N = 2, 4, 8 runs execute."""
    out = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--batch-log2", "15", "--no-cpu-baseline"],
                 RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    assert out["n_gpus"] == 1 and out["config"]["collective"] == "rccl" and out["value"] > 0


def test_bench_gpus2_from_a_bare_command_shared_gpu():
    """`python bench.py --gpus 2` with no launcher: the parent starts the two ranks itself.  Both share cuda:0 here
    (rehearsal mode), so the numbers mean nothing; the control flow is the N > 1 one."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch-log2", "15", "--no-cpu-baseline"], env=dict(env, TENDON_BENCH_SHARED_GPU="1"),
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["config"]["rehearsal_shared_gpu"] is True


def test_config4_workload_world1_rccl_and_world2_rehearsal_agree():
    """`bench.py --workload config4`: the PRM build with vertices, neighbour rows and edges sharded over the ranks.  World
    size 1 runs the RCCL form (device-tensor all-gathers); world size 2 rehearses the N > 1 control flow with both ranks on
    cuda:0 (gloo).  The gathered vertex mask, the vertex count, the edge list and the verdict count must be the same for both
    world sizes ("gathered mask identical for G = 1, 2, 4, 8")."""
    common = ["--workload", "config4", "--config4-log2", "15", "--config4-k", "6", "--steps", "1", "--warmup", "1"]
    one = _bench(["--gpus", "1"] + common, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    assert one["n_gpus"] == 1 and one["config"]["collective"] == "rccl" and one["config"]["ranks_seen"] == 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common,
                       env=dict(env, TENDON_BENCH_SHARED_GPU="1"), capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    two = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    assert two["n_gpus"] == 2 and two["config"]["ranks_seen"] == 2 and two["scaling"] == "strong"
    for key in ("candidates", "valid_vertices", "candidate_edges", "valid_edges", "vertex_mask_crc32", "edge_list_crc32"):
        assert one["config"][key] == two["config"][key], key
    assert 0 < one["config"]["valid_vertices"] < 1 << 15 and 0 < one["config"]["valid_edges"] <= one["config"]["candidate_edges"]
    assert set(two["config"]["collectives_alone"]) == {"vertex_mask", "knn_rows", "edge_mask", "vertex_signatures"}
    assert one["config"]["vertex_signatures_handed_over"] and two["config"]["vertex_signatures_handed_over"]
    assert one["config"]["signature_rows_on_the_wire"] == "packed" and two["config"]["signature_rows_on_the_wire"] == "packed"
    raw = _bench(["--gpus", "1"] + common, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", TENDON_HIP_SIG_WIRE="raw")
    assert raw["config"]["signature_rows_on_the_wire"] == "raw"
    for key in ("valid_vertices", "candidate_edges", "valid_edges", "vertex_mask_crc32", "edge_list_crc32"):
        assert one["config"][key] == raw["config"][key], key
    # ... and without the signature hand-over (every rank's edge call integrates the vertices itself): the same build
    # ... and through the host-array forms of every phase (tr_knn_range / tr_knn_table_edges / tr_validate_edges_indexed): the same build
    plain = _bench(["--gpus", "1"] + common, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", TENDON_BENCH_NO_SIGNATURES="1")
    host = _bench(["--gpus", "1"] + common, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", TENDON_BENCH_HOST_ARRAYS="1")
    assert one["config"]["device_resident_between_phases"] and plain["config"]["device_resident_between_phases"]
    assert not plain["config"]["vertex_signatures_handed_over"] and not host["config"]["device_resident_between_phases"]
    for other in (plain, host):
        for key in ("valid_vertices", "candidate_edges", "valid_edges", "vertex_mask_crc32", "edge_list_crc32"):
            assert one["config"][key] == other["config"][key], key
    assert all(v > 0 for v in two["config"]["phases_ms"].values()) and two["value"] > 0


def test_config4_per_shard_projection_on_one_gpu():
    """`bench.py --workload config4 --emulate-world 2 4`: every rank's shard of every phase run alone on this GPU, its outputs
    compared with the world-1 run's slice inside the workload (it aborts on a difference); the report must carry the per-rank
    times, the replicated work, the all-gather payloads and the projection label."""
    out = _bench(["--workload", "config4", "--config4-log2", "15", "--config4-k", "6", "--steps", "1", "--warmup", "0",
                  "--emulate-world", "2", "4"])
    assert "projection" in out and out["n_gpus"] == 1 and set(out["emulated_worlds"]) == {"2", "4"}
    w1 = out["world_1_ms"]
    assert all(w1[k] > 0 for k in ("vertices", "knn_rows", "edges", "total"))
    for g, w in out["emulated_worlds"].items():
        assert len(w["phases_ms_per_rank"]["edges"]) == int(g) and all(t > 0 for t in w["phases_ms_per_rank"]["edges"])
        assert w["compute_critical_path_ms"] > 0 and 0 <= w["replicated_fraction_of_critical_path"] < 1
        assert set(w["allgather_bytes_per_rank"]) == {"vertex_mask", "vertex_signatures", "knn_rows", "edge_mask"}
    assert 0 < out["valid_vertices"] < 1 << 15 and out["candidate_edges"] > 0
    sw = out["signature_wire"]
    assert sw["raw_row_bytes"] == 576 and sw["packed_row_bytes"] == 104 and sw["pack_all_rows_ms"] > 0 and sw["unpack_all_rows_ms"] > 0
    assert out["emulated_worlds"]["4"]["signature_row_bytes_on_the_wire"] == 104
