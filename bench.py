#!/usr/bin/env python3
"""bench.py -- FK+collision checks/sec (3-tendon helical robot, 256^3 voxel environment).

One "step" = one pass of the hot path (fk_verdict: Cosserat-rod FK by RK4 with the whole validity predicate -- length limits,
self collision, backbone against the voxel grid -- in one launch, plus its fallback launch; for N>1 also
the all-gather of the validity bitmask) over one batch of 2^20 synthetic configurations PER GPU
(weak scaling), inputs resident in HBM before the timed region.  Workload = BASELINE.json
configs[1] as specified in BASELINE.md section 2 (seeded, synthetic).

    python bench.py --gpus N --steps K --warmup W
N > 1 works both ways: under an external launcher (python -m torch.distributed.run --nproc-per-node N ...,
which sets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), or as the bare command above -- then this process
starts N fresh rank processes itself, BEFORE it has imported torch or touched the GPU, waits for them and
exits non-zero if any of them failed.

Rank 0 prints ONE JSON line; see README / DESIGN.md for the `roofline` and `cpu_baseline` objects.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6       # 256 CU x 4 SIMD x 16 fp64 FMA lanes/clk x 2 flop x 2.4 GHz
BATCH_LOG2 = 20


def algorithmic_bytes_per_check(S, P, N):
    """SURVEY.md section 8(d), unfused pair: K1 reads 8S, writes 24P + 8(N+1) + 1; K2 reads 24P (+L_i,
    converged), writes 1 bit."""
    k1 = 8 * S + 24 * P + 8 * (N + 1) + 1
    k2 = 24 * P + 8 * N + 1 + 0.125
    return k1, k2


def algorithmic_flops_per_rk4_step(N):
    """Hand count of one RK4 step of the RESTRUCTURED right-hand side (csrc/fk_kernel.hpp: strain_rates_routed + the stage
    updates of fk_uniform_body), every +, -, x one flop, an FMA two, a reciprocal (square root) seed one plus its Newton step
    as written -- independent of what the compiler emits, so that executing more instructions cannot raise the fraction:
      per tendon and evaluation   128   (pd 10, |pd|^2 5, rsqrt + Newton 7, c / q / sdot 7, A 12 + 1, e / g 8, B 18, H 12,
                                         Q / P 10, w 13, pd.w 5, a_i 12, b 8)
      tendon-independent          241   (rhat / rhat^2 folds 5, K u / K (v - e3) 7, c 24, d 12, M11 5, K_bt + H 7, and the
                                         6 x 6 solve as an unrolled L D L^T: 35 FMAs + 15 products + 6 reciprocals (5 each) to
                                         factor, 30 FMAs + 6 products for the two triangular solves = 181; rounds 2 - 3 ran the
                                         two-adjugate Schur form, 272 in all)
      per stage                   78 + 2 N   (p quadrature 21, L_i 2 N, R' = R u^ 27, accumulators 30), + 30 stage state x 3
      per step                    4 (128 N + 241 + 78 + 2 N) + 90 + 3
    = 2 929 (N = 3), 3 449 (N = 4) (rounds 2 - 3: 3 053 / 3 573); what the gfx950 ISA of the same source executes: profiles/isa_counts.json.
    DESIGN.md section 5 carries the derivation; SURVEY 8(d)'s 1.1 Mflop per check is the reference's formulation as written
    (dense 3x3 products, two general inverses), which this kernel does not execute."""
    return 4 * (128 * N + 241 + 78 + 2 * N) + 93


def host_core_share(omp_max):
    """Host cores this process may actually use: the cgroup CPU quota when there is one (a GPU box grants a
    share of the host to each job; running 256 threads on a 16-core quota was 30 % SLOWER than 16), else the
    affinity mask, never more than OpenMP's maximum."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, omp_max))


def cpu_baseline(irt, robot, vox, states, budget_s=12.0):
    """The oracle's OpenMP port (oracle/_build/liboracle_omp.so) on this box's host cores, on a
    bounded sample of the same configurations.  kind = "port": the reference itself cannot be
    built."""
    from oracle import oracle as orc
    s = robot.specs
    orb = orc.Robot([t.C for t in robot.tendons], [t.D for t in robot.tendons], r=robot.r, L=s.L, dL=s.dL,
                    ro=s.ro, ri=s.ri, E=s.E, nu=s.nu, max_tension=[t.max_tension for t in robot.tendons],
                    min_length=[t.min_length for t in robot.tendons], max_length=[t.max_length for t in robot.tendons],
                    residual_threshold=robot.residual_threshold, lib="omp")
    og = orc.Grid(vox.Nx(), vox.limits(), lib="omp")
    og.blocks()[...] = vox.blocks
    threads = host_core_share(orc.max_threads())
    # calibrate on a small slice, then size the sample for ~budget_s of CPU work
    t0 = time.perf_counter()
    orc.validate_batch(orb, og, states[:2048], nthreads=threads, lib=orc.omp_lib())
    orc.validate_batch(orb, og, states[:2048], nthreads=threads, lib=orc.omp_lib())
    rate = 2 * 2048 / (time.perf_counter() - t0)
    m = int(min(len(states), max(4096, rate * budget_s)))
    t0 = time.perf_counter()
    valid, _, used = orc.validate_batch(orb, og, states[:m], nthreads=threads, lib=orc.omp_lib())
    dt = time.perf_counter() - t0
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = None
    return dict(value=m / dt, unit="checks/s", cores=int(used), kind="port", per_core=m / dt / max(1, int(used)), cpu_model=model,
                build="oracle/tendon_oracle.c, gcc -O3 -march=native -fopenmp",
                sample="first %d of the %d configurations of rank 0's batch, %.1f s wall, OpenMP schedule(dynamic,1)"
                       % (m, len(states), dt)), valid, m


def spawn_ranks(n_ranks):
    """`python bench.py --gpus N` without a launcher: start N fresh children (one per GPU) with the
    torch.distributed environment set, relay their exit status.  Nothing in this parent process has initialised
    HIP or imported torch at this point, and the children are new processes (no fork of GPU state, no exec of a
    process that touched the GPU)."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for p in list(pending):
                code = p.poll()
                if code is None:
                    continue
                pending.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    for q in pending:                 # a rank died: the others would wait in a collective for ever
                        q.terminate()
            time.sleep(0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


def load_isa_counts():
    """profiles/isa_counts.json (written by profiles/count_isa.py from the gfx950 assembly of this source tree;
    tests/test_kernel_resources.py fails when it is stale): fp64 flops per RK4 step of the hot kernels."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
    except Exception:
        return {}


def config4_one_gpu_extras():
    """configs[3] at full size (2^20 candidates) through ONE rank's code path of `--workload config4`: what every phase costs
    before it is divided over ranks.  Two builds, the faster reported."""
    try:
        import torch
        irt = importlib.import_module("interactive-rate-tendons_amd")
        W, D = irt.workloads, irt.distributed
        robot = W.robot_config3()
        vox, _ = W.reach_environment(seed=7, n_spheres=64)
        checker = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        eng = checker.engine
        rb = irt.RoadmapBuilder(checker, irt.VoxelBackboneMotionValidator(checker), seed=3)
        M, k, seed = 1 << 20, 10, 3
        box = D.sampling_box(robot)
        vv = D.ShardedVertexValidator(robot, seed=seed, device="cuda", box=box, validate_candidates=D.device_candidate_validator(eng, seed, box))
        best, dev_s, sig_s, same, d_edges, d_bits, d_sig = None, float("inf"), float("inf"), True, None, None, None
        for _ in range(3):
            torch.cuda.synchronize()
            t = [time.perf_counter()]
            mask = vv.run(M, rank=0, world_size=1, keep_on_device=True); torch.cuda.synchronize(); t.append(time.perf_counter())
            verts = D.gather_valid_vertices_dev(eng, seed, M, mask, box=box)[0].cpu().numpy(); t.append(time.perf_counter())
            edges = rb.knn_edges_gpu(verts, k + 1); t.append(time.perf_counter())
            ev, nfk = rb.validate_edges(verts, edges); t.append(time.perf_counter())
            d = np.diff(t)
            if best is None or d.sum() < best[0].sum():
                best = (d, len(verts), len(edges), int(ev.sum()), int(nfk.sum()))
            # the same build with nothing but counts crossing PCIe: the vertices stay where they were compacted, the edge list and
            # the verdict words are device arrays (tr_knn_edges_dev, tr_validate_edges_indexed_dev)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mask = vv.run(M, rank=0, world_size=1, keep_on_device=True)
            d_verts = D.gather_valid_vertices_dev(eng, seed, M, mask, box=box)[0]
            nvd = d_verts.shape[0]
            if d_edges is None or d_edges.shape[0] < nvd * (k + 1):
                d_edges = torch.empty((nvd * (k + 1) * 9 // 8, 2), dtype=torch.int32, device="cuda")
                d_bits = torch.empty((d_edges.shape[0] + 63) // 64, dtype=torch.int64, device="cuda")
            ned = eng.knn_edges_dev(d_verts, nvd, k + 1, d_edges)
            eng.validate_edges_indexed_dev(d_verts, nvd, d_edges, ned, d_bits, None, rb.mv.min_tension_change, rb.mv.min_rotation_change,
                                           rb.mv.min_retraction_change)
            torch.cuda.synchronize()
            dev_s = min(dev_s, time.perf_counter() - t0)
            same = same and ned == len(edges) and bool(np.array_equal(irt.unpack_bits(d_bits.cpu().numpy().view(np.uint64), ned), ev))
            # ... and with the vertices' signatures handed from the vertex phase to the edge call (tr_validate_candidates_sig_dev ->
            # tr_validate_edges_indexed_sig_dev): the accepted vertices are not integrated a second time
            sw = eng.signature_words()
            if sw:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                if d_sig is None:
                    d_sig = torch.empty((M, sw), dtype=torch.int32, device="cuda")
                    d_vsig = torch.empty((M, sw), dtype=torch.int32, device="cuda")
                    d_mask2 = torch.empty((M + 63) // 64, dtype=torch.int64, device="cuda")
                    d_cand = torch.empty(M * eng.state_size, dtype=torch.float64, device="cuda")
                    d_vs = torch.empty(M * eng.state_size, dtype=torch.float64, device="cuda")
                eng.validate_candidates_sig_dev(seed, 0, M, d_mask2, d_sig, box=box)
                eng.candidate_states_dev(seed, 0, M, d_cand, box=box)
                nv2 = eng.compact_rows_dev(d_mask2, M, d_cand, eng.state_size, d_vs, M)
                eng.compact_rows_dev(d_mask2, M, d_sig.view(torch.float64).reshape(-1), sw // 2, d_vsig.view(torch.float64).reshape(-1), M)
                ne2 = eng.knn_edges_dev(d_vs, nv2, k + 1, d_edges)
                eng.validate_edges_indexed_dev(d_vs, nv2, d_edges, ne2, d_bits, None, rb.mv.min_tension_change, rb.mv.min_rotation_change,
                                               rb.mv.min_retraction_change, d_vertex_sig=d_vsig)
                torch.cuda.synchronize()
                sig_s = min(sig_s, time.perf_counter() - t0)
                same = same and ne2 == len(edges) and bool(np.array_equal(irt.unpack_bits(d_bits.cpu().numpy().view(np.uint64), ne2), ev))
        d, nv, ne, nok, nfk = best
        return {"candidates": M, "valid_vertices": nv, "candidate_edges": ne, "valid_edges": nok, "build_s": float(d.sum()),
                "build_device_resident_s": float(dev_s), "build_device_resident_signatures_handed_over_s": float(sig_s) if sig_s < float("inf") else None,
                "device_resident_verdicts_equal": same,
                "vertex_phase_checks_per_s": M / d[0], "valid_vertices_per_s": nv / d[0], "compact_and_download_s": float(d[1]),
                "knn_edge_list_s": float(d[2]), "edges_validated_per_s": ne / d[3], "edge_fk_samples_per_s": nfk / d[3]}
    except Exception as e:                                  # noqa: BLE001 -- reported, not raised
        return {"error": repr(e)[:300]}


def secondary_metrics():
    """BASELINE configs 1, 3 and 5 at full size on this GPU (bench_roadmap.py, without its CPU leg), condensed; never part of
    `value`, and a failure here must not cost the headline line."""
    try:
        import bench_roadmap
        r = bench_roadmap.run(["--no-cpu"])
        c3, c5, q = r["config3"], r["config5"], r["config5"]["queries"]
        return {
            "config3_100k_vertices_k10": {
                "valid_vertices_per_s": c3["valid_vertices_per_s"], "knn_edge_list_s": c3["knn_gpu_seconds_incl_pcie_and_dedup"],
                "edges": c3["edges"], "edges_validated_per_s": c3["edges_per_s"], "edges_validated_per_s_repeat_call": c3["edges_per_s_repeat"],
                "edges_validated_per_s_vertex_signatures_handed_over": c3["edges_per_s_device_resident_signatures_handed_over"], "edge_fk_samples_per_s": c3["edge_fk_samples_per_s"],
                "fk_samples_per_edge_mean": c3["fk_samples_per_edge"]["mean"], "connect_all_edges_s": q["connect_all_edges_s"],
                "create_roadmap_s": c3["create_roadmap"]["seconds"],
                "edge_kernel_roofline": c3.get("edge_kernel_roofline")},
            "config5_10k_queries": {
                "queries_per_s": q["default_schedule"]["queries_per_s"], "rounds": q["default_schedule"]["rounds"],
                "queries_per_s_lazy": q["lazy"]["queries_per_s"], "lazy_rounds": q["lazy"]["rounds"], "lazy_items_checked": q["lazy"]["items_checked"],
                "queries_per_s_lazy_host_threads_only": q["searches_on_host_threads_only"]["lazy_queries_per_s"], "lazy_searches": q["lazy"]["searches"],
                "queries_per_s_eager_incl_revalidation": q["eager"]["queries_per_s_incl_revalidation"],
                "revalidate_all_cached_sets_ms": q["eager"]["revalidate_all_ms"], "cached_sets": q["roadmap_vertices"] + q["roadmap_edges"],
                "vertex_caches_built_per_s": c5["vertex_caches_built_per_s"], "edge_caches_built_per_s": c5["edge_caches_built_per_s"],
                "rooflines": q.get("rooflines"), "solved_fraction": q["solved_fraction"],
                "with_64_landmark_tables": q.get("landmarks_64")},
            "config4_on_one_gpu": config4_one_gpu_extras(),
            "config1_fk_only": r["config1"], "sphere_checker_checks_per_s": r["sphere_checker"]["checks_per_s"],
            "rotation_retraction_robot": {k: r["rotation_retraction_robot"][k] for k in ("robot", "checks_per_s", "edges", "edges_per_s",
                                                                                         "edge_fk_samples_per_s")},
            "source": "bench_roadmap.py (python bench.py --workload roadmap prints all of it)"}
    except Exception as e:                                  # noqa: BLE001 -- reported, not raised
        return {"error": repr(e)[:400]}


def setup_ranks(args):
    """One process per GPU: device selection and the process group (RCCL; gloo through host memory in rehearsal mode)."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but the launcher set WORLD_SIZE=%d" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    # TENDON_BENCH_SHARED_GPU=1 is a REHEARSAL mode for boxes with one GPU: every rank uses cuda:0 and the
    # mask is gathered with gloo through host memory, so the N>1 control flow can be exercised (the numbers
    # mean nothing).  The real N>1 path below is one rank per GPU with RCCL.
    rehearsal = os.environ.get("TENDON_BENCH_SHARED_GPU") == "1"
    dev_index = 0 if rehearsal else local_rank
    if dev_index >= torch.cuda.device_count():
        raise SystemExit("rank %d needs cuda:%d but this node has %d GPU(s) (TENDON_BENCH_SHARED_GPU=1 rehearses N>1 on one GPU)"
                         % (rank, dev_index, torch.cuda.device_count()))
    torch.cuda.set_device(dev_index)
    # a launcher's environment (even with one rank) selects the distributed path: process group + all-gather
    use_dist = world > 1 or ("WORLD_SIZE" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    return torch, dist, world, rank, local_rank, dev_index, rehearsal, use_dist


def run_config4(args, torch, dist, world, rank, dev_index, rehearsal, use_dist):
    """BASELINE configs[3]: the PRM build at 2^20 candidate vertices with every phase sharded over the ranks (strong scaling:
    the job is fixed, ranks divide it) --
      1. vertex phase: rank g validates candidates [g M/G, (g+1) M/G) of the counter-based sequence, generated in HBM
         (tr_validate_candidates_sig_dev), and the mask words -- and the accepted candidates' signature rows, which the edge phase
         would otherwise recompute for ALL vertices on every rank -- are all-gathered as device tensors (RCCL);
      2. every rank regenerates the M candidates and compacts them by the gathered mask (tr_candidate_states_dev,
         tr_compact_rows_dev): the same vertex array everywhere, no states on the wire;
      3. connection loop: the rank's rows of the exact k-nearest table against all vertices (tr_knn_range), all-gather of the
         int32 rows, the deduplicated edge list from the whole table on every rank (tr_knn_table_edges);
      4. edge phase: the rank's contiguous shard of the edge list through tr_validate_edges_indexed_sig_dev, all-gather of the
         verdict words.
    One step = the whole pipeline.  `value` = candidates through the vertex phase per second of vertex-phase time (the
    metric's FK+collision checks, whole job); the other phases are reported beside it, each as the max over ranks, with the
    collectives timed on their own afterwards."""
    import zlib
    irt = importlib.import_module("interactive-rate-tendons_amd")
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    checker = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox, device=dev_index)
    mv = irt.VoxelBackboneMotionValidator(checker)
    eng = checker.engine
    rb = irt.RoadmapBuilder(checker, mv, seed=3)
    M, k, seed = 1 << args.config4_log2, args.config4_k, 3
    dev = "cuda:%d" % dev_index
    box = D.sampling_box(robot)
    # Between the phases everything stays in HBM (vertices, neighbour rows, edge list, verdict words: tr_knn_range_dev,
    # tr_knn_table_edges_dev, tr_validate_edges_indexed_dev); TENDON_BENCH_HOST_ARRAYS=1: the host-array forms of round 2.
    # The accepted candidates' backbone signatures travel with the mask (one more all-gather), so that no rank integrates the whole
    # vertex set again for its shard of the edges; TENDON_BENCH_NO_SIGNATURES=1: without (every rank's edge call does its vertex pass)
    resident = not os.environ.get("TENDON_BENCH_HOST_ARRAYS")
    hand_over = resident and eng.signature_words() > 0 and not os.environ.get("TENDON_BENCH_NO_SIGNATURES")
    vv = D.ShardedVertexValidator(robot, seed=seed, device=dev, box=box,
                                  validate_candidates=D.device_candidate_validator(eng, seed, box, signatures=hand_over))
    compact = D.device_row_compactor(eng)
    codec = D.signature_wire_codec(eng) if hand_over else None    # the rows travel delta-coded (TENDON_HIP_SIG_WIRE=raw: as they are)
    space = (mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def pipeline():
        t = [time.perf_counter()]
        vsig = None
        if hand_over:
            mask, vsig = vv.run_with_rows(M, compact, codec=codec)
        else:
            mask = vv.run(M, keep_on_device=True)
        torch.cuda.synchronize(); t.append(time.perf_counter())
        verts_dev, _ = D.gather_valid_vertices_dev(eng, seed, M, mask, box=box)
        if resident:
            torch.cuda.synchronize(); t.append(time.perf_counter())
            d_edges = torch.empty((max(1, verts_dev.shape[0] * (k + 1)), 2), dtype=torch.int32, device=dev)
            ne = D.sharded_knn_edges_dev(eng, verts_dev, k + 1, d_edges)         # k neighbours + the vertex itself, as nearestK returns
            t.append(time.perf_counter())
            words = D.sharded_edge_verdicts_dev(eng, verts_dev, d_edges, ne, space, d_vertex_sig=vsig)
            torch.cuda.synchronize(); t.append(time.perf_counter())
            return mask, verts_dev, d_edges[:ne], words, np.diff(t)
        verts = verts_dev.cpu().numpy()
        t.append(time.perf_counter())
        edges = rb.knn_edges_sharded(verts, k + 1, device=dev)                  # k neighbours + the vertex itself, as nearestK returns
        t.append(time.perf_counter())
        ev = rb.validate_edges_sharded(verts, edges, device=dev)
        t.append(time.perf_counter())
        return mask, verts, edges, ev, np.diff(t)

    for _ in range(args.warmup):
        pipeline()
    fence()
    eng.profile_begin()
    t0 = time.perf_counter()
    phase = np.zeros(4)
    for _ in range(args.steps):
        mask, verts, edges, ev, dt = pipeline()
        phase += dt
    fence()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_end()
    phase /= args.steps
    if resident:                        # (to the host for the checksums only, outside the timed region)
        verts, edges = verts.cpu().numpy(), edges.cpu().numpy()
        ev = irt.unpack_bits(ev.cpu().numpy().view(np.uint64), len(edges))
    mask_words = mask.cpu().numpy().view(np.uint64)
    sums = np.array([zlib.crc32(mask_words[: (M + 63) // 64].tobytes()), len(verts), len(edges), int(ev.sum()), zlib.crc32(edges.tobytes())],
                    dtype=np.int64)
    coll = {}
    if use_dist:
        cdev = "cpu" if rehearsal else dev
        tt = torch.tensor(np.concatenate([[elapsed], phase]), dtype=torch.float64, device=cdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, phase = float(tt[0]), tt[1:].cpu().numpy()
        lo, hi = torch.tensor(sums, device=cdev), torch.tensor(sums, device=cdev)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        if not torch.equal(lo, hi):
            raise SystemExit("rank %d: the ranks disagree on the gathered mask / vertices / edge list / verdicts" % rank)
        # the three collectives on their own, same buffers and sizes
        _, _, vshard = D.shard_bounds(M, world, rank)
        _, _, nshard = D.shard_bounds(len(verts), world, rank)
        _, _, eshard = D.shard_bounds(len(edges), world, rank)
        bufs = {"vertex_mask": torch.zeros(vshard // 64, dtype=torch.int64, device=dev),
                "knn_rows": torch.zeros(nshard * (k + 1), dtype=torch.int32, device=dev),
                "edge_mask": torch.zeros(eshard // 64, dtype=torch.int64, device=dev)}
        if hand_over:                   # (padded to the largest shard's accepted count; about the same on every rank)
            bufs["vertex_signatures"] = torch.zeros((len(verts) + world - 1) // world * (eng.signature_packed_words() if codec is not None else eng.signature_words()),
                                                    dtype=torch.int32, device=dev)
        for name, b in bufs.items():
            D.allgather_mask(b)
            fence()
            t1 = time.perf_counter()
            for _ in range(10):
                D.allgather_mask(b)
            fence()
            tg = torch.tensor([(time.perf_counter() - t1) / 10], dtype=torch.float64, device=cdev)
            dist.all_reduce(tg, op=dist.ReduceOp.MAX)
            coll[name] = {"ms": 1e3 * float(tg.item()), "bytes_per_rank": b.numel() * b.element_size()}
    if rank == 0:
        out = {
            "metric": "FK+collision checks/sec (4-tendon PRM vertex phase, 256^3 voxel env)",
            "value": M / phase[0], "unit": "checks/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "configs[3]: PRM roadmap, 2^%d candidate vertices (4-tendon quadratic-routed robot, tau~U[0,20)^4, 256^3 grid, "
                                   "64 spheres) sharded over the ranks: vertex mask all-gather -> %d-NN rows all-gather -> edge verdicts "
                                   "all-gather" % (args.config4_log2, k),
                       "ranks_seen": dist.get_world_size() if use_dist else 1, "rehearsal_shared_gpu": rehearsal,
                       "collective": ("gloo(host)" if rehearsal else "rccl") if use_dist else None,
                       "device_resident_between_phases": bool(resident), "vertex_signatures_handed_over": bool(hand_over),
                       "signature_rows_on_the_wire": (getattr(vv, "rows_on_the_wire", "raw") if hand_over else None),
                       "candidates": M, "valid_vertices": int(len(verts)), "candidate_edges": int(len(edges)), "valid_edges": int(ev.sum()),
                       "vertex_mask_crc32": int(sums[0]), "edge_list_crc32": int(sums[4]),
                       "phases_ms": {"vertices_incl_allgather": 1e3 * phase[0], "regenerate_compact_download": 1e3 * phase[1],
                                     "knn_rows_incl_allgather_and_edge_list": 1e3 * phase[2], "edges_incl_allgather": 1e3 * phase[3]},
                       "collectives_alone": coll,
                       "rates": {"valid_vertices_per_s": len(verts) / phase[0], "edges_validated_per_s": len(edges) / phase[3]}},
            "kernels": {name: {"launches": v["launches"], "total_ms": v["total_ms"]} for name, v in prof.items()},
        }
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    return 0

def run_config4_projection(args, torch, dev_index):
    """`--workload config4 --emulate-world G [G ...]`: the per-rank critical path of the sharded PRM build at world sizes this box
    cannot run, measured on ONE GPU -- a PROJECTION, not a scaling curve.  The build runs once at world size 1; then, for every world
    size G and every rank r = 0 .. G-1 in turn, rank r's shard of each phase runs alone on the GPU with the other ranks' gathered
    inputs taken from the world-1 run (the candidate sequence, the gathered mask, the vertices' signature rows, the neighbour table and
    the edge list are the same for every world size), its outputs compared with the world-1 run's slice.  Reported per G: each
    phase's time per rank and its maximum over the ranks, the work every rank repeats (regenerating and compacting the candidates,
    deriving the edge list from the gathered table), the bytes each rank contributes to each all-gather, and two labelled MODELS of
    the collectives' time (xGMI: 7 links per GPU; all links at once, and a ring).  What it cannot show: RCCL's real latency, link
    contention, host jitter across 8 processes."""
    irt = importlib.import_module("interactive-rate-tendons_amd")
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    checker = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox, device=dev_index)
    mv = irt.VoxelBackboneMotionValidator(checker)
    eng = checker.engine
    M, k, seed = 1 << args.config4_log2, args.config4_k, 3
    dev = "cuda:%d" % dev_index
    box = D.sampling_box(robot)
    sw, S = eng.signature_words(), eng.state_size
    if sw == 0:
        raise SystemExit("--emulate-world needs a context that hands vertex signatures over (backbone checker, no retraction)")
    validate = D.device_candidate_validator(eng, seed, box, signatures=True)
    compact = D.device_row_compactor(eng)
    space = (mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change)
    reps = max(1, args.steps)

    def timed(fn):
        """fastest of `reps` runs of fn, in ms, and its last result"""
        best, out = 1e30, None
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = fn()
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        return 1e3 * best, out

    def vertex_shard(G, r):
        start, stop, shard = D.shard_bounds(M, G, r)
        n_real = max(0, min(stop, M) - start)
        local, rows = validate(start, n_real, shard // 64)
        return local, compact(local, n_real, rows), shard

    def knn_shard(verts, G, r):
        n = verts.shape[0]
        start, stop, shard = D.shard_bounds(n, G, r)
        n_real = max(0, min(stop, n) - start)
        rows = torch.full((shard, k + 1), -1, dtype=torch.int32, device=dev)
        if n_real > 0:
            eng.knn_range_dev(verts, n, start, n_real, k + 1, rows)
        return rows, shard

    def edge_shard(verts, vsig, d_edges, ne, G, r):
        start, stop, shard = D.shard_bounds(ne, G, r)
        n_real = max(0, min(stop, ne) - start)
        words = torch.zeros(shard // 64, dtype=torch.int64, device=dev)
        if n_real > 0:
            eng.validate_edges_indexed_dev(verts, verts.shape[0], d_edges[start:start + n_real], n_real, words, None, *space, d_vertex_sig=vsig)
            if n_real % 64:
                words[n_real // 64] &= (1 << (n_real % 64)) - 1
        return words, shard

    # ---- world size 1: the reference results and times ----
    for _ in range(max(1, args.warmup)):
        vertex_shard(1, 0)
    t_v1, (mask, vsig, _) = timed(lambda: vertex_shard(1, 0))
    t_regen, (verts, _) = timed(lambda: D.gather_valid_vertices_dev(eng, seed, M, mask, box=box))
    nv = verts.shape[0]
    t_k1, (table, _) = timed(lambda: knn_shard(verts, 1, 0))
    table = table[:nv].contiguous()
    d_edges = torch.empty((max(1, nv * (k + 1)), 2), dtype=torch.int32, device=dev)
    t_list, ne = timed(lambda: eng.edges_from_knn_dev(table.reshape(-1), nv, k + 1, d_edges))
    eng.reserve_edges(ne)
    edge_shard(verts, vsig, d_edges, ne, 1, 0)
    t_e1, (ewords, _) = timed(lambda: edge_shard(verts, vsig, d_edges, ne, 1, 0))
    codec = D.signature_wire_codec(eng)
    t_unpack = 0.0
    if codec is not None:
        packed, bad = codec.pack(vsig)
        assert bad == 0, "%d signature rows of valid vertices could not be delta-coded" % bad
        back = codec.unpack(packed)
        P_ = eng.num_points
        assert torch.equal(back[:, :P_], vsig[:, :P_]), "signature rows differ after pack + unpack"
        t_pack1 = timed(lambda: codec.pack(vsig))[0]
        t_unpack = timed(lambda: codec.unpack(packed))[0]
    base = {"vertices": t_v1, "regenerate_and_compact": t_regen, "knn_rows": t_k1, "edge_list_from_table": t_list, "edges": t_e1}
    world1 = sum(base.values())
    worlds = {}
    link_gbs = 64.0                     # one direction of one xGMI link (MI355X_MICROARCH.md: 7 links x ~153 GB/s bidirectional per GPU)
    for G in args.emulate_world:
        per = {"vertices": [], "knn_rows": [], "edges": []}
        counts = []
        for r in range(G):
            t, (local, mine, vshard) = timed(lambda: vertex_shard(G, r))
            per["vertices"].append(t)
            lo = r * (vshard // 64)
            ref = mask[lo:lo + vshard // 64]
            assert torch.equal(local[: ref.numel()], ref) and not bool(local[ref.numel():].any()), "rank %d of %d: vertex mask differs from the world-1 run" % (r, G)
            counts.append(int(mine.shape[0]))
            t, (rows, nshard) = timed(lambda: knn_shard(verts, G, r))
            per["knn_rows"].append(t)
            lo = r * nshard
            n_real = max(0, min(lo + nshard, nv) - lo)
            assert torch.equal(rows[:n_real], table[lo:lo + n_real]), "rank %d of %d: neighbour rows differ from the world-1 run" % (r, G)
            t, (w, eshard) = timed(lambda: edge_shard(verts, vsig, d_edges, ne, G, r))
            per["edges"].append(t)
            lo = r * (eshard // 64)
            ref = ewords[lo:lo + eshard // 64]
            assert torch.equal(w[: ref.numel()], ref), "rank %d of %d: edge verdicts differ from the world-1 run" % (r, G)
        assert sum(counts) == nv
        # the signature rows travel delta-coded (tr_pack_signatures_dev on the rank's rows before the all-gather, tr_unpack_signatures_dev
        # on all rows after it: both on every rank's critical path); TENDON_HIP_SIG_WIRE=raw: as they are
        row_bytes = (eng.signature_packed_words() if codec is not None else sw) * 4
        payload = {"vertex_mask": vshard // 8, "vertex_signatures": max(counts) * row_bytes, "knn_rows": nshard * (k + 1) * 4, "edge_mask": eshard // 8}
        # a rank receives (G - 1) shards: over all its links at once, or one after the other round a ring; 20 us per step of latency
        model_links = {n: 1e3 * (b / (link_gbs * 1e9) + 20e-6) for n, b in payload.items()}
        model_ring = {n: 1e3 * (G - 1) * (b / (link_gbs * 1e9) + 20e-6) for n, b in payload.items()}
        phases = {n: max(v) for n, v in per.items()}
        replicated = {"regenerate_and_compact": t_regen, "edge_list_from_table": t_list}
        if codec is not None:
            phases["signature_pack"] = timed(lambda: codec.pack(vsig[: max(counts)]))[0]
            replicated["signature_unpack"] = t_unpack
        crit = sum(phases.values()) + sum(replicated.values())
        worlds[str(G)] = {
            "phases_ms_max_over_ranks": phases, "phases_ms_per_rank": per, "replicated_ms_on_every_rank": replicated,
            "compute_critical_path_ms": crit, "replicated_fraction_of_critical_path": sum(replicated.values()) / crit,
            "allgather_bytes_per_rank": payload, "allgather_model_ms_all_links": model_links, "allgather_model_ms_ring": model_ring,
            "projected_build_ms": {"all_links": crit + sum(model_links.values()), "ring": crit + sum(model_ring.values())},
            "projected_speedup_over_world_1": {"compute_only": world1 / crit, "all_links": world1 / (crit + sum(model_links.values())),
                                               "ring": world1 / (crit + sum(model_ring.values()))},
            "shard_efficiency": {n: base[n] / (G * phases[n]) for n in phases if n in base},
            "signature_row_bytes_on_the_wire": row_bytes,
        }
    out = {"projection": True,
           "what": "per-rank critical path of BASELINE configs[3] (2^%d candidates, %d neighbours) at world sizes emulated on ONE GPU: rank r's shard "
                   "of each phase run alone, in turn, with the gathered inputs of a world-1 run; collectives are MODELLED, not run; this is "
                   "not a scaling curve" % (args.config4_log2, k),
           "n_gpus": 1, "candidates": M, "valid_vertices": int(nv), "candidate_edges": int(ne), "repetitions_fastest_of": reps,
           "edge_schedule": eng.edge_schedule_last(), "world_1_ms": dict(base, total=world1), "emulated_worlds": worlds,
           "signature_wire": None if codec is None else {"raw_row_bytes": sw * 4, "packed_row_bytes": eng.signature_packed_words() * 4,
                                                         "pack_all_rows_ms": t_pack1, "unpack_all_rows_ms": t_unpack, "rows": int(nv)},
           "collective_model": "bytes_per_rank / %.0f GB/s + 20 us, once (all 7 links at once) or G - 1 times (ring)" % link_gbs}
    print(json.dumps(out), flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-log2", type=int, default=BATCH_LOG2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary metrics (configs 1, 3, 5 at full size; ~15 s on the GPU)")
    ap.add_argument("--config4-log2", type=int, default=20, help="--workload config4: log2 of the candidate vertices (whole job)")
    ap.add_argument("--config4-k", type=int, default=10, help="--workload config4: neighbours per vertex")
    ap.add_argument("--workload", default="config2", choices=["config2", "roadmap", "config4"],
                    help="config2 (default): the headline line.  roadmap: BASELINE configs 3 and 5 at full size on one GPU "
                         "(bench_roadmap.py: 100k-vertex PRM, k-NN edges, edge validation with the FK-samples/edge histogram, voxel "
                         "caches, 10k lazy queries) -- prints that script's JSON object instead of the headline line.  config4: "
                         "BASELINE configs[3], the PRM build at 2^20 candidates with vertices, neighbour rows and edges sharded over "
                         "--gpus N ranks (strong scaling), one JSON line with per-phase and per-collective times")
    ap.add_argument("--emulate-world", type=int, nargs="+", default=None, metavar="G",
                    help="--workload config4 on ONE GPU: run every rank's shard of every phase alone, in turn, for each world size G given, and "
                         "print a PROJECTION of the per-rank critical path (max over ranks per phase, replicated work, all-gather payloads; the "
                         "collectives are modelled).  Not a scaling curve")
    ap.add_argument("--traffic-json", default=os.path.join(ROOT, "profiles", "traffic_latest.json"),
                    help="per-launch HBM bytes of the dominant kernel from a separate rocprofv3 --pmc pass")
    args = ap.parse_args()

    if args.workload == "roadmap":
        import bench_roadmap
        return bench_roadmap.main([] if not args.no_cpu_baseline else ["--no-cpu"])
    if args.workload == "config4" and args.steps == 20 and args.warmup == 3:
        args.steps, args.warmup = 3, 1                      # a step is the whole roadmap build
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # build once here: N ranks compiling into the same object directory at once would race (hipcc only; no GPU call)
        importlib.import_module("interactive-rate-tendons_amd._lib").build()
        sys.exit(spawn_ranks(args.gpus))

    torch, dist, world, rank, local_rank, dev_index, rehearsal, use_dist = setup_ranks(args)
    if args.workload == "config4" and args.emulate_world:
        if world != 1:
            raise SystemExit("--emulate-world runs on one GPU (--gpus 1)")
        return run_config4_projection(args, torch, dev_index)
    if args.workload == "config4":
        return run_config4(args, torch, dist, world, rank, dev_index, rehearsal, use_dist)

    irt = importlib.import_module("interactive-rate-tendons_amd")
    W = irt.workloads
    robot = W.robot_config2()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    checker = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox, device=dev_index)
    eng = checker.engine
    n = 1 << args.batch_log2
    S, P, N = eng.state_size, eng.num_points, eng.n_tendons
    # every rank validates its own shard of the global candidate sequence (weak scaling)
    # ... generated in HBM by the library's counter-based generator (tr_candidate_states_dev; nothing is uploaded); the host
    # mirror of the same sequence feeds the CPU baseline and the verdict comparison below
    box = irt.distributed.sampling_box(robot, tau_max=10.0)
    d_states = torch.empty(n * S, dtype=torch.float64, device="cuda")
    eng.candidate_states_dev(2024, rank * n, n, d_states, box=box)
    states = irt.distributed.candidate_states(robot, seed=2024, start=rank * n, count=n, tau_max=10.0)
    if not np.array_equal(d_states[: 4096 * S].cpu().numpy().reshape(4096, S), states[:4096]):
        raise SystemExit("rank %d: device candidate generator differs from its host mirror" % rank)
    d_bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
    d_tips = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    d_all = torch.zeros(world * (n // 64), dtype=torch.int64, device="cuda") if use_dist else None
    eng.reserve(n)

    h_all = torch.zeros(world * (n // 64), dtype=torch.int64) if (use_dist and rehearsal) else None

    def step():
        eng.validate_batch_dev(d_states, n, d_bits, d_tips)
        if use_dist:
            if rehearsal:
                dist.all_gather_into_tensor(h_all, d_bits.cpu())
            else:
                dist.all_gather_into_tensor(d_all, d_bits)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    eng.profile_begin()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    prof = eng.profile_read()
    eng.profile_end()
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # every rank's shard of the gathered mask must be that rank's own verdicts
        gathered = (h_all if rehearsal else d_all.cpu()).numpy().view(np.uint64)
        mine = d_bits.cpu().numpy().view(np.uint64)
        if not np.array_equal(gathered[rank * (n // 64):(rank + 1) * (n // 64)], mine):
            raise SystemExit("rank %d: gathered validity mask does not contain this rank's shard" % rank)
        # the collective on its own ("report 1/2/4/8 with the all-gather time broken out"): the same K all-gathers
        # of the same buffers without the kernels, after the timed region
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            if rehearsal:
                dist.all_gather_into_tensor(h_all, d_bits.cpu())
            else:
                dist.all_gather_into_tensor(d_all, d_bits)
        fence()
        tg = torch.tensor([(time.perf_counter() - t1) / args.steps], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        allgather_ms = 1e3 * float(tg.item())

    valid_bits = d_bits.cpu().numpy().view(np.uint64)
    valid = irt.unpack_bits(valid_bits, n)
    checks = world * n * args.steps

    if rank == 0:
        k1b, k2b = algorithmic_bytes_per_check(S, P, N)
        kvb = 8 * S + 24 + 0.125                   # fk_verdict: state in, tip + one verdict bit out (fused verdict-only variant)
        k1 = prof["fk_rk4_batch"]
        k2 = prof["backbone_voxel_sweep"]
        kf = prof.get("fk_sweep_fused", {"launches": 0, "total_ms": 0.0})
        kv = prof.get("fk_verdict", {"launches": 0, "total_ms": 0.0})
        k1_ms = k1["total_ms"] / max(1, k1["launches"])
        k2_ms = k2["total_ms"] / max(1, k2["launches"])
        kf_ms = kf["total_ms"] / max(1, kf["launches"])
        kv_ms = kv["total_ms"] / max(1, kv["launches"])
        # dominant = most device time in the timed region.  tr_validate_batch* normally runs fk_verdict (the whole
        # predicate in one pass, no point storage) followed by fk_sweep_fused_list on the few configurations that need
        # the exact self-collision sweep (slot fk_sweep_fused); TENDON_HIP_FUSED=1 runs K1 + K2 as one kernel over
        # stored points (fk_sweep_fused: K1's writes plus K2's reads of the same points), =0 launches them separately.
        cands = [("fk_verdict", kv, kv_ms, kvb), ("fk_sweep_fused", kf, kf_ms, k1b + k2b), ("fk_rk4_batch", k1, k1_ms, k1b),
                 ("backbone_voxel_sweep", k2, k2_ms, k2b)]
        dom_name, dom, dom_ms, dom_bytes = max(cands, key=lambda c: c[1]["total_ms"])
        units_per_launch = n * args.steps / max(1, dom["launches"])
        achieved = dom_bytes * units_per_launch / (dom_ms * 1e-3) / 1e9
        # measured HBM traffic of the dominant kernel: a separate rocprofv3 --pmc pass (profiles/collect.sh), valid only
        # for the kernel sources it was collected on (source_hash), else null
        traffic, traffic_source = None, None
        if os.path.exists(args.traffic_json):
            try:
                tj = json.load(open(args.traffic_json))
                if tj.get("source_hash") == irt._lib.source_hash():
                    traffic = tj.get(dom_name, {}).get("hbm_bytes_per_launch")
                    traffic_source = tj.get("source")
                else:
                    traffic_source = "stale: %s was collected on other kernel sources" % os.path.relpath(args.traffic_json, ROOT)
            except Exception:
                traffic = None
        # fp64 flops per check: flops per RK4 step counted from this source tree's gfx950 assembly
        # (profiles/isa_counts.json <- profiles/count_isa.py; FMA = 2) x RK4 steps per configuration.  Initial
        # bending and the sweep's arithmetic are not counted.
        isa = load_isa_counts()
        # fk_verdict and fk_sweep_fused hold the same RK4 step (rk4_step<N>: the body compiled without any per-point hook);
        # the per-point sweep / signature hooks and the deferred walks are NOT counted as useful flops
        isa_key = {"fk_verdict": "rk4_step<%d>" % N, "fk_sweep_fused": "rk4_step<%d>" % N,
                   "fk_rk4_batch": "fk_rk4_batch_uniform<%d,false,false>" % N}.get(dom_name)
        flops_per_step = isa.get(isa_key, {}).get("flops_per_step")
        flops_per_check = flops_per_step * (P - 1) if flops_per_step else None
        hbm = {"achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
               "algorithmic_bytes_per_check": dom_bytes}
        valu = None
        flops_alg = algorithmic_flops_per_rk4_step(N) * (P - 1)
        if flops_per_check:
            tf = flops_per_check * units_per_launch / (dom_ms * 1e-3) / 1e12
            valu = {"achieved": tf, "peak": FP64_VALU_PEAK_TF, "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TF,
                    "flops_per_check": flops_per_check, "flops_per_rk4_step": flops_per_step, "rk4_steps": P - 1,
                    "flops_source": "profiles/isa_counts.json[%s]" % isa_key}
        # the roofline that binds the kernel is the one it sits closest to; the other is kept beside it
        if valu and valu["frac"] >= hbm["frac"]:
            # `frac` prices the launch with the ALGORITHMIC flop count (hand count of the restructured right-hand side:
            # algorithmic_flops_per_rk4_step); the executed count of the compiled kernel is kept beside it
            tf_alg = flops_alg * units_per_launch / (dom_ms * 1e-3) / 1e12
            roofline = dict(bound="fp64_valu", kernel=dom_name, achieved=tf_alg, peak=valu["peak"], unit=valu["unit"],
                            frac=tf_alg / valu["peak"], traffic=traffic, traffic_source=traffic_source, avg_launch_ms=dom_ms,
                            flops_per_check_algorithmic=flops_alg, flops_algorithmic_source="bench.py: algorithmic_flops_per_rk4_step (hand count, DESIGN.md section 5)",
                            flops_per_check=flops_per_check, flops_source=valu["flops_source"],
                            executed={"achieved": valu["achieved"], "frac": valu["frac"]}, hbm=hbm)
        else:
            roofline = dict(bound="hbm", kernel=dom_name, achieved=hbm["achieved"], peak=hbm["peak"], unit=hbm["unit"],
                            frac=hbm["frac"], traffic=traffic, traffic_source=traffic_source, avg_launch_ms=dom_ms,
                            algorithmic_bytes_per_check=dom_bytes, fp64_valu=valu)
        out = {
            "metric": "FK+collision checks/sec (3-tendon, 256^3 voxel env)",
            "value": checks / elapsed,
            "unit": "checks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "configs[1]: 3-tendon helical-routed robot (C=[2*pi*k/3, 5], D=[0.01], L=0.2, dL=L/128, "
                                   "129 backbone points), 2^%d seeded configs per GPU per step (tau~U[0,10)^3), batched FK + "
                                   "256^3 voxel collision (64 seeded spheres r=0.02 in reach)" % args.batch_log2,
                       "rehearsal_shared_gpu": rehearsal, "batch_per_gpu": n, "parallelism": "shard%d+allgather(bitmask)" % world if use_dist else "single",
                       "collective": ("gloo(host)" if rehearsal else "rccl") if use_dist else None,
                       "allgather_ms_per_step": allgather_ms if use_dist else None,
                       "allgather_bytes_per_rank": n // 8 if use_dist else None,
                       "valid_fraction_rank0": float(valid.mean())},
            "roofline": roofline,
            "kernels": {"fk_verdict": {"avg_ms": kv_ms, "launches": kv["launches"], "bytes_per_check": kvb},
                        "fk_sweep_fused": {"avg_ms": kf_ms, "launches": kf["launches"], "bytes_per_check": k1b + k2b},
                        "fk_rk4_batch": {"avg_ms": k1_ms, "launches": k1["launches"], "bytes_per_check": k1b},
                        "backbone_voxel_sweep": {"avg_ms": k2_ms, "launches": k2["launches"], "bytes_per_check": k2b}},
        }
        if world == 1 and not args.no_cpu_baseline:
            # PCIe-inclusive rate through the host-buffer entry point (pageable numpy arrays in, bits + tips out);
            # reported for DESIGN.md, never the headline value
            eng.validate_batch(states, True, False)           # first call creates the pinned staging buffers
            best = float("inf")
            for _ in range(3):
                t1 = time.perf_counter()
                eng.validate_batch(states, True, False)
                best = min(best, time.perf_counter() - t1)
            out["pcie_inclusive_checks_per_s"] = n / best
            cb, cpu_valid, m = cpu_baseline(irt, robot, vox, states)
            out["cpu_baseline"] = cb
            out["config"]["verdicts_match_cpu_sample"] = bool(np.array_equal(valid[:m], cpu_valid))
            if not args.no_extras:
                out["extras"] = secondary_metrics()
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
