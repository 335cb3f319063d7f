#!/usr/bin/env python3
"""One pass over EVERY kernel of the library at BASELINE sizes, for rocprofv3 (profiles/collect_all.sh runs it under
--kernel-trace --stats and under separate --pmc passes).  Writes <out>/units.json: per kernel name the ALGORITHMIC work
of all of its launches in this run (bytes and / or fp64 flops, SURVEY.md section 8d's per-unit figures x units), which
profiles/summarize_all.py divides by the kernel's total duration.

    python profiles/workload_all.py <out_dir> [--small]
"""
import importlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def with_env(env, fn):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return fn()
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


def main():
    out_dir = sys.argv[1]
    small = "--small" in sys.argv
    os.makedirs(out_dir, exist_ok=True)
    import torch
    irt = importlib.import_module("interactive-rate-tendons_amd")
    W = irt.workloads
    isa = json.load(open(os.path.join(ROOT, "profiles", "isa_counts.json")))
    units = {}

    def add(kernel, **kw):
        u = units.setdefault(kernel, {"bytes": 0.0, "flops": 0.0, "units": 0.0, "unit": kw.pop("unit", "")})
        for k, v in kw.items():
            u[k] = u.get(k, 0.0) + float(v)

    vox, centres = W.reach_environment(seed=7, n_spheres=64)
    reps = 3
    # ---- tr_validate_batch_dev in its three schedules, 3- and 4-tendon robots (configs 2 and 3) -------------------
    for robot, tag, tau, logn in ((W.robot_config2(), 3, 10.0, 20), (W.robot_config3(), 4, 20.0, 19)):
        n = 1 << (logn - (4 if small else 0))
        P, S, N = 129, robot.state_size(), len(robot.tendons)
        st = torch.from_numpy(W.random_states(robot, n, seed=3, tau_max=tau)).cuda()
        bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        tips = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        fl_fused = isa["rk4_step<%d>" % N]["flops_per_step"] * (P - 1)
        fl_k1 = isa["fk_rk4_batch_uniform<%d,false,false>" % N]["flops_per_step"] * (P - 1)
        for mode, names in (("2", ("fk_verdict",)), ("1", ("fk_sweep_fused",)), ("0", ("fk_rk4_batch_uniform", "backbone_voxel_sweep"))):
            chk = with_env({"TENDON_HIP_FUSED": mode}, lambda: irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox))
            chk.engine.reserve(n)
            for _ in range(reps):
                chk.engine.validate_batch_dev(st, n, bits, tips)
            torch.cuda.synchronize()
            k1b = 8 * S + 24 * P + 8 * (N + 1) + 1 + 24
            k2b = 24 * P + 8 * N + 1 + 0.125
            if mode == "2":
                add("fk_verdict<%d>" % N, bytes=reps * n * (8 * S + 24 + 0.125), flops=reps * n * fl_fused, units=reps * n, unit="checks")
            elif mode == "1":
                add("fk_sweep_fused<%d>" % N, bytes=reps * n * (k1b + k2b), flops=reps * n * fl_fused, units=reps * n, unit="checks")
            else:
                add("fk_rk4_batch_uniform<%d>" % N, bytes=reps * n * k1b, flops=reps * n * fl_k1, units=reps * n, unit="FK")
                add("backbone_voxel_sweep", bytes=reps * n * k2b, units=reps * n, unit="shapes")
            del chk
    # ---- retraction robots, s_start ~ U[0, L) as a planner samples it: K1r + K2 on stored points in arrival order
    # (TENDON_HIP_FUSED=1), and the verdict-only form on the batch ordered by backbone length (default) -----------------
    for mk, N in ((W.robot_config2, 3), (W.robot_config3, 4)):
        robot = mk()
        robot.enable_retraction = True
        n = 1 << (19 - (4 if small else 0))
        stn = W.random_states(robot, n, seed=4, tau_max=10.0 if N == 3 else 20.0)
        st = torch.from_numpy(stn).cuda()
        bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        # points per configuration ~ 129 * (1 - s/L) on average: count the algorithmic bytes and flops of the mean
        pm = float(np.mean(np.ceil((0.2 - stn[:, -1]) / (0.2 / 128)) + 1))
        for mode in ("1", "2"):
            chk = with_env({"TENDON_HIP_FUSED": mode}, lambda: irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox))
            chk.engine.reserve(n)
            for _ in range(reps):
                chk.engine.validate_batch_dev(st, n, bits)
            torch.cuda.synchronize()
            if mode == "1":
                add("fk_rk4_batch_retract<%d>" % N, bytes=reps * n * (8 * (N + 1) + 24 * pm + 16 * N + 5), units=reps * n, unit="FK",
                    flops=reps * n * isa["fk_rk4_batch_uniform<%d,false,false>" % N]["flops_per_step"] * (pm - 1))
                add("backbone_voxel_sweep", bytes=reps * n * (24 * pm + 16 * N + 5), units=reps * n, unit="shapes")
            else:
                add("fk_verdict_retract<%d>" % N, bytes=reps * n * (8 * (N + 1) + 24 + 0.125 + 4), units=reps * n, unit="checks",
                    flops=reps * n * isa["rk4_step<%d>" % N]["flops_per_step"] * (pm - 1))
            del chk
    # ---- config 3 roadmap: vertices, k-NN, indexed edges, caches, K4 ---------------------------------------------------
    robot = W.robot_config3()
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
    V = 100000 // (16 if small else 1)
    states, _ = rb.sample_valid_vertices(V, batch=1 << 17)
    k = 10
    edges = rb.knn_edges_gpu(states, k + 1)
    # since the search runs on sorted states a wave visits only the candidates within its seed radius: V^2 pairs are DECIDED,
    # about a fifth of them computed -- no flop or byte figure is claimed for the row
    add("knn_wave_query<4>", units=float(V) * V, unit="pair distances decided (a wave per query, three-key cell grid: ~10^3 computed per query)")
    chk.engine.reserve_edges(len(edges))
    chk.engine.profile_begin()
    valid, nfk = rb.validate_edges(states, edges)
    prof_e = chk.engine.profile_read()
    chk.engine.profile_end()
    n_samples = int(nfk.sum()) - 2 * len(edges) + V          # FK samples actually integrated (vertices once)
    add("edge path (tr_validate_edges_indexed)", units=len(edges), unit="edges", fk_samples=n_samples)
    # the launches behind the roadmap calls, so that the FK kernels' rows cover ALL of their launches: vertex sampling and
    # the edge samples run through fk_verdict<4> (the samples also write a 4 P-byte signature row), the voxel caches below
    # through fk_sweep_fused<4> (stored points)
    fl4 = isa["rk4_step<4>"]["flops_per_step"] * 128
    cand = rb.timing["vertices"]["candidates"]
    add("fk_verdict<4>", bytes=(cand + V) * (8 * 4 + 24 + 0.125), flops=(cand + V) * fl4, units=cand + V, unit="checks")   # vertex sampling + the edge call's vertex pass? no: see below
    # the edge call evaluates its V vertices and its own samples through the SIG variant (a 4 P-byte signature row per sample)
    units["fk_verdict<4>"]["bytes"] -= V * (8 * 4 + 24 + 0.125); units["fk_verdict<4>"]["flops"] -= V * fl4; units["fk_verdict<4>"]["units"] -= V
    add("fk_verdict<4> +sig", bytes=n_samples * (8 * 4 + 4 * 129 + 0.125), flops=n_samples * fl4, units=n_samples,
        unit="edge samples (up to four lanes' launches share the GPU, so the per-launch durations this row sums overlap: the call as a whole integrates 8.6e7 samples/s, r03/edge_timeline_v2.txt)")
    add("edge_filter", bytes=2.0 * 129 * 4 * (n_samples + len(edges)), units=n_samples + len(edges), unit="interval tests on cell signatures (upper bound: early exit from the tip)")
    # the same edges through the device-resident form: the edge queue (round 4), ONE persistent launch that integrates the samples,
    # folds them into their edges and finishes the levels (its vertex pass is one more launch of the +sig kernel)
    import torch
    d_states, d_edges = torch.from_numpy(states).cuda(), torch.from_numpy(np.ascontiguousarray(edges, dtype=np.int32)).cuda()
    d_bits = torch.zeros((len(edges) + 63) // 64, dtype=torch.int64, device="cuda")
    mvq = rb.mv
    for _ in range(2):
        chk.engine.validate_edges_indexed_dev(d_states, V, d_edges, len(edges), d_bits, None, mvq.min_tension_change, mvq.min_rotation_change, mvq.min_retraction_change)
    sch = chk.engine.edge_schedule_last()
    assert np.array_equal(irt.unpack_bits(d_bits.cpu().numpy().view(np.uint64), len(edges)), valid) and sch["flags"] == 0
    # (summarize_all.py takes every launch of a kernel to do the same work and drops the first: the queue's first launch is tr_reserve_edges'
    # warm-up over an EMPTY queue, so the two real launches' work is entered as three launches' worth)
    add("fk_edge_queue<4>", bytes=3 * sch["samples"] * (8 * 4 + 4 * 129 + 32 + 3 * 4 * 129), flops=3 * sch["samples"] * fl4, units=3 * sch["samples"],
        unit="edge samples through the edge queue (a sample: its state and interval record in, its signature row out, and the three rows its interval is tested against; the first launch is the warm-up over an empty queue)")
    units["fk_verdict<4> +sig"]["bytes"] += 2 * V * (8 * 4 + 4 * 129 + 0.125); units["fk_verdict<4> +sig"]["flops"] += 2 * V * fl4; units["fk_verdict<4> +sig"]["units"] += 2 * V
    del d_states, d_edges, d_bits
    e_ok = edges[valid]
    vc = rb.vertex_caches(states)
    ec = rb.edge_caches(states, e_ok)
    add("backbone_voxelize", bytes=(V + n_samples) * (24.0 * 129) + 12.0 * (int(vc["offsets"][-1])), units=V + n_samples, unit="shapes")
    n_cache_samples = 2 * V + int(ec["n_fk"].sum()) - 2 * len(e_ok)        # vertex caches, then the indexed edge caches (vertices once more)
    add("fk_sweep_fused<4>", bytes=n_cache_samples * (2 * 24.0 * 129 + 8 * 4 + 4 * 129), flops=n_cache_samples * fl4, units=n_cache_samples, unit="checks")
    add("cache merge (rocPRIM sort + reduce)", units=int(ec["offsets"][-1]), unit="unique edge blocks")
    new_vox, _ = W.reach_environment(seed=7, n_spheres=72)
    prm = irt.VoxelCachedLazyPRM(chk, states, e_ok)
    prm.set_caches(vc, ec)
    prm.set_obstacles(new_vox)
    nblk = int(vc["offsets"][-1] + ec["offsets"][-1])
    for _ in range(5):
        prm.clearValidity()
        prm.revalidate()
    add("cached_blocks_vs_grid", bytes=5 * (12.0 * nblk + 8.0 * (V + len(e_ok))), units=5 * (V + len(e_ok)), unit="cached sets",
        working_set_MiB=12.0 * nblk / 2 ** 20)
    rng = np.random.default_rng(17)
    q = rng.integers(0, V, size=(2000, 2))
    prm.clearValidity()
    # (strictly lazy: the default schedule would test every cached set once more -- a full-size cached_blocks_vs_grid launch that
    # the row above does not count; the search kernel has its own section below)
    with_env({"TENDON_HIP_SEARCH": "host", "TENDON_HIP_LAZY_ONLY": "1"}, lambda: prm.solveWithRoadmap(q[:, 0], q[:, 1]))
    add("cached_subset_vs_grid", units=prm.stats["items_checked"], unit="cached sets (lazy rounds)")
    # ---- the graph searches of the query loop on the device: 10 000 queries, validity known, every search on the kernel ------
    prm.prepare(16)
    q = rng.integers(0, V, size=(10000, 2))
    prm.clearValidity()
    prm.revalidate()
    # (a sixth launch of cached_blocks_vs_grid, same size as the five above: counted, so that the row's bytes match its launches)
    units["cached_blocks_vs_grid"]["bytes"] += 12.0 * nblk + 8.0 * (V + len(e_ok)); units["cached_blocks_vs_grid"]["units"] += V + len(e_ok)
    sreps, dev_exp = 3, 0
    for _ in range(sreps):                  # every round on the kernel, with the default budget (a search over it is handed back: its expansions so far count)
        with_env({"TENDON_HIP_SEARCH": "device", "TENDON_HIP_SEARCH_BUDGET": "10000"}, lambda: prm.solveWithRoadmap(q[:, 0], q[:, 1]))
        dev_exp += prm.search_stats["expanded_device"]
    deg = 2.0 * len(e_ok) / V
    S_, L_ = states.shape[1], 16
    add("roadmap_astar", bytes=dev_exp * (48 + deg * (16 + 2 + 1 + 32 + 8 * S_ + 4 * L_ + 32)), units=dev_exp,
        unit="vertex expansions (three times 10 000 searches, a wave each; bytes as tr_roadmap_profile counts them: record + row header + per arc: arc, validity, arc count, neighbour record read and written, state and landmark rows)",
        searches=prm.search_stats["device"], list_moves=prm.search_stats["list_moves"])
    # ---- sphere checker (K8) and environment edits (K7) ------------------------------------------------------------------
    r2 = W.robot_config2()
    raw = irt.VoxelOctree(256)
    raw.set_xlim(-0.25, 0.25); raw.set_ylim(-0.25, 0.25); raw.set_zlim(-0.25, 0.25)
    for c in centres:
        raw.add_sphere(c, 0.005)
    sc = irt.VoxelValidityChecker(r2, irt.VoxelEnvironment(), raw)
    n = 1 << (20 - (4 if small else 0))
    st = torch.from_numpy(W.random_states(r2, n, seed=3, tau_max=10.0)).cuda()
    bits = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
    sc.engine.reserve(n)
    for _ in range(reps):
        sc.engine.validate_batch_dev(st, n, bits)
    torch.cuda.synchronize()
    # since round 2 the sphere-swept checker runs through the verdict-only kernel too (one distance-field gather per point,
    # 4 B, from a 64 MiB field that lives in the Infinity Cache: not counted as algorithmic HBM bytes)
    add("fk_verdict<3> spheres", bytes=reps * n * (8 * 3 + 24 + 0.125), flops=reps * n * isa["rk4_step<3>"]["flops_per_step"] * 128,
        units=reps * n, unit="checks")
    # K8 on stored points, as retraction robots still use it
    sc0 = with_env({"TENDON_HIP_FUSED": "1"}, lambda: irt.VoxelValidityChecker(r2, irt.VoxelEnvironment(), raw))
    sc0.engine.reserve(n)
    for _ in range(reps):
        sc0.engine.validate_batch_dev(st, n, bits)
    torch.cuda.synchronize()
    add("spheres_vs_grid", bytes=reps * n * (24.0 * 129 + 4 * 129), units=reps * n, unit="shapes")
    add("fk_sweep_fused<3>", bytes=reps * n * (2 * 24.0 * 129 + 8 * 3 + 8 * 4 + 1.125), flops=reps * n * isa["rk4_step<3>"]["flops_per_step"] * 128,
        units=reps * n, unit="checks")
    del sc0
    extra = np.column_stack([rng.uniform(-0.15, 0.15, (8, 3)), np.full(8, 0.02)])
    caps = np.column_stack([rng.uniform(-0.15, 0.15, (8, 3)), rng.uniform(-0.15, 0.15, (8, 3)), np.full(8, 0.01)])
    for _ in range(3):
        chk.engine.set_grid(new_vox.Nx(), new_vox.limits(), new_vox.blocks)
        chk.add_spheres(extra); chk.add_capsules(caps); chk.dilate_sphere(robot.r); chk.remove_interior()
    add("environment edits (grid_add_spheres / grid_add_capsules / grid_dilate_step / grid_remove_interior / dilate2_blocks)",
        bytes=3 * 2.0 * 2 ** 21, units=3, unit="edit sequences on a 2 MiB grid")
    units["_edge_path_profile_slots"] = prof_e
    units["_meta"] = {"vertices": V, "edges": int(len(edges)), "valid_edges": int(len(e_ok)), "fk_samples": n_samples,
                      "cache_blocks": nblk, "small": small}
    json.dump(units, open(os.path.join(out_dir, "units.json"), "w"), indent=1)
    print("workload done:", json.dumps(units["_meta"]))


if __name__ == "__main__":
    main()
