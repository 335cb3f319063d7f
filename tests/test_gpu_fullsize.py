"""Properties at BASELINE's full size (2^20 configurations, config 2) that do not need the oracle to
run for minutes: determinism, chunking invariance, permutation equivariance, rotation equivariance of
FK, zero-tension known answer, agreement of the three routes to a voxel verdict, plus an oracle check
on a random sample of the big batch."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 1 << 20


@pytest.fixture(scope="module")
def setup(irt):
    import torch
    W = irt.workloads
    robot = W.robot_config2()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    states = irt.distributed.candidate_states(robot, seed=2024, start=0, count=N, tau_max=10.0)
    d_states = torch.from_numpy(states).cuda()
    return dict(robot=robot, vox=vox, chk=chk, states=states, d_states=d_states, torch=torch)


def _bits(setup, d_states, n):
    torch = setup["torch"]
    bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    tips = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
    setup["chk"].engine.validate_batch_dev(d_states, n, bits, tips, flags)
    torch.cuda.synchronize()
    return bits.cpu().numpy().view(np.uint64), tips.cpu().numpy(), flags.cpu().numpy()


def test_full_batch_deterministic_and_chunk_invariant(irt, setup):
    b1, t1, f1 = _bits(setup, setup["d_states"], N)
    b2, t2, f2 = _bits(setup, setup["d_states"], N)
    assert np.array_equal(b1, b2) and np.array_equal(t1, t2) and np.array_equal(f1, f2)
    valid = irt.unpack_bits(b1, N)
    assert 0.8 < valid.mean() < 0.9 and np.array_equal(valid, f1 == 15)
    # the same configurations in ragged pieces (sizes not multiples of 64) give the same verdicts
    torch = setup["torch"]
    cuts = [0, 1, 64, 1000, 65537, 300001, N]
    pieces = []
    for a, b in zip(cuts[:-1], cuts[1:]):
        d = torch.from_numpy(setup["states"][a:b]).cuda()
        pieces.append(irt.unpack_bits(_bits(setup, d, b - a)[0], b - a))
    assert np.array_equal(np.concatenate(pieces), valid)
    setup["valid"], setup["tips"], setup["flags"] = valid, t1, f1


def test_permutation_equivariance(irt, setup):
    torch = setup["torch"]
    perm = np.random.default_rng(0).permutation(N)
    d = torch.from_numpy(setup["states"][perm]).cuda()
    b, t, _ = _bits(setup, d, N)
    assert np.array_equal(irt.unpack_bits(b, N), setup["valid"][perm])
    assert np.array_equal(t, setup["tips"][perm])              # lanes do not influence each other: bit-identical


def test_sample_of_full_batch_matches_oracle(irt, orc, helpers, setup):
    idx = np.random.default_rng(1).choice(N, 6000, replace=False)
    want, tips, _ = orc.validate_batch(helpers.oracle_robot(orc, setup["robot"], lib="omp"),
                                       helpers.oracle_grid(orc, setup["vox"]), setup["states"][idx],
                                       nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(setup["valid"][idx], want)
    assert np.abs(setup["tips"][idx] - tips).max() <= 1e-9


def test_three_routes_to_the_voxel_verdict_agree(irt, setup):
    """validate (K1+K2) == voxelize (K5) + check_cached (K4), on 2^17 configurations."""
    n = 1 << 17
    eng = setup["chk"].engine
    out = eng.voxelize_batch(setup["states"][:n])
    hit = eng.check_cached(out["block_ids"], out["masks"], out["offsets"])
    fl = setup["flags"][:n]
    shape_ok = (fl & 7) == 7
    assert np.array_equal(out["shape_valid"], shape_ok)
    assert np.array_equal(~hit[shape_ok], (fl[shape_ok] & 8) != 0)
    # (the voxel caches come from the stored-point kernel, the flags above from fk_verdict: the same RK4 body in two kernels,
    # whose multiply-adds hipcc may contract differently -- equal to rounding)
    assert np.abs(out["tips"] - setup["tips"][:n]).max() <= 1e-13


def test_fk_rotation_equivariance_full_size(irt):
    """shape([tau, theta]) = Rz(theta) * shape([tau, 0]) for 2^18 configurations (TendonResult::rotate_z)."""
    import torch
    W = irt.workloads
    robot = W.robot_config2()
    robot.enable_rotation = True
    n = 1 << 18
    st = W.random_states(robot, n, seed=91, tau_max=12.0)
    st0 = st.copy(); st0[:, 3] = 0.0
    eng = robot.engine()
    P = eng.num_points
    eng.reserve(n)

    def run(s):
        d = torch.from_numpy(s).cuda()
        px, py, pz = (torch.empty(P * n, dtype=torch.float64, device="cuda") for _ in range(3))
        Li = torch.empty(3 * n, dtype=torch.float64, device="cuda")
        conv = torch.empty(n, dtype=torch.uint8, device="cuda")
        eng.fk_batch_dev(d, n, n, px, py, pz, d_Li=Li, d_conv=conv)
        torch.cuda.synchronize()
        return px.view(P, n), py.view(P, n), pz.view(P, n), Li, conv
    x, y, z, Li, conv = run(st)
    x0, y0, z0, Li0, conv0 = run(st0)
    th = torch.from_numpy(st[:, 3]).cuda()
    c, s = torch.cos(th), torch.sin(th)
    assert float((x - (c * x0 - s * y0)).abs().max()) < 1e-14
    assert float((y - (s * x0 + c * y0)).abs().max()) < 1e-14
    assert float((z - z0).abs().max()) < 1e-15
    assert torch.equal(Li, Li0) and torch.equal(conv, conv0)


def test_zero_tension_known_answer_full_size(irt, setup):
    """tau = 0 for 2^16 configurations: the home shape (straight, in free space by construction) is valid."""
    torch = setup["torch"]
    n = 1 << 16
    d = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
    b, t, f = _bits(setup, d, n)
    assert irt.unpack_bits(b, n).all() and (f == 15).all()
    assert np.abs(t - np.array([0.0, 0.0, 0.2])).max() < 1e-13


def test_config4_sizes_on_one_rank(irt):
    """BASELINE config 4's sizes through one rank's code path (world size 1): the mask of 2^20 candidate vertices from the
    sharded validator equals direct validation of the same candidate sequence in ragged pieces; two ranks' shares of the
    neighbour table (tr_knn_range over ~600 k accepted vertices) are the rows a 4 096-query probe of the whole search gives;
    the edge list derived from gathered rows is symmetric-free, ordered and duplicate-free; a shard of those edges validates
    the same alone and inside a larger batch."""
    import torch
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    eng = chk.engine
    M = 1 << 20

    def validate_local(states):
        d = torch.from_numpy(states).cuda()
        bits = torch.zeros((len(states) + 63) // 64, dtype=torch.int64, device="cuda")
        eng.validate_batch_dev(d, len(states), bits)
        torch.cuda.synchronize()
        return bits.cpu().numpy()

    mask = irt.unpack_bits(D.ShardedVertexValidator(robot, validate_local, seed=3, device="cuda").run(M, rank=0, world_size=1), M)
    cand = D.candidate_states(robot, 3, 0, M)
    for lo, hi in ((0, 70001), (400000, 470003), (M - 50001, M)):             # ragged pieces, not multiples of 64
        assert np.array_equal(chk.is_valid(cand[lo:hi]), mask[lo:hi])
    assert 0.5 < mask.mean() < 0.7
    verts = np.ascontiguousarray(cand[mask])
    n = len(verts)
    k = 11
    shares = [D.shard_bounds(n, 8, r) for r in (0, 5)]
    rows = [eng.knn(verts, k, query_range=(s0, min(s1, n) - s0))[0] for s0, s1, _ in shares]
    for (s0, s1, _), r in zip(shares, rows):
        assert r.shape == (min(s1, n) - s0, k) and np.array_equal(r[:, 0], np.arange(s0, min(s1, n)))     # every vertex is its own nearest
        # neighbour lists of a probe inside the share, computed as a small range of their own (other slicing of the candidates)
        p0 = s0 + 1234
        probe = eng.knn(verts, k, query_range=(p0, 4096))[0]
        assert np.array_equal(probe, r[1234:1234 + 4096])
        d = np.linalg.norm(verts[r[:64, 1:]] - verts[s0:s0 + 64, None, :], axis=2)
        assert (np.diff(d, axis=1) >= 0).all()                               # ordered by distance
    table = np.full((n, k), -1, np.int32)
    for (s0, s1, _), r in zip(shares, rows):
        table[s0:s0 + len(r)] = r
    edges = eng.edges_from_knn(table)
    assert (edges[:, 0] < edges[:, 1]).all()
    key = edges[:, 0].astype(np.int64) << 32 | edges[:, 1]
    assert (np.diff(key) > 0).all() and len(edges) > 0.9 * (k - 1) * sum(len(r) for r in rows) * 0.5
    mv = irt.VoxelBackboneMotionValidator(chk)
    part = edges[:200000]
    whole = mv.check_motion_indexed(verts, part)
    alone = mv.check_motion_indexed(verts, part[70000:90000])
    assert np.array_equal(whole["valid"][70000:90000], alone["valid"]) and np.array_equal(whole["n_fk"][70000:90000], alone["n_fk"])
    assert whole["valid"].mean() > 0.9


def test_full_size_rotation_retraction_robot_order_invariance(irt, orc, helpers):
    """The planner's full state space at full size: config 3's robot with rotation and retraction, 2^20 states over the whole
    space.  A batch this large is integrated in the order of its backbone lengths (fk_verdict_retract); the same states in ragged
    pieces -- pieces below 8192 states run in arrival order -- and as a permutation must give the same verdict, flags and tip
    for every configuration, bit for bit (a lane's result does not depend on the wave it runs in), with the device-side
    ordering switched off as well, and equal to the oracle's on a sample."""
    import os
    import torch
    W = irt.workloads
    robot = W.robot_config3()
    robot.enable_rotation = True
    robot.enable_retraction = True
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    states = W.random_states(robot, N, seed=77, tau_max=20.0)
    states[:64, -1] = np.linspace(0.0, 0.2, 64)                   # both ends of the range, exactly

    def run(chk, st):
        n = len(st)
        d = torch.from_numpy(np.ascontiguousarray(st)).cuda()
        bits = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
        tips = torch.zeros(n, 3, dtype=torch.float64, device="cuda")
        flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
        chk.engine.validate_batch_dev(d, n, bits, tips, flags)
        torch.cuda.synchronize()
        return irt.unpack_bits(bits.cpu().numpy().view(np.uint64), n), tips.cpu().numpy(), flags.cpu().numpy()

    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    v, t, f = run(chk, states)
    assert 0.5 < v.mean() < 0.95 and np.array_equal(v, f == 15)
    v2, t2, f2 = run(chk, states)
    assert np.array_equal(v, v2) and np.array_equal(t, t2) and np.array_equal(f, f2)
    cuts = [0, 1, 64, 1000, 8191, 8192 + 8191, 65537, 300001, N]
    pv, pt, pf = zip(*[run(chk, states[a:b]) for a, b in zip(cuts[:-1], cuts[1:])])
    assert np.array_equal(np.concatenate(pv), v) and np.array_equal(np.concatenate(pt), t) and np.array_equal(np.concatenate(pf), f)
    perm = np.random.default_rng(5).permutation(N)
    vp, tp, fp = run(chk, states[perm])
    assert np.array_equal(vp, v[perm]) and np.array_equal(tp, t[perm]) and np.array_equal(fp, f[perm])
    os.environ["TENDON_HIP_RETRACT_SORT"] = "0"
    try:
        va, ta, fa = run(irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox), states)
    finally:
        os.environ.pop("TENDON_HIP_RETRACT_SORT", None)
    assert np.array_equal(va, v) and np.array_equal(ta, t) and np.array_equal(fa, f)
    idx = np.random.default_rng(6).choice(N, 5000, replace=False)
    want, tips, _ = orc.validate_batch(helpers.oracle_robot(orc, robot, lib="omp"), helpers.oracle_grid(orc, vox), states[idx],
                                       nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(v[idx], want)
    ok = (f[idx] & 1) > 0
    assert np.abs(t[idx][ok] - tips[ok]).max() <= 1e-9


def test_full_size_roadmap_edges_two_lanes_equal_one(irt, orc, helpers):
    """Config 3 at full size: 100 k valid milestones, their 10-NN edge list (~588 k edges) validated by tr_validate_edges_indexed on the
    default number of lanes (four at this size: parts of the edge list on as many streams, pool of 2^24 per-sample slots without point
    planes), on two and on one lane (TENDON_HIP_EDGE_LANES): verdicts, FK counts and domain-error counts are equal edge for edge, and the oracle's on a sample."""
    import os
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    res = {}
    for lanes in ("1", "2", "default"):                 # default: by the edge count, four lanes at this size
        old = os.environ.get("TENDON_HIP_EDGE_LANES")
        os.environ.pop("TENDON_HIP_EDGE_LANES", None)
        if lanes != "default":
            os.environ["TENDON_HIP_EDGE_LANES"] = lanes
        try:
            chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        finally:
            os.environ.pop("TENDON_HIP_EDGE_LANES", None) if old is None else os.environ.__setitem__("TENDON_HIP_EDGE_LANES", old)
        rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
        states, _ = rb.sample_valid_vertices(100000, batch=1 << 17)
        edges = rb.knn_edges_gpu(states, 11)
        res[lanes] = chk.engine.validate_edges_indexed(states, edges, rb.mv.min_tension_change, rb.mv.min_rotation_change, rb.mv.min_retraction_change)
        res[lanes + "again"] = chk.engine.validate_edges_indexed(states, edges, rb.mv.min_tension_change, rb.mv.min_rotation_change,
                                                                   rb.mv.min_retraction_change)       # sized by the rate the first call saw
    assert len(edges) > 500000
    for k in ("valid", "n_fk", "n_domain_errors"):
        assert np.array_equal(res["1"][k], res["2"][k]) and np.array_equal(res["2"][k], res["2again"][k]) and np.array_equal(res["1"][k], res["1again"][k]), k
        assert np.array_equal(res["1"][k], res["default"][k]) and np.array_equal(res["1"][k], res["defaultagain"][k]), k
    assert 0.9 < res["2"]["valid"].mean() < 1.0 and res["2"]["n_fk"].max() > 10
    idx = np.random.default_rng(8).choice(len(edges), 3000, replace=False)
    ov, onf, _ = orc.check_motion_batch(helpers.oracle_robot(orc, robot, lib="omp"), helpers.oracle_grid(orc, vox), states[edges[idx, 0]],
                                        states[edges[idx, 1]], nthreads=0, lib=orc.omp_lib())
    assert np.array_equal(res["2"]["valid"][idx], ov) and np.array_equal(res["2"]["n_fk"][idx][ov], onf[ov])


def test_full_size_config4_stays_in_hbm_and_agrees_with_the_host_forms(irt):
    """Config 4 at full size on one GPU: 2^20 candidates validated with signature output, the ~6 x 10^5 accepted vertices and their rows
    compacted in HBM, the 10-NN edge list (~3.5 M edges) built there and validated with and without the signature hand-over: the two
    masks are the same words, and the same as the host-array forms' (tr_knn_edges + tr_validate_edges_indexed on downloaded vertices)."""
    import torch
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    eng, mv = chk.engine, irt.VoxelBackboneMotionValidator(chk)
    M, k, seed, S, sw, box = 1 << 20, 11, 3, eng.state_size, eng.signature_words(), D.sampling_box(robot)
    d_mask = torch.zeros(M // 64, dtype=torch.int64, device="cuda")
    d_sig = torch.empty((M, sw), dtype=torch.int32, device="cuda")
    eng.validate_candidates_sig_dev(seed, 0, M, d_mask, d_sig, box=box)
    d_verts, _ = D.gather_valid_vertices_dev(eng, seed, M, d_mask, box=box)
    nv = d_verts.shape[0]
    d_vsig = D.device_row_compactor(eng)(d_mask, M, d_sig)
    del d_sig
    assert d_vsig.shape == (nv, sw) and 500000 < nv < 700000
    d_edges = torch.empty((nv * k, 2), dtype=torch.int32, device="cuda")
    ne = eng.knn_edges_dev(d_verts, nv, k, d_edges)
    space = (mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change)
    words = {}
    for name, sig in (("integrated", None), ("handed over", d_vsig)):
        words[name] = torch.zeros((ne + 63) // 64, dtype=torch.int64, device="cuda")
        assert eng.validate_edges_indexed_dev(d_verts, nv, d_edges, ne, words[name], None, *space, d_vertex_sig=sig) == 0
    assert torch.equal(words["integrated"], words["handed over"])
    verts = d_verts.cpu().numpy()
    edges = eng.knn_edges(verts, k)
    assert len(edges) == ne > 3000000 and np.array_equal(d_edges[:ne].cpu().numpy(), edges)
    host = eng.validate_edges_indexed(verts, edges, *space)
    got = irt.unpack_bits(words["handed over"].cpu().numpy().view(np.uint64), ne)
    assert np.array_equal(got, host["valid"]) and 0.98 < got.mean() < 1.0 and host["n_domain_errors"] == 0


def test_full_size_config5_query_loop(irt, orc, helpers, monkeypatch):
    """BASELINE configs[4] at full size: a 10^5-vertex roadmap with vertex and edge voxel caches (built and kept on the device), the
    obstacle grid changed (8 more spheres), 10 000 start / goal queries through the lazy loop (tr_roadmap_solve) and through the
    eager form (tr_roadmap_revalidate, then search) -- the oracle's sequential query loop on 320 sampled queries (status, path,
    cost), the revalidation counts against the oracle's cached-set test of EVERY item, and every vertex and edge of every returned
    path validated from scratch (FK + predicate, checkMotion) in the new environment."""
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    new_vox, _ = W.reach_environment(seed=7, n_spheres=72)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
    states, _ = rb.sample_valid_vertices(100000, batch=1 << 17)
    edges = rb.knn_edges_gpu(states, 11)
    valid, _ = rb.validate_edges(states, edges)
    e_ok = edges[valid]                                        # createRoadmap removes invalid edges (:1543-1551)
    vc, ec = rb.vertex_caches(states), rb.edge_caches(states, e_ok)
    assert len(e_ok) > 500000 and vc["shape_valid"].all() and ec["fully_valid"].all()
    vd, ed = rb.vertex_caches(states, device=True), rb.edge_caches(states, e_ok, device=True)
    prm = irt.VoxelCachedLazyPRM(chk, states, e_ok)
    prm.set_caches(vd, ed)
    del vd, ed
    prm.set_obstacles(new_vox)
    nq = 10000
    pairs = np.random.default_rng(17).integers(0, len(states), size=(nq, 2))
    # default: the loop tests every cached set in one launch once that is cheaper than another round of searches ...
    dflt = prm.solveWithRoadmap(pairs[:, 0], pairs[:, 1])
    st_dflt = dict(prm.stats)
    prm.clearValidity()
    # ... strictly lazy (the reference's loop item by item), which is what the oracle restates
    monkeypatch.setenv("TENDON_HIP_LAZY_ONLY", "1")
    lazy = prm.solveWithRoadmap(pairs[:, 0], pairs[:, 1])
    monkeypatch.delenv("TENDON_HIP_LAZY_ONLY")
    st_lazy = dict(prm.stats)
    v_lazy, e_lazy = prm.validity()
    assert st_lazy["rounds"] >= 2 and 0 < st_lazy["items_checked"] < 0.6 * (len(states) + len(e_ok))     # lazy: most items never looked at
    assert np.array_equal(dflt["status"], lazy["status"]) and np.array_equal(dflt["cost"], lazy["cost"])
    assert np.array_equal(dflt["path_vertices"], lazy["path_vertices"]) and st_dflt["rounds"] <= st_lazy["rounds"]
    solved = lazy["status"] == 0
    assert 0.8 < solved.mean() < 1.0 and (lazy["status"] >= 2).any()
    # the oracle: its own query loop on a sample of the queries; the cached-set test of every item
    orb, og = helpers.oracle_robot(orc, robot, lib="omp"), helpers.oracle_grid(orc, new_vox)
    hit_v = orc.check_cached(og, vc["block_ids"], vc["masks"], vc["offsets"])
    hit_e = orc.check_cached(og, ec["block_ids"], ec["masks"], ec["offsets"])
    assert np.array_equal(v_lazy[v_lazy > 0] == 2, hit_v[v_lazy > 0]) and np.array_equal(e_lazy[e_lazy > 0] == 2, hit_e[e_lazy > 0])
    orm = orc.Roadmap(orb, states, e_ok, None, vc, ec, lib=orc.omp_lib())
    code = {-2: 2, -3: 3, 0: 1}
    sample = np.random.default_rng(18).choice(nq, 320, replace=False)
    for q in sample:
        w = orm.query(og, pairs[q, 0], pairs[q, 1])
        assert lazy["status"][q] == (0 if w["n"] > 0 else code[w["n"]]), (q, w["n"], lazy["status"][q])
        if w["n"] > 0:
            assert lazy["cost"][q] == w["cost"] and np.array_equal(lazy["paths"][q], w["path"]), (q, w["cost"], lazy["cost"][q])
    # eager: one pass over all 6.8 x 10^5 cached sets, then the same answers without a single item check
    prm.clearValidity()
    nv, ne = prm.revalidate()
    assert nv == int(hit_v.sum()) > 0 and ne == int(hit_e.sum()) > 0
    v_all, e_all = prm.validity()
    assert np.array_equal(v_all == 2, hit_v) and np.array_equal(e_all == 2, hit_e) and (v_all > 0).all() and (e_all > 0).all()
    eager = prm.solveWithRoadmap(pairs[:, 0], pairs[:, 1])
    assert prm.stats["rounds"] <= 1 and prm.stats["items_checked"] == 0
    assert np.array_equal(eager["status"], lazy["status"]) and np.array_equal(eager["cost"], lazy["cost"])
    assert np.array_equal(eager["path_vertices"], lazy["path_vertices"]) and np.array_equal(eager["path_offsets"], lazy["path_offsets"])
    # every returned path from scratch in the new environment
    chk2 = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), new_vox)
    pv = np.unique(lazy["path_vertices"])
    assert chk2.is_valid(states[pv]).all()
    po, pvx = lazy["path_offsets"], lazy["path_vertices"]
    inner = np.ones(len(pvx), dtype=bool)
    inner[po[1:][po[1:] > po[:-1]] - 1] = False                # the last vertex of every non-empty path starts no edge
    a, b = pvx[:-1][inner[:-1]], pvx[1:][inner[:-1]]
    und = np.unique(np.stack([np.minimum(a, b), np.maximum(a, b)], 1), axis=0)
    assert len(und) > 20000
    assert irt.VoxelBackboneMotionValidator(chk2).check_motion(states[und[:, 0]], states[und[:, 1]]).all()
    # a path's ends are the query's, its cost the sum of its edges' state-space distances (tension-only space: Euclidean)
    for q in np.flatnonzero(solved)[:200]:
        p = lazy["paths"][q]
        assert p[0] == pairs[q, 0] and p[-1] == pairs[q, 1]
        d = sum(np.linalg.norm(states[x] - states[y]) for x, y in zip(p[:-1], p[1:]))
        assert abs(d - lazy["cost"][q]) <= 1e-12 * max(1.0, d)
