"""The C++ host shim (include/tendon_hip_shim.hpp) -- the layer reference-side C++ code would
call -- compiled with g++ against libtendon_hip.so.  CPU: it compiles, links and maps errors to the
reference's exception types.  GPU: its results match the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "interactive-rate-tendons_amd")


def _build(tmp_path, irt, name="shim_test"):
    irt.build()
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", name + ".cpp"), "-o", exe, "-L", PKG, "-ltendon_hip",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_shim_compiles_and_maps_errors(tmp_path, irt):
    exe = _build(tmp_path, irt)
    out = subprocess.check_output([exe, "--no-gpu"], text=True)
    assert "caught 3" in out
    _build(tmp_path, irt, "shim_roadmap_test")            # the roadmap builder's side of the header compiles and links too


@pytest.mark.gpu
def test_shim_results_match_oracle(tmp_path, irt, orc, helpers):
    exe = _build(tmp_path, irt)
    out = subprocess.check_output([exe], text=True).splitlines()
    robot = irt.workloads.robot_config2()
    orb = helpers.oracle_robot(orc, robot)
    states = [[0, 0, 0], [8, 3, 1], [2.5, 9.0, 4.0], [19.0, 0.5, 17.0]]
    shapes = [l.split() for l in out if l.startswith("shape")]
    assert len(shapes) == 4
    for s, row in zip(states, shapes):
        want = orb.shape(s)
        assert int(row[1]) == 129 and int(row[6]) == int(want["converged"])
        assert np.abs(np.array(row[2:5], float) - want["p"][-1]).max() <= 1e-9
        assert abs(float(row[5]) - want["L_i"][0]) <= 1e-10
    vox = irt.VoxelOctree(256)
    vox.set_xlim(-0.25, 0.25); vox.set_ylim(-0.25, 0.25); vox.set_zlim(-0.25, 0.25)
    og = helpers.oracle_grid(orc, vox)
    for ix in range(143, 154):
        for iy in range(64, 192):
            for iz in range(179, 240):
                og.set_cell(ix, iy, iz)
    want = [orc.is_valid_state(orb, og, s) for s in states]
    single = [int(l.split()[1]) for l in out if l.startswith("isValid")]
    batch = [l.split() for l in out if l.startswith("batch")]
    assert single == [int(w[0]) for w in want]
    assert [int(b[1]) for b in batch] == single
    assert [int(b[2]) for b in batch] == [w[2] for w in want]
    assert len(set(single)) == 2                       # the slab blocks some and not others
    errs = [l for l in out if l.startswith("invalid_argument")]
    assert len(errs) == 2 and "State is not the right size" in errs[0] and "VoxelBackboneValidityChecker" in errs[1]
    # edges through the shim: bisection, last_valid, discrete, caches
    ea, eb = [[0, 0, 0], [8, 3, 1], [1.0, 2.0, 0.5]], [[2.5, 9.0, 4.0], [8.5, 3.5, 1.2], [1.5, 2.2, 0.4]]
    edges = [l.split() for l in out if l.startswith("edge ")]
    dbatch = [l.split() for l in out if l.startswith("dbatch")]
    ecache = [l.split() for l in out if l.startswith("ecache")]
    vcache = [l.split() for l in out if l.startswith("vcache")]
    assert len(edges) == len(dbatch) == len(ecache) == 3 and len(vcache) == 4
    for a, b, row, drow, crow in zip(ea, eb, edges, dbatch, ecache):
        w = orc.check_motion(orb, og, a, b, want_swept=True)
        wu = orc.check_motion_until_invalid(orb, og, a, b)
        wd = orc.check_motion_discrete(orb, og, a, b, until_invalid=True)
        assert int(row[1]) == int(w["valid"]) and int(row[2]) == int(wu["is_fully_valid"]) and float(row[3]) == wu["last_valid_t"]
        assert float(row[4]) == a[0] + (b[0] - a[0]) * wu["last_valid_t"]
        assert int(row[5]) == int(wd["is_fully_valid"]) and float(row[6]) == wd["last_valid_t"]
        assert [int(drow[1]), int(drow[2])] == [int(wd["is_fully_valid"]), wd["n_fk"]]
        assert int(crow[1]) == int(w["is_fully_valid"])
        if w["is_fully_valid"]:
            assert int(crow[2]) == w["swept"].nblocks() and int(crow[3]) == int(og.collides(w["swept"]))
    for s, row in zip(states, vcache):
        ok, _, fl = orc.is_valid_state(orb, og, s)
        assert int(row[1]) == int((fl & 7) == 7) and (not int(row[1]) or int(row[3]) == int(not (fl & 8)))
    # obstacle edits and k-NN through the shim
    g2 = helpers.oracle_grid(orc, vox)
    g2.blocks()[...] = og.blocks()
    g2.add_sphere([0.05, 0.05, 0.1], 0.02); g2.add_sphere([-0.1, 0.0, 0.05], 0.01)
    g2.dilate_sphere(0.004); g2.remove_interior(True)
    ed = [l.split() for l in out if l.startswith("edited")][0]
    assert int(ed[1]) == g2.ncells() and int(ed[2]) == int(g2.cell(166, 166, 179)) and int(ed[3]) == int(g2.cell(128, 128, 128))
    knn = [l.split() for l in out if l.startswith("knn")]
    S = np.array(states, float)
    D = np.linalg.norm(S[:, None, :] - S[None, :, :], axis=2)
    for i, row in enumerate(knn):
        order = np.argsort(D[i], kind="stable")
        assert int(row[1]) == i and int(row[2]) == order[1] and abs(float(row[3]) - D[i, order[1]]) < 1e-12
    sph = [int(l.split()[1]) for l in out if l.startswith("spheres")]
    assert sph == [int(orc.is_valid_state_spheres(orb, og, s)[0]) for s in states]
    # cached roadmap through the shim: same sets as the pairwise form, queries equal the oracle's sequential loop
    redges = np.array([[0, 1], [1, 2], [2, 3], [3, 0], [0, 2]])
    S4 = np.array(states, float)
    iec = [l.split() for l in out if l.startswith("iecache")]
    assert len(iec) == 5
    vcs, ecs = [], []
    for (a_, b_), row in zip(redges, iec):
        w = orc.check_motion(orb, og, S4[a_], S4[b_], want_swept=True)
        assert int(row[1]) == int(w["is_fully_valid"]) and (not w["is_fully_valid"] or int(row[2]) == w["swept"].nblocks())
        ecs.append(w["swept"].export_blocks() if w["is_fully_valid"] else (np.zeros(0, np.uint32), np.zeros(0, np.uint64)))
    con = [l.split() for l in out if l.startswith("connect")]
    assert len(con) == 5
    for (a_, b_), row, irow in zip(redges, con, iec):
        ok = orc.check_motion(orb, og, S4[a_], S4[b_])["valid"]
        assert int(row[1]) == int(ok) and int(row[2]) == (int(irow[2]) if ok else 0)
    ref = orc.Grid(256, vox.limits())
    shape_ok = []
    for s in states:
        ok, _, fl = orc.is_valid_state(orb, og, s)
        shape_ok.append((fl & 7) == 7)
        g_ = ref.empty_copy()
        if shape_ok[-1]:
            g_.add_piecewise_line(orb.shape(s)["p"])
        vcs.append(g_.export_blocks() if shape_ok[-1] else (np.zeros(0, np.uint32), np.zeros(0, np.uint64)))
    csr = lambda items, present: dict(offsets=np.concatenate([[0], np.cumsum([len(i[0]) for i in items])]),
                                      block_ids=np.concatenate([i[0] for i in items]), masks=np.concatenate([i[1] for i in items]),
                                      present=np.array(present, bool))
    e_present = [bool(int(r[1])) for r in iec]
    orm = orc.Roadmap(orb, S4, redges, None, csr(vcs, shape_ok), csr(ecs, e_present))
    queries = [l.split() for l in out if l.startswith("query")]
    assert len(queries) == 3
    code = {-2: 2, -3: 3, 0: 1}
    for (s_, g_), row in zip(((0, 2), (0, 0), (2, 3)), queries):
        w = orm.query(og, s_, g_)
        assert int(row[1]) == (0 if w["n"] > 0 else code[w["n"]])
        if w["n"] > 0:
            assert float(row[2]) == w["cost"] and [int(v) for v in row[3:]] == list(w["path"])
    idx = [l.split() for l in out if l.startswith("indexed")]
    assert len(idx) == 3 and all(r[1] == r[2] for r in idx) and [int(r[1]) for r in idx] == [int(e[1]) for e in edges]


def _knn_edge_set(states, k, rows):
    """The connection loop on the host: for every vertex of `rows` its k nearest among all states (itself included, as nearestK
    on a structure that already holds it), as the undirected edge set ordered by (lo, hi)."""
    pairs = set()
    for v in rows:
        d = states - states[v]
        s = d[:, 0] * d[:, 0]
        for j in range(1, states.shape[1]):                # the engine's left-to-right sum
            s = s + d[:, j] * d[:, j]
        for n in np.argsort(np.sqrt(s), kind="stable")[:k]:
            if n != v:
                pairs.add((min(v, int(n)), max(v, int(n))))
    return np.array(sorted(pairs), dtype=np.int32).reshape(-1, 2)


@pytest.mark.gpu
def test_shim_builds_and_queries_a_roadmap_like_create_roadmap(tmp_path, irt, orc, helpers):
    """tests/cpp/shim_roadmap_test.cpp drives motion_planning::VoxelCachedLazyPRM of the C++ shim the way apps/create_roadmap.cpp
    drives the reference's (createRoadmap with option flags, a growing second call, precomputeValidity, clearDisconnectedVertices,
    solveWithRoadmap, a lazy roadmap); every stage is checked against the oracle: vertex sets = the first candidates of the
    sequence that pass the option's test under the oracle's verdicts, candidate edges = the k-nearest connection loop, edge
    verdicts and FK counts = the oracle's checkMotion, block lists = the oracle's voxel sets, queries = the oracle's lazy loop."""
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import connected_components, dijkstra
    exe = _build(tmp_path, irt, "shim_roadmap_test")
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    grid_file, out = tmp_path / "grid.u64", tmp_path / "out"
    np.ascontiguousarray(vox.blocks, dtype=np.uint64).tofile(grid_file)
    out.mkdir()
    assert "roadmap stages written" in subprocess.check_output([exe, str(grid_file), str(out)], text=True)
    S, SEED = 4, 11
    ld = lambda name, dt: np.fromfile(out / name, dtype=dt)

    def graph(tag):
        c = lambda t: dict(offsets=ld("%s_%s_off.i64" % (tag, t), np.int64), block_ids=ld("%s_%s_ids.u32" % (tag, t), np.uint32),
                           masks=ld("%s_%s_masks.u64" % (tag, t), np.uint64), present=ld("%s_%s_usable.u8" % (tag, t), np.uint8).astype(bool))
        return dict(states=ld(tag + "_states.f64", np.float64).reshape(-1, S), tips=ld(tag + "_tips.f64", np.float64).reshape(-1, 3),
                    edges=ld(tag + "_edges.i32", np.int32).reshape(-1, 2), vc=c("vc"), ec=c("ec"),
                    vstat=ld(tag + "_vstat.u8", np.uint8), estat=ld(tag + "_estat.u8", np.uint8))

    def report(tag):
        meta = ld(tag + "_meta.i64", np.int64)
        return dict(cand=ld(tag + "_cand.i32", np.int32).reshape(-1, 2), acc=ld(tag + "_acc.u8", np.uint8).astype(bool),
                    nfk=ld(tag + "_nfk.i32", np.int32), cidx=ld(tag + "_cidx.i64", np.int64), tried=int(meta[0]), k=int(meta[1]))

    orb, og = helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox)
    orb_omp, omp = helpers.oracle_robot(orc, robot, lib="omp"), orc.omp_lib()
    item = lambda c, i: (c["block_ids"][c["offsets"][i]:c["offsets"][i + 1]], c["masks"][c["offsets"][i]:c["offsets"][i + 1]])
    rng = np.random.default_rng(4)

    def vertex_blocks(state):
        g = og.empty_copy()
        g.add_piecewise_line(orb.shape(state)["p"])
        return g.export_blocks()

    # ---- stage A: createRoadmap(2000, ValidateVertices | ValidateEdges), KBoundedStrategy(8) ----
    A, ra = graph("A"), report("A")
    cand = irt.distributed.candidate_states(robot, SEED, 0, ra["tried"])
    ok, tips, _ = orc.validate_batch(orb_omp, og, cand, nthreads=0, lib=omp)
    idx = np.flatnonzero(ok)[:2000]
    assert len(idx) == 2000 and idx[-1] + 1 == ra["tried"] and np.array_equal(ra["cidx"], idx)
    assert np.array_equal(A["states"], cand[idx]) and np.abs(A["tips"] - tips[idx]).max() <= 1e-9
    assert ra["k"] == 8 and np.array_equal(ra["cand"], _knn_edge_set(A["states"], 8, range(2000)))
    want, wn, _ = orc.check_motion_batch(orb_omp, og, A["states"][ra["cand"][:, 0]], A["states"][ra["cand"][:, 1]], nthreads=0, lib=omp)
    assert np.array_equal(ra["acc"], want) and np.array_equal(ra["nfk"][want], wn[want]) and 0.3 < want.mean() < 1.0
    assert np.array_equal(A["edges"], ra["cand"][want]) and (A["vstat"] == 1).all() and (A["estat"] == 1).all()
    assert A["vc"]["present"].all() and A["ec"]["present"].all() and len(A["ec"]["offsets"]) == len(A["edges"]) + 1
    for v in rng.choice(2000, 40, replace=False):
        ids, masks = vertex_blocks(A["states"][v])
        assert np.array_equal(item(A["vc"], v)[0], ids) and np.array_equal(item(A["vc"], v)[1], masks)
    for e in rng.choice(len(A["edges"]), 40, replace=False):
        w = orc.check_motion(orb, og, A["states"][A["edges"][e, 0]], A["states"][A["edges"][e, 1]], want_swept=True)
        ids, masks = w["swept"].export_blocks()
        assert w["valid"] and np.array_equal(item(A["ec"], e)[0], ids) and np.array_equal(item(A["ec"], e)[1], masks)

    # ---- stage B: createRoadmap(2300, VoxelizeVertices | VoxelizeEdges): shape checks only, the sequence continues ----
    B, rb = graph("B"), report("B")
    candB = irt.distributed.candidate_states(robot, SEED, ra["tried"], rb["tried"])
    shape_ok = np.array([(orc.is_valid_state(orb, og, s_)[2] & 7) == 7 for s_ in candB])
    idxB = np.flatnonzero(shape_ok)[:300]
    assert len(idxB) == 300 and idxB[-1] + 1 == rb["tried"] and np.array_equal(rb["cidx"], ra["tried"] + idxB)
    assert np.array_equal(B["states"][:2000], A["states"]) and np.array_equal(B["states"][2000:], candB[idxB])
    assert (B["vstat"][:2000] == 1).all() and (B["vstat"][2000:] == 0).all() and B["vc"]["present"].all()
    assert np.array_equal(rb["cand"], _knn_edge_set(B["states"], 8, range(2000, 2300))) and (rb["cand"][:, 1] >= 2000).all()
    wb = [orc.check_motion(orb, og, B["states"][a_], B["states"][b_]) for a_, b_ in rb["cand"]]
    fully = np.array([w["is_fully_valid"] for w in wb])
    assert np.array_equal(rb["acc"], fully) and np.array_equal(rb["nfk"][fully], np.array([w["n_fk"] for w in wb])[fully])
    assert np.array_equal(B["edges"], np.concatenate([A["edges"], rb["cand"][fully]]))
    assert (B["estat"][:len(A["edges"])] == 1).all() and (B["estat"][len(A["edges"]):] == 0).all()
    new_valid = np.array([w["valid"] for w in wb])[fully]
    assert not new_valid.all()                                  # some new edges collide: stage C has something to remove

    # ---- stage C: precomputeValidity removes the colliding new milestones (with their edges) and the colliding new edges ----
    Cg = graph("C")
    v_ok = np.ones(2300, bool)
    v_ok[2000:] = orc.validate_batch(orb_omp, og, B["states"][2000:], nthreads=0, lib=omp)[0]
    assert not v_ok.all()
    renum = np.cumsum(v_ok) - 1
    e_ok = v_ok[B["edges"][:, 0]] & v_ok[B["edges"][:, 1]]
    e_ok[len(A["edges"]):] &= new_valid
    assert np.array_equal(Cg["states"], B["states"][v_ok]) and np.array_equal(Cg["edges"], renum[B["edges"][e_ok]])
    assert (Cg["vstat"] == 1).all() and (Cg["estat"] == 1).all() and Cg["ec"]["present"].all()
    keep_e = np.flatnonzero(e_ok)
    for j in rng.choice(len(keep_e), 40, replace=False):
        assert np.array_equal(item(Cg["ec"], j)[0], item(B["ec"], keep_e[j])[0]) and np.array_equal(item(Cg["ec"], j)[1], item(B["ec"], keep_e[j])[1])

    # ---- stage D: clearDisconnectedVertices keeps the largest component ----
    Dg = graph("D")
    V = len(Cg["states"])
    adj = coo_matrix((np.ones(len(Cg["edges"])), (Cg["edges"][:, 0], Cg["edges"][:, 1])), shape=(V, V))
    _, lab = connected_components(adj, directed=False)
    big = lab == np.argmax(np.bincount(lab))
    rn = np.cumsum(big) - 1
    eb = big[Cg["edges"][:, 0]]
    assert np.array_equal(Dg["states"], Cg["states"][big]) and np.array_equal(Dg["edges"], rn[Cg["edges"][eb]])

    # ---- stage E: 300 queries on the finished roadmap: all solved, optimal, along edges, nothing tested again ----
    st, go = ld("E_starts.i32", np.int32), ld("E_goals.i32", np.int32)
    status, cost, pv, po = ld("E_status.i32", np.int32), ld("E_cost.f64", np.float64), ld("E_paths.i32", np.int32), ld("E_poff.i64", np.int64)
    assert (status == 0).all() and ld("E_stats.i64", np.int64)[1] == 0
    Vd = len(Dg["states"])
    wts = np.sqrt(((Dg["states"][Dg["edges"][:, 0]] - Dg["states"][Dg["edges"][:, 1]]) ** 2).sum(axis=1))
    gD = coo_matrix((wts, (Dg["edges"][:, 0], Dg["edges"][:, 1])), shape=(Vd, Vd)).tocsr()
    us = np.unique(st)
    dist = dijkstra(gD, directed=False, indices=us)
    eset = set(map(tuple, Dg["edges"].tolist()))
    for q in range(len(st)):
        p_ = pv[po[q]:po[q + 1]]
        assert p_[0] == st[q] and p_[-1] == go[q]
        assert all((min(a_, b_), max(a_, b_)) in eset for a_, b_ in zip(p_[:-1], p_[1:]))
        assert abs(cost[q] - dist[np.searchsorted(us, st[q]), go[q]]) <= 1e-9

    # ---- stage F: a lazy roadmap (no option, PRM* strategy): unchecked candidates, every candidate edge; the first query voxelises ----
    F0, rf, Fg = graph("F0"), report("F0"), graph("F")
    candF = irt.distributed.candidate_states(robot, SEED, 0, 600)
    kstar = int(np.ceil((np.e + np.e / 4) * np.log(600)))
    assert rf["tried"] == 600 and rf["k"] == kstar and np.array_equal(F0["states"], candF)
    assert not F0["vc"]["present"].any() and not F0["ec"]["present"].any() and np.isnan(F0["tips"]).all()
    assert np.array_equal(F0["edges"], _knn_edge_set(candF, kstar, range(600))) and rf["acc"].all()
    shapeF = np.array([(orc.is_valid_state(orb, og, s_)[2] & 7) == 7 for s_ in candF])
    assert np.array_equal(Fg["vc"]["present"], shapeF) and (Fg["vstat"][~shapeF] == 2).all()
    assert np.array_equal(Fg["edges"], F0["edges"]) and not shapeF.all()
    orm = orc.Roadmap(orb, candF, Fg["edges"], None, Fg["vc"], Fg["ec"])
    stF, goF = ld("F_starts.i32", np.int32), ld("F_goals.i32", np.int32)
    statusF, costF, pvF, poF = ld("F_status.i32", np.int32), ld("F_cost.f64", np.float64), ld("F_paths.i32", np.int32), ld("F_poff.i64", np.int64)
    code = {-2: 2, -3: 3, 0: 1}
    seen = set()
    for q in range(len(stF)):
        w = orm.query(og, stF[q], goF[q])
        assert statusF[q] == (0 if w["n"] > 0 else code[w["n"]])
        seen.add(int(statusF[q]))
        if w["n"] > 0:
            assert costF[q] == w["cost"] and np.array_equal(pvF[poF[q]:poF[q + 1]], w["path"])
    assert 0 in seen and len(seen) >= 2
    for e in rng.choice(len(Fg["edges"]), 30, replace=False):
        w = orc.check_motion(orb, og, candF[Fg["edges"][e, 0]], candF[Fg["edges"][e, 1]], want_swept=True)
        assert bool(Fg["ec"]["present"][e]) == w["is_fully_valid"]
        if w["is_fully_valid"]:
            assert np.array_equal(item(Fg["ec"], e)[0], w["swept"].export_blocks()[0])
