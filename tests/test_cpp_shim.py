"""The C++ host shim (include/tendon_hip_shim.hpp) -- the layer reference-side C++ code would
call -- compiled with g++ against libtendon_hip.so.  CPU: it compiles, links and maps errors to the
reference's exception types.  GPU: its results match the oracle."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "interactive-rate-tendons_amd")


def _build(tmp_path, irt):
    irt.build()
    exe = str(tmp_path / "shim_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "shim_test.cpp"), "-o", exe, "-L", PKG, "-ltendon_hip",
                           "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,-rpath," + PKG, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_shim_compiles_and_maps_errors(tmp_path, irt):
    exe = _build(tmp_path, irt)
    out = subprocess.check_output([exe, "--no-gpu"], text=True)
    assert "caught 3" in out


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
