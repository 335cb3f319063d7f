"""The graph searches of the query loop on the device (csrc/search_kernel.hpp: roadmap_astar, one wave per query) against the host
threads' A* -- the statement of astarSearch / constructSolution (motion-planning/VoxelCachedLazyPRM.cpp:2689-2771, 2950-2976) that
tests/test_gpu_lazy_prm.py pins to the oracle: same statuses, costs, paths, validity bytes, rounds and items checked, whichever side
searches.  With one vertex per step (TENDON_HIP_SEARCH_K=1) the kernel is that search statement for statement and the expansions are
counted equal too; with its default of six per step it expands a few vertices the host would not have (the answers do not move)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _strictly_lazy(monkeypatch):
    """Rounds, items tested and validity bytes are compared between schedules here, so the loop stays strictly lazy
    (TENDON_HIP_LAZY_ONLY=1); the last test lifts it: by default the loop turns eager once enough queries are still open after a
    round (a rule on counts, so that path is reproducible too)."""
    monkeypatch.setenv("TENDON_HIP_LAZY_ONLY", "1")


def _prm(irt, n_vertices, k, seed, n_new_spheres=76):
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=seed)
    states, _ = rb.sample_valid_vertices(n_vertices, batch=8192)
    edges = rb.knn_edges_gpu(states, k)
    valid, _ = rb.validate_edges(states, edges)
    edges = edges[valid]
    prm = irt.VoxelCachedLazyPRM(chk, states, edges)
    prm.set_caches(rb.vertex_caches(states), rb.edge_caches(states, edges))
    new_vox, _ = W.reach_environment(seed=7, n_spheres=n_new_spheres)
    prm.set_obstacles(new_vox)
    return prm, states


def _solve(prm, starts, goals, eager):
    prm.clearValidity()
    if eager:
        prm.revalidate()
    out = prm.solveWithRoadmap(starts, goals)
    v, e = prm.validity()
    return out, dict(prm.stats), v, e


def _same(a, b, expansions=True):
    (ra, sa, va, ea), (rb, sb, vb, eb) = a, b
    assert np.array_equal(ra["status"], rb["status"]) and np.array_equal(ra["cost"], rb["cost"])
    assert np.array_equal(ra["path_offsets"], rb["path_offsets"]) and np.array_equal(ra["path_vertices"], rb["path_vertices"])
    assert np.array_equal(va, vb) and np.array_equal(ea, eb)
    assert sa["rounds"] == sb["rounds"] and sa["items_checked"] == sb["items_checked"] and sa["astar_runs"] == sb["astar_runs"]
    if expansions:
        assert sa["expanded"] == sb["expanded"], (sa, sb)


@pytest.mark.parametrize("landmarks", [16, 0, 6])
def test_device_searches_equal_the_host_searches(irt, monkeypatch, landmarks):
    prm, states = _prm(irt, 2500, 6, seed=21)
    prm.prepare(landmarks)
    rng = np.random.default_rng(4)
    nq = 700
    starts, goals = rng.integers(0, len(states), nq), rng.integers(0, len(states), nq)
    goals[:5] = starts[:5]
    for eager in (False, True):
        monkeypatch.setenv("TENDON_HIP_SEARCH", "host")
        ref = _solve(prm, starts, goals, eager)
        assert (ref[0]["status"] == 0).sum() > 0.5 * nq and (ref[0]["status"] != 0).any()
        monkeypatch.setenv("TENDON_HIP_SEARCH", "device")          # every round on the device, no expansion budget
        monkeypatch.setenv("TENDON_HIP_SEARCH_K", "1")
        _same(ref, _solve(prm, starts, goals, eager))
        monkeypatch.delenv("TENDON_HIP_SEARCH_K")
        out = _solve(prm, starts, goals, eager)
        _same(ref, out, expansions=False)
        # (searches of ~50 expansions on this small roadmap: taking six vertices per step, the last steps expand vertices the host
        # never reaches; on the 10^5-vertex roadmap, searches of ~1000 expansions, the excess is 1 - 2 %)
        assert ref[1]["expanded"] <= out[1]["expanded"] < 3 * ref[1]["expanded"], (ref[1], out[1])
        assert prm.search_stats["device"] == ref[1]["astar_runs"] and prm.search_stats["handed_back"] == 0
        monkeypatch.delenv("TENDON_HIP_SEARCH")                    # the default: large rounds shared with the host threads
        _same(ref, _solve(prm, starts, goals, eager), expansions=False)
        ss = prm.search_stats                                      # (the host's share follows the clock: anything up to 8 %)
        assert ss["device"] > 0.8 * (ref[1]["astar_runs"] - ss["answered_by_components"]) - 50 and ss["host_meanwhile"] > 0, ss
        # a budget that most searches exceed (they are handed back to the host threads) and a large host share: the expansions
        # the kernel spent before giving up count as well, so only the answers are compared
        monkeypatch.setenv("TENDON_HIP_SEARCH_BUDGET", "40")
        monkeypatch.setenv("TENDON_HIP_SEARCH_HOST_SHARE", "25")
        _same(ref, _solve(prm, starts, goals, eager), expansions=False)
        assert prm.search_stats["handed_back"] > 100 and prm.search_stats["host_meanwhile"] > 100, prm.search_stats
        monkeypatch.delenv("TENDON_HIP_SEARCH_BUDGET")
        monkeypatch.delenv("TENDON_HIP_SEARCH_HOST_SHARE")


def test_device_searches_with_open_lists_beyond_the_lds_part(irt, monkeypatch):
    """20 000 vertices, no landmark bounds: the open list of a search outgrows the kernel's LDS part (1024 entries), so the threshold
    drops (entries move to the list in HBM) and rises again (they come back) many times per search."""
    prm, states = _prm(irt, 20000, 8, seed=5, n_new_spheres=70)
    prm.prepare(0)
    rng = np.random.default_rng(8)
    nq = 600
    order = np.argsort(states[:, 0] + states[:, 1])                # end points from opposite corners of the sampled box: long searches
    starts, goals = rng.choice(order[:2000], nq).astype(np.int64), rng.choice(order[-2000:], nq).astype(np.int64)
    monkeypatch.setenv("TENDON_HIP_SEARCH", "host")
    ref = _solve(prm, starts, goals, True)
    assert prm.search_stats["device"] == 0
    monkeypatch.setenv("TENDON_HIP_SEARCH", "device")
    monkeypatch.setenv("TENDON_HIP_SEARCH_K", "1")
    _same(ref, _solve(prm, starts, goals, True))
    monkeypatch.delenv("TENDON_HIP_SEARCH_K")
    _same(ref, _solve(prm, starts, goals, True), expansions=False)
    ss = prm.search_stats
    assert ss["device"] == ref[1]["astar_runs"] and ss["handed_back"] == 0 and ss["list_moves"] > nq, ss   # ... and it did happen
    monkeypatch.setenv("TENDON_HIP_SEARCH", "host")
    ref = _solve(prm, starts, goals, False)
    monkeypatch.setenv("TENDON_HIP_SEARCH", "device")
    _same(ref, _solve(prm, starts, goals, False), expansions=False)


@pytest.mark.parametrize("lc0,pool", [("8", None), ("8", "6,1,0"), (None, None)])
def test_device_searches_on_wide_vertices_small_tables_and_an_exhausted_pool(irt, monkeypatch, lc0, pool):
    """14 neighbours per vertex: a third of the vertices have more arcs than one adjacency row of the kernel holds (16) and chain a
    second row.  With tables of 256 records per search every search outgrows its own table and moves -- rehashing -- into the shared
    pool's larger ones, some of them twice; with a pool of 6 + 1 tables most find none free and are handed back to the host threads.
    Same statuses, costs, paths, validity, rounds as the host searches every time; with one vertex per step the same expansions."""
    if lc0 is not None:
        monkeypatch.setenv("TENDON_HIP_SEARCH_LC0", lc0)
    if pool is not None:
        monkeypatch.setenv("TENDON_HIP_SEARCH_POOL", pool)
    prm, states = _prm(irt, 8000, 15, seed=13, n_new_spheres=72)
    deg = np.bincount(prm.edges.ravel(), minlength=len(states))
    assert (deg > 16).mean() > 0.1 and deg.max() > 20
    prm.prepare(4)                                                 # weak bounds: searches of a few hundred expansions
    rng = np.random.default_rng(3)
    nq = 600
    starts, goals = rng.integers(0, len(states), nq), rng.integers(0, len(states), nq)
    for eager in (True, False):
        monkeypatch.setenv("TENDON_HIP_SEARCH", "host")
        ref = _solve(prm, starts, goals, eager)
        monkeypatch.setenv("TENDON_HIP_SEARCH", "device")
        if pool is None:
            monkeypatch.setenv("TENDON_HIP_SEARCH_K", "1")
            _same(ref, _solve(prm, starts, goals, eager))
            monkeypatch.delenv("TENDON_HIP_SEARCH_K")
        out = _solve(prm, starts, goals, eager)
        _same(ref, out, expansions=False)
        ss = prm.search_stats
        if lc0 is None:
            assert ss["handed_back"] == 0 and ss["table_growths"] == 0, ss
        elif pool is None:
            assert ss["handed_back"] == 0 and ss["table_growths"] > nq // 2, ss
        else:
            assert ss["handed_back"] > nq // 4 and ss["table_growths"] > 0 and ss["device"] > 0, ss


def test_device_searches_on_a_600k_vertex_roadmap(irt, monkeypatch):
    """Config 4's roadmap size (6 x 10^5 vertices, 3.4 x 10^6 candidate edges of the 10-nearest connection loop, taken as valid: the
    searches need the graph and validity bytes only -- tr_roadmap_set_validity): the searches' state on the device is what it is
    at any size (a table per search in flight + the shared pool; round 4's 32 B x V per slot left 671 of 3 072 slots here), every
    slot is resident, and 3 000 queries come back with the host threads' statuses, costs and paths -- all on the device, and through
    the default shared schedule."""
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=3)
    states, _ = rb.sample_valid_vertices(600000)
    edges = rb.knn_edges_gpu(states, 11)
    assert len(edges) > 3000000
    prm = irt.VoxelCachedLazyPRM(chk, states, edges)
    prm.set_validity(np.ones(len(states), np.uint8), np.ones(len(edges), np.uint8))
    prm.prepare(16)
    rng = np.random.default_rng(6)
    nq = 3000
    starts, goals = rng.integers(0, len(states), nq), rng.integers(0, len(states), nq)
    monkeypatch.setenv("TENDON_HIP_SEARCH", "host")
    ref = prm.solveWithRoadmap(starts, goals)
    assert (ref["status"] == 0).mean() > 0.99 and prm.stats["items_checked"] == 0
    for mode in ("device", None):
        if mode is None:
            monkeypatch.delenv("TENDON_HIP_SEARCH")
        else:
            monkeypatch.setenv("TENDON_HIP_SEARCH", mode)
        out = prm.solveWithRoadmap(starts, goals)
        assert np.array_equal(ref["status"], out["status"]) and np.array_equal(ref["cost"], out["cost"])
        assert np.array_equal(ref["path_offsets"], out["path_offsets"]) and np.array_equal(ref["path_vertices"], out["path_vertices"])
        ss = prm.search_stats
        if mode == "device":
            assert ss["device"] + ss["handed_back"] == nq - int((starts == goals).sum()) and ss["handed_back"] < nq // 10 and ss["table_growths"] > 0, ss
        else:
            assert ss["device"] > 0.8 * nq, ss
        assert prm.search_profile["launches"] >= 1 and prm.search_profile["kernel_ms"] > 0


def test_roadmap_with_parallel_edges_stays_on_the_host(irt, monkeypatch):
    """Two edges between the same pair of vertices: the kernel relaxes a vertex's arcs in parallel lanes, so such a roadmap is
    searched on the host whatever the switch says -- and gives the answers it always gave."""
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=3)
    states, _ = rb.sample_valid_vertices(300, batch=4096)
    edges = rb.knn_edges_gpu(states, 5)
    valid, _ = rb.validate_edges(states, edges)
    edges = edges[valid]
    doubled = np.concatenate([edges, edges[:7][:, ::-1]])
    rng = np.random.default_rng(2)
    starts, goals = rng.integers(0, len(states), 64), rng.integers(0, len(states), 64)
    outs = []
    for ed in (edges, doubled):
        prm = irt.VoxelCachedLazyPRM(chk, states, ed)
        prm.set_caches(rb.vertex_caches(states), rb.edge_caches(states, ed))
        prm.set_obstacles(vox)
        monkeypatch.setenv("TENDON_HIP_SEARCH", "device")
        outs.append(prm.solveWithRoadmap(starts, goals))
    assert np.array_equal(outs[0]["status"], outs[1]["status"]) and np.array_equal(outs[0]["cost"], outs[1]["cost"])
    assert np.array_equal(outs[0]["path_vertices"], outs[1]["path_vertices"])


def test_unreachable_goals_are_answered_by_component_labels(irt, monkeypatch):
    """A cluttered environment cuts the roadmap into pieces: queries across pieces are answered from the component labels of the
    round (the reference's solutionComponent test) instead of a search that walks the start's whole component -- same statuses,
    costs, paths and validity as with the labels switched off, on the host threads and on the device; by default the labels are
    computed from the first solve on in which a search walked a component in vain."""
    prm, states = _prm(irt, 6000, 6, seed=9, n_new_spheres=160)
    prm.prepare(16)
    rng = np.random.default_rng(12)
    nq = 800
    starts, goals = rng.integers(0, len(states), nq), rng.integers(0, len(states), nq)

    def answers_equal(want, out):
        (ra, sa, va, ea), (rb, sb, vb, eb) = want, out
        assert np.array_equal(ra["status"], rb["status"]) and np.array_equal(ra["cost"], rb["cost"])
        assert np.array_equal(ra["path_vertices"], rb["path_vertices"]) and np.array_equal(va, vb) and np.array_equal(ea, eb)
        assert sa["rounds"] == sb["rounds"] and sa["items_checked"] == sb["items_checked"]

    monkeypatch.setenv("TENDON_HIP_SEARCH", "host")
    monkeypatch.setenv("TENDON_HIP_COMPONENTS", "0")
    ref = {eager: _solve(prm, starts, goals, eager) for eager in (True, False)}
    n_nopath = int((ref[True][0]["status"] == 1).sum())
    assert n_nopath > 20 and (ref[True][0]["status"] == 0).sum() > 100, np.bincount(ref[True][0]["status"])
    assert prm.search_stats["answered_by_components"] == 0
    monkeypatch.setenv("TENDON_HIP_COMPONENTS", "1")               # labels in every round
    for mode in ("host", "device", None):
        if mode is None:
            monkeypatch.delenv("TENDON_HIP_SEARCH")
        else:
            monkeypatch.setenv("TENDON_HIP_SEARCH", mode)
        for eager in (True, False):
            out = _solve(prm, starts, goals, eager)
            answers_equal(ref[eager], out)
            if eager:
                assert prm.search_stats["answered_by_components"] == n_nopath, (prm.search_stats, n_nopath)
                assert out[1]["expanded"] < ref[eager][1]["expanded"]
    # the default: the first solve finds out (searches that walk thousands of vertices in vain), the second one uses labels
    monkeypatch.delenv("TENDON_HIP_COMPONENTS")
    monkeypatch.setenv("TENDON_HIP_SEARCH", "host")
    prm2, _ = _prm(irt, 6000, 6, seed=9, n_new_spheres=160)
    prm2.prepare(16)
    first = _solve(prm2, starts, goals, True)
    answers_equal(ref[True], first)
    if first[1]["expanded"] > 2000 * 20:                           # (it did walk in vain)
        second = _solve(prm2, starts, goals, True)
        answers_equal(ref[True], second)
        assert prm2.search_stats["answered_by_components"] == n_nopath and second[1]["expanded"] < first[1]["expanded"]


def test_lazy_loop_turns_eager_when_queries_keep_coming_back(irt, monkeypatch):
    """Cluttered environment, validity unknown: the lazy loop alone needs many rounds (every round finds the open queries new
    candidate paths through items nobody looked at); by default tr_roadmap_solve tests every cached set in one launch as soon as
    that is cheaper than another round of searches.  Same answers, fewer rounds, and afterwards no item is unknown."""
    prm, states = _prm(irt, 6000, 6, seed=9, n_new_spheres=160)
    prm.prepare(16)
    rng = np.random.default_rng(12)
    nq = 800
    starts, goals = rng.integers(0, len(states), nq), rng.integers(0, len(states), nq)
    lazy = _solve(prm, starts, goals, False)                       # (strictly lazy: the fixture)
    assert (lazy[2] == 0).any() and lazy[1]["rounds"] >= 4, lazy[1]
    monkeypatch.delenv("TENDON_HIP_LAZY_ONLY")
    out = _solve(prm, starts, goals, False)
    assert np.array_equal(lazy[0]["status"], out[0]["status"]) and np.array_equal(lazy[0]["cost"], out[0]["cost"])
    assert np.array_equal(lazy[0]["path_vertices"], out[0]["path_vertices"])
    assert out[1]["rounds"] < lazy[1]["rounds"] and not (out[2] == 0).any() and not (out[3] == 0).any(), (lazy[1], out[1])
    known = lazy[2] > 0
    assert np.array_equal(lazy[2][known], out[2][known])             # what the lazy loop knew, it knew right
    # the switch reads counts, not clocks: the default path leaves the same rounds, items tested and validity bytes run after run
    # and whichever side -- host threads, kernel, both -- runs the searches
    assert out[1]["rounds"] <= 2
    for mode in ("host", "device", None, None):
        if mode is None:
            monkeypatch.delenv("TENDON_HIP_SEARCH", raising=False)
        else:
            monkeypatch.setenv("TENDON_HIP_SEARCH", mode)
        again = _solve(prm, starts, goals, False)
        assert np.array_equal(again[0]["status"], out[0]["status"]) and np.array_equal(again[0]["path_vertices"], out[0]["path_vertices"])
        assert again[1]["rounds"] == out[1]["rounds"] and again[1]["items_checked"] == out[1]["items_checked"], (again[1], out[1])
        assert np.array_equal(again[2], out[2]) and np.array_equal(again[3], out[3])


def test_search_state_follows_the_round_size_and_can_be_released(irt, monkeypatch):
    """The searches' tables are sized by the largest round so far, not by what the device could hold: a 600-query round holds tables
    for 768 searches in flight, a 3 000-query round the device's full set; tr_roadmap_release_search_state hands them back and the
    next round allocates them again -- the answers are the host's every time."""
    import torch

    def _same(a, b, expansions=False):          # (the searches run differ: the component labels answer a few queries on one side only)
        for k in ("status", "cost", "path_offsets", "path_vertices"):
            assert np.array_equal(a[0][k], b[0][k]), k
        assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])

    prm, states = _prm(irt, 2500, 6, seed=21)
    prm.prepare(8)
    rng = np.random.default_rng(9)
    pairs = rng.integers(0, len(states), size=(3000, 2)).astype(np.int32)
    monkeypatch.setenv("TENDON_HIP_SEARCH", "host")
    want_small, want_all = _solve(prm, pairs[:600, 0], pairs[:600, 1], True), _solve(prm, pairs[:, 0], pairs[:, 1], True)
    monkeypatch.setenv("TENDON_HIP_SEARCH", "device")
    assert prm.search_state_bytes() == 0
    got = _solve(prm, pairs[:600, 0], pairs[:600, 1], True)
    _same(got, want_small, expansions=False)
    small = prm.search_state_bytes()
    assert prm.search_stats["device"] > 0 and 100e6 < small < 1.5e9                    # 768 slots x 176 KiB + a pool in proportion
    got = _solve(prm, pairs[:, 0], pairs[:, 1], True)
    _same(got, want_all, expansions=False)
    full = prm.search_state_bytes()
    assert full > 2.5 * small
    free0 = torch.cuda.mem_get_info()[0]
    assert prm.release_search_state() == full and prm.search_state_bytes() == 0
    prm.reserve_search_state(600)                                                        # ahead of a batch: the same tables as the 600-query round left
    assert prm.search_state_bytes() > 0.9 * small - 64e3 and prm.release_search_state() > 0
    assert torch.cuda.mem_get_info()[0] - free0 > 0.9 * full                             # back with the device, not parked
    got = _solve(prm, pairs[:600, 0], pairs[:600, 1], True)
    _same(got, want_small, expansions=False)
    assert prm.search_stats["device"] > 0 and prm.search_state_bytes() == small


def test_two_roadmaps_queried_from_two_host_threads_at_once(irt, monkeypatch):
    """Two planners, each queried from its own Python thread at the same time (the C calls run without the GIL): the library's team of
    host threads serves one call, the other starts threads of its own, both launch their searches on the device -- and each gets the
    answers it gets alone."""
    import threading
    monkeypatch.delenv("TENDON_HIP_LAZY_ONLY", raising=False)
    a, sa = _prm(irt, 2500, 6, seed=21)
    b, sb = _prm(irt, 3000, 7, seed=22)
    for p in (a, b):
        p.prepare(8)
    rng = np.random.default_rng(31)
    qa = rng.integers(0, len(sa), size=(1500, 2)).astype(np.int32)
    qb = rng.integers(0, len(sb), size=(1500, 2)).astype(np.int32)
    alone = [p.solveWithRoadmap(q[:, 0], q[:, 1]) for p, q in ((a, qa), (b, qb))]
    for rep in range(3):
        got, errs = [None, None], []

        def run(i, p, q):
            try:
                p.clearValidity()
                got[i] = p.solveWithRoadmap(q[:, 0], q[:, 1])
            except Exception as e:                      # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=run, args=(0, a, qa)), threading.Thread(target=run, args=(1, b, qb))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for w, g in zip(alone, got):
            for k in ("status", "cost", "path_offsets", "path_vertices"):
                assert np.array_equal(w[k], g[k]), (rep, k)


@pytest.mark.parametrize("form", ["eager", "lazy"])
def test_searches_answered_by_the_device_sweep_equal_the_host_searches(irt, monkeypatch, form):
    """A host search that passes TENDON_HIP_SEARCH_SWEEP expansions is abandoned and answered by relaxing its source's distances over all
    valid arcs at once on the device (tr_roadmap_search_sweeps).  With the bar at 40 expansions most searches of this roadmap go that way
    -- hand-backs of the kernel (a budget of 60) and the host's own share alike: statuses, costs, PATHS, validity and rounds are those of
    plain A* on the host threads, unreachable goals included."""
    prm, states = _prm(irt, 3000, 6, seed=23, n_new_spheres=90)
    prm.prepare(8)
    rng = np.random.default_rng(6)
    pairs = rng.integers(0, len(states), size=(1200, 2)).astype(np.int32)
    eager = form == "eager"
    monkeypatch.setenv("TENDON_HIP_SEARCH", "host")
    monkeypatch.setenv("TENDON_HIP_SEARCH_SWEEP", "0")
    want = _solve(prm, pairs[:, 0], pairs[:, 1], eager)
    assert prm.search_sweeps() == 0 and (want[0]["status"] == 0).sum() > 300 and (want[0]["status"] == 1).sum() > 0
    prm.reserve_search_state(1200)                                       # the rows the sweep reads (a round on the device does the same)
    monkeypatch.setenv("TENDON_HIP_SEARCH_SWEEP", "40")
    def same_answers(got):                   # (the count of searches run differs: once a search has walked a component in vain, labels answer such queries)
        for k in ("status", "cost", "path_offsets", "path_vertices"):
            assert np.array_equal(got[0][k], want[0][k]), k
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
        assert got[1]["rounds"] == want[1]["rounds"] and got[1]["items_checked"] == want[1]["items_checked"]

    got = _solve(prm, pairs[:, 0], pairs[:, 1], eager)
    assert prm.search_sweeps() > 200
    same_answers(got)
    monkeypatch.delenv("TENDON_HIP_SEARCH")                              # the shared schedule: hand-backs over a budget of 60 meet the bar too
    monkeypatch.setenv("TENDON_HIP_SEARCH_BUDGET", "60")
    got = _solve(prm, pairs[:, 0], pairs[:, 1], eager)
    assert prm.search_sweeps() > 50 and prm.search_stats["handed_back"] > 50
    same_answers(got)
