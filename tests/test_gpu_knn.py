"""GPU k-nearest-neighbours in the reference's state-space metric (tr_knn) against brute-force numpy
with the oracle's compound distance, and the edge set against the host cKDTree path."""
import ctypes as C
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _oracle_dist_matrix(orc, orb, st):
    n = len(st)
    D = np.empty((n, n))
    f = orb.lib.orc_state_distance
    for i in range(n):
        for j in range(n):
            D[i, j] = f(C.byref(orb.c), orc._dp(st[i]), orc._dp(st[j]))
    return D


@pytest.mark.parametrize("rot,ret", [(False, False), (True, False), (True, True)])
def test_knn_matches_oracle_metric(irt, orc, helpers, rot, ret):
    W = irt.workloads
    robot = W.robot_config2()
    robot.enable_rotation, robot.enable_retraction = rot, ret
    st = W.random_states(robot, 300, seed=81)
    idx, dist = robot.engine().knn(st, 9)
    D = _oracle_dist_matrix(orc, helpers.oracle_robot(orc, robot), st)
    want = np.argsort(D, axis=1, kind="stable")[:, :9]
    assert np.array_equal(idx[:, 0], np.arange(300)) and (dist[:, 0] == 0).all()     # self first
    assert np.array_equal(idx, want)
    assert np.abs(dist - np.take_along_axis(D, want, 1)).max() <= 1e-12
    assert (np.diff(dist, axis=1) >= 0).all()
    # bounded strategy: entries beyond max_distance are dropped
    md = float(np.median(dist[:, 4]))
    idx2, dist2 = robot.engine().knn(st, 9, max_distance=md)
    assert np.array_equal(idx2 >= 0, dist <= md) and np.array_equal(idx2[idx2 >= 0], idx[dist <= md])


def test_knn_edges_equal_host_kdtree(irt):
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=16)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=2)
    st = W.random_states(robot, 20000, seed=82)
    t0 = time.perf_counter(); e_gpu = rb.knn_edges_gpu(st, 11); t1 = time.perf_counter()
    e_cpu = rb.knn_edges(st, 10)                       # k counts the vertex itself on the GPU path
    assert np.array_equal(e_gpu, e_cpu)
    print("knn 20k x 4: gpu %.3f s (incl. PCIe + edge dedup), host cKDTree %.3f s" % (t1 - t0, rb.timing["knn"]["seconds"]))


def test_knn_errors(irt):
    e = irt.workloads.robot_config2().engine()
    with pytest.raises(irt.InvalidArgument):
        e.knn(np.zeros((10, 3)), 0)
    with pytest.raises(irt.InvalidArgument):
        e.knn(np.zeros((10, 3)), 500)
    idx, dist = e.knn(np.zeros((0, 3)), 4)
    assert idx.shape == (0, 4)
    idx, dist = e.knn(np.array([[1.0, 2.0, 3.0], [1.0, 2.0, 4.0]]), 4)     # fewer states than k
    assert idx.tolist() == [[0, 1, -1, -1], [1, 0, -1, -1]] and np.isinf(dist[:, 2:]).all()


def test_knn_ties_follow_a_stable_sort_of_the_distances(irt):
    """Neighbour lists are ordered by the distance CompoundStateSpace::distance returns, ties in index order -- across
    the candidate slices the kernel works in, and also where two different squared distances round to the same root."""
    e = irt.workloads.robot_config2().engine()
    rng = np.random.default_rng(5)
    st = rng.integers(0, 7, (3000, 3)).astype(float)          # a small lattice: masses of exact ties and duplicate states
    idx, dist = e.knn(st, 12)
    for lo in range(0, 3000, 500):
        D = np.sqrt(((st[lo:lo + 500, None, :] - st[None, :, :]) ** 2).sum(-1))
        want = np.argsort(D, axis=1, kind="stable")[:, :12]
        assert np.array_equal(idx[lo:lo + 500], want)
        assert np.array_equal(dist[lo:lo + 500], np.take_along_axis(D, want, 1))
    # 2^60 + 256 and 2^60 are different doubles with the same correctly rounded square root, 2^30
    big = np.array([[0.0, 0, 0], [2.0 ** 30, 16.0, 0], [2.0 ** 30, 0, 0], [2.0 ** 30, 0, 16.0]])
    assert np.sqrt(2.0 ** 60 + 256) == 2.0 ** 30
    idx, dist = e.knn(big, 4)
    assert idx[0].tolist() == [0, 1, 2, 3] and dist[0].tolist() == [0.0, 2.0 ** 30, 2.0 ** 30, 2.0 ** 30]


def test_kstar_connection_strategy(irt):
    """og::KStarStrategy: k = ceil((e + e/dim) ln n) -- 40 for the 4-dimensional config-3 space at 10^5 vertices."""
    W = irt.workloads
    e4, e3 = W.robot_config3().engine(), W.robot_config2().engine()
    assert e4.kstar_k(100000) == int(np.ceil((np.e + np.e / 4) * np.log(1e5))) == 40
    assert e3.kstar_k(1000) == int(np.ceil((np.e + np.e / 3) * np.log(1000)))
    with pytest.raises(irt.InvalidArgument):
        e3.kstar_k(0)
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=16)
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=2)
    st = W.random_states(robot, 3000, seed=83)
    k = chk.engine.kstar_k(len(st))
    e = rb.knn_edges_star(st)
    assert np.array_equal(e, rb.knn_edges(st, k)) and len(e) >= len(st) * k // 2


def test_knn_edge_list_on_device_equals_host_dedup(irt):
    """tr_knn_edges: the undirected, duplicate-free, ordered edge set of the k-nearest table, also with the bounded
    strategy's distance limit and a robot with rotation + retraction coordinates."""
    W = irt.workloads
    robot = W.robot_config2()
    robot.enable_rotation, robot.enable_retraction = True, True
    e = robot.engine()
    st = W.random_states(robot, 5000, seed=84)
    for md in (np.inf, 6.0):
        idx, dist = e.knn(st, 8, md)
        src = np.repeat(np.arange(len(st)), 8)
        dst = idx.reshape(-1).astype(np.int64)
        keep = (dst >= 0) & (dst != src)
        key = np.unique(np.minimum(src[keep], dst[keep]) * len(st) + np.maximum(src[keep], dst[keep]))
        want = np.stack([key // len(st), key % len(st)], 1)
        got = e.knn_edges(st, 8, md)
        assert np.array_equal(got, want) and (md == np.inf or len(want) < 7 * len(st) // 2)
    assert e.knn_edges(st[:1], 4).shape == (0, 2)


def test_query_ranges_are_rows_of_the_whole_table(irt):
    """tr_knn_range: any range of queries gives exactly those rows of tr_knn's tables (the candidate slicing differs with
    the number of queries; the merge is order-exact), and tr_knn_table_edges builds tr_knn_edges' edge set from a table
    handed in by the caller -- the two halves of a connection loop spread over several GPUs."""
    W = irt.workloads
    for mk, rot, n in ((W.robot_config3, False, 9000), (W.robot_config2, True, 700)):
        robot = mk()
        robot.enable_rotation = rot
        vox, _ = W.reach_environment(seed=7, n_spheres=8)
        eng = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox).engine
        st = W.random_states(robot, n, seed=71)
        st[5] = st[4]                                           # an exact tie
        idx, dist = eng.knn(st, 9)
        for first, count in ((0, 64), (0, n), (n // 3, n // 2), (n - 1, 1), (130, 1000 if n > 2000 else 100)):
            ri, rd = eng.knn(st, 9, query_range=(first, count))
            assert ri.shape == (count, 9) and np.array_equal(ri, idx[first:first + count]) and np.array_equal(rd, dist[first:first + count])
        assert eng.knn(st, 9, query_range=(7, 0))[0].shape == (0, 9)
        for bad in ((-1, 5), (n - 2, 3)):
            with pytest.raises(irt.OutOfRange):
                eng.knn(st, 9, query_range=bad)
        assert np.array_equal(eng.edges_from_knn(idx), eng.knn_edges(st, 9))
        far = eng.knn(st, 9, max_distance=float(np.median(dist[:, 3])))[0]     # rows with missing neighbours (-1)
        assert (far < 0).any() and np.array_equal(eng.edges_from_knn(far), eng.knn_edges(st, 9, max_distance=float(np.median(dist[:, 3]))))
        with pytest.raises(irt.OutOfRange):
            eng.edges_from_knn(np.full((4, 3), 4, np.int32))


def test_degenerate_inputs_of_the_sorted_search(irt):
    """The neighbour search sorts by the first coordinate and prunes by it: inputs where that coordinate says nothing (all
    equal), where everything ties (identical states), fewer states than k, a single state, and a large set whose first
    coordinate takes only two values -- against a stable argsort of the numpy distance matrix."""
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=8)
    eng = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox).engine
    rng = np.random.default_rng(5)

    def check(st, k):
        idx, dist = eng.knn(st, k)
        D = np.linalg.norm(st[:, None, :] - st[None, :, :], axis=2)
        order = np.argsort(D, axis=1, kind="stable")[:, :k]
        n = len(st)
        kk = min(k, n)
        assert np.array_equal(idx[:, :kk], order[:, :kk]), np.argwhere(idx[:, :kk] != order[:, :kk])[:5]
        assert np.allclose(dist[:, :kk], np.take_along_axis(D, order[:, :kk], 1), rtol=0, atol=1e-15)
        if kk < k:
            assert (idx[:, kk:] == -1).all() and np.isinf(dist[:, kk:]).all()
        edges = eng.knn_edges(st, k)
        want = {(min(i, j), max(i, j)) for i in range(n) for j in order[i, :kk] if i != j}
        assert {tuple(e) for e in edges} == want

    st = rng.uniform(0, 20, (700, 4))
    st[:, 0] = 3.0                                             # the sort coordinate carries no information
    check(st, 9)
    check(np.tile(rng.uniform(0, 20, (1, 4)), (300, 1)), 7)     # identical states: all distances 0, order by index
    check(rng.uniform(0, 20, (5, 4)), 9)                        # fewer states than k
    check(rng.uniform(0, 20, (1, 4)), 3)
    big = rng.uniform(0, 20, (6000, 4))                         # large enough for the seeding pass and candidate slices
    big[:, 0] = np.where(rng.random(6000) < 0.5, 1.0, 19.0)
    big[100:140] = big[99]                                      # a cluster of duplicates
    check(big, 11)


@pytest.mark.parametrize("rot,ret", [(True, False), (False, True), (True, True)])
def test_windowed_search_with_rotation_and_retraction_metrics(irt, orc, helpers, rot, ret):
    """The sorted-coordinate search (seeding pass, windows, candidate slices: n >= 4096) under the compound metrics: the first
    tension still bounds the distance from below, whatever the rotation / retraction terms add.  Sampled query rows against
    a stable argsort of the oracle's distances to ALL states; the whole table against the search with another window."""
    import os
    W = irt.workloads
    robot = W.robot_config2()
    robot.enable_rotation, robot.enable_retraction = rot, ret
    n, k = 6000, 10
    st = W.random_states(robot, n, seed=83)
    st[17] = st[16]                                            # an exact tie
    eng = robot.engine()
    idx, dist = eng.knn(st, k)
    orb = helpers.oracle_robot(orc, robot)
    f = orb.lib.orc_state_distance
    rng = np.random.default_rng(3)
    for q in list(rng.integers(0, n, 24)) + [16, 17]:
        d = np.array([f(C.byref(orb.c), orc._dp(st[q]), orc._dp(st[j])) for j in range(n)])
        want = np.argsort(d, kind="stable")[:k]
        assert np.array_equal(idx[q], want), (q, idx[q], want)
        assert np.abs(dist[q] - d[want]).max() <= 1e-12
    os.environ["TENDON_HIP_KNN_HW_DIV"] = "4"
    try:
        idx2, dist2 = eng.knn(st, k)
    finally:
        os.environ.pop("TENDON_HIP_KNN_HW_DIV")
    assert np.array_equal(idx, idx2) and np.array_equal(dist, dist2)
    md = float(np.median(dist[:, 5]))
    idx3, _ = eng.knn(st, k, max_distance=md)
    assert np.array_equal(idx3 >= 0, dist <= md) and np.array_equal(idx3[idx3 >= 0], idx[dist <= md])


def test_wave_per_query_and_lane_per_query_kernels_give_the_same_tables(irt, monkeypatch):
    """k <= 64 runs a wave per query over a three-key cell grid (knn_wave_query), larger k -- or TENDON_HIP_KNN=lanes -- a lane per
    query over two keys (knn_bruteforce): identical tables, entry for entry, at small k, at the 64 / 65 boundary, with a distance
    bound, with duplicates and with fewer states than k."""
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=8)
    eng = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox).engine
    rng = np.random.default_rng(23)
    st = rng.uniform(0, 20, (9000, 4))
    st[500:540] = st[499]                                       # ties
    small = rng.uniform(0, 20, (40, 4))

    def table(states, k, max_distance=np.inf, lanes=False):
        if lanes:
            monkeypatch.setenv("TENDON_HIP_KNN", "lanes")
        else:
            monkeypatch.delenv("TENDON_HIP_KNN", raising=False)
        return eng.knn(states, k, max_distance)

    for states, k, md in ((st, 11, np.inf), (st, 64, np.inf), (st, 3, 2.5), (small, 64, np.inf), (st[:3000], 33, 6.0)):
        a, b = table(states, k, md), table(states, k, md, lanes=True)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (k, md)
    # k = 65 always takes the lane-per-query kernel; its first 64 columns are the k = 64 table
    monkeypatch.delenv("TENDON_HIP_KNN", raising=False)
    i65, d65 = eng.knn(st[:4000], 65)
    i64, d64 = eng.knn(st[:4000], 64)
    assert np.array_equal(i65[:, :64], i64) and np.array_equal(d65[:, :64], d64)


def test_device_resident_range_and_table_forms(irt):
    """tr_knn_range_dev / tr_knn_table_edges_dev (states, rows and edge list in HBM) give the rows of tr_knn_range and the edge list of
    tr_knn_table_edges -- ranges that are not whole shards, a table with -1 padding entries, a truncating capacity -- and the same errors."""
    import torch
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=16)
    eng = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox).engine
    n, k = 5000, 9
    states = W.random_states(robot, n, seed=77)
    d_states = torch.from_numpy(states).cuda()
    table, _ = eng.knn(states, k)
    d_table = torch.full((n, k), -5, dtype=torch.int32, device="cuda")
    for first, count in ((0, 1700), (1700, 1601), (3301, 1699)):
        rows = torch.full((count, k), -5, dtype=torch.int32, device="cuda")
        eng.knn_range_dev(d_states, n, first, count, k, rows)
        assert np.array_equal(rows.cpu().numpy(), table[first:first + count])
        d_table[first:first + count] = rows
    want = eng.edges_from_knn(table)
    d_edges = torch.full((n * k, 2), -7, dtype=torch.int32, device="cuda")
    ne = eng.edges_from_knn_dev(d_table, n, k, d_edges)
    assert ne == len(want) and np.array_equal(d_edges[:ne].cpu().numpy(), want) and (d_edges[ne:] == -7).all()
    holes = table.copy()
    holes[::7, 3:] = -1                                                         # rows with fewer neighbours (a bounded search)
    assert np.array_equal(_edges_dev(eng, holes, n, k), eng.edges_from_knn(holes))
    few = torch.zeros((100, 2), dtype=torch.int32, device="cuda")
    assert eng.edges_from_knn_dev(d_table, n, k, few) == ne and np.array_equal(few.cpu().numpy(), want[:100])
    bad = d_table.clone()
    bad[17, 2] = n
    with pytest.raises(irt.OutOfRange):
        eng.edges_from_knn_dev(bad, n, k, d_edges)
    bad[17, 2] = -2
    with pytest.raises(irt.OutOfRange):
        eng.edges_from_knn_dev(bad, n, k, d_edges)
    with pytest.raises(irt.OutOfRange):
        eng.knn_range_dev(d_states, n, n - 10, 11, k, torch.zeros((11, k), dtype=torch.int32, device="cuda"))
    assert eng.edges_from_knn_dev(d_table, n, k, d_edges) == ne                  # the context is usable afterwards


def _edges_dev(eng, table, n, k):
    import torch
    d_e = torch.empty((n * k, 2), dtype=torch.int32, device="cuda")
    m = eng.edges_from_knn_dev(torch.from_numpy(np.ascontiguousarray(table, dtype=np.int32)).cuda(), n, k, d_e)
    return d_e[:m].cpu().numpy()
