"""The oracle (and the host mirror) against vectors produced by the REFERENCE's own octree class.

tests/golden/treenode_*.npz were written by tests/golden/make_treenode_golden.py driving
`collision::detail::TreeNode<N>` compiled from /root/reference/cpp/src/collision/detail/TreeNode.h/.hxx as they lie
(oracle/_ref, `make -C oracle ref`).  They pin SURVEY 8(a)11 (octree AND octree, TreeNode.hxx:164-174,268) and the
block-storage half of 8(a)10 (set_block / union_block, :74-95,140-148), plus the order voxel sets are serialised in
(visit_leaves, :176-190).  CPU only; the GPU legs are in tests/test_gpu_treenode_pin.py.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
OP_SET, OP_UNION, OP_INTERSECT, OP_BLOCK = 0, 1, 2, 3
FULL = 2 ** 64 - 1


def _ids(bxyz, Nb):
    b = np.asarray(bxyz, dtype=np.int64).reshape(-1, 3)
    return ((b[:, 0] * Nb + b[:, 1]) * Nb + b[:, 2]).astype(np.uint32)


@pytest.mark.parametrize("N", [8, 32, 256])
def test_block_operations_replayed_on_the_oracle_grid(orc, N):
    """set_block / union_block / intersect_block / block: every return value, the leaf count after every operation,
    and the final leaves in the reference's visit order."""
    d = np.load(os.path.join(GOLD, "treenode_ops.npz"))
    g = orc.Grid(N)
    op, bxyz, val, ret, nb_after = (d["N%d_%s" % (N, k)] for k in ("op", "bxyz", "val", "ret", "nblocks_after"))
    for i in range(len(op)):
        b = tuple(int(x) for x in bxyz[i])
        v = int(val[i])
        if op[i] == OP_SET:
            g.set_block(*b, v)
        elif op[i] == OP_UNION:
            assert g.union_block(*b, v) == int(ret[i]), i
        elif op[i] == OP_INTERSECT:                          # VoxelOctree::intersect_block = AND, old value returned
            old = g.block(*b)
            g.set_block(*b, old & v)
            assert old == int(ret[i]), i
        else:
            assert g.block(*b) == int(ret[i]), i
        if i % 16 == 0 or i == len(op) - 1:
            assert g.nblocks() == int(nb_after[i]), i        # a leaf exists exactly where a block is non-zero
    ids, masks = g.export_blocks(leaf_order=True)
    assert np.array_equal(ids, _ids(d["N%d_leaves_bxyz" % N], N // 4))
    assert np.array_equal(masks, d["N%d_leaves_val" % N])
    assert bool(d["N%d_is_empty" % N]) == (g.nblocks() == 0)


@pytest.mark.parametrize("N", [8, 32, 256])
def test_leaf_order_is_the_interleaved_key(irt, N):
    """The product's host-side ordering (collision.leaf_order_of: roadmap files, voxel files) = the reference's."""
    d = np.load(os.path.join(GOLD, "treenode_ops.npz"))
    want = _ids(d["N%d_leaves_bxyz" % N], N // 4)
    shuffled = np.random.default_rng(5).permutation(want)
    got = shuffled[irt.collision.leaf_order_of(shuffled, N // 4)]
    assert np.array_equal(got, want)
    v = irt.VoxelOctree.from_sparse(N, (0, 1) * 3, want, d["N%d_leaves_val" % N])
    seen = []
    v.visit_leaves(lambda bx, by, bz, m: seen.append((bx, by, bz, m)))
    assert [s[:3] for s in seen] == [tuple(r) for r in d["N%d_leaves_bxyz" % N].tolist()]
    assert [s[3] for s in seen] == d["N%d_leaves_val" % N].tolist()
    assert [r[:3] for r in v.to_json()["VoxelOctree"]["data"]] == d["N%d_leaves_bxyz" % N].tolist()


@pytest.mark.parametrize("N", [256, 64])
def test_octree_and_octree_verdicts(orc, N):
    """obstacles.collides(item) for every item tree = the oracle's dense AND (orc_grid_collides, orc_check_cached)."""
    d = np.load(os.path.join(GOLD, "treenode_collides.npz"))
    Nb = N // 4
    og = orc.Grid(N)
    og.blocks().reshape(-1)[_ids(d["N%d_obst_bxyz" % N], Nb)] = d["N%d_obst_val" % N]
    ids, masks, off = _ids(d["N%d_item_bxyz" % N], Nb), d["N%d_item_val" % N].astype(np.uint64), d["N%d_item_offsets" % N]
    want = d["N%d_hit" % N]
    assert 0.2 < want.mean() < 0.8
    assert np.array_equal(orc.check_cached(og, ids, masks, off), want)
    for i in range(0, len(want), 7):
        it = orc.Grid(N)
        it.blocks().reshape(-1)[ids[off[i]:off[i + 1]]] = masks[off[i]:off[i + 1]]
        assert og.collides(it) == bool(want[i]) and it.collides(og) == bool(want[i])
        assert it.nblocks() == off[i + 1] - off[i]
    with pytest.raises(ValueError):
        og.collides(orc.Grid(N // 2))                         # dimension mismatch: std::invalid_argument (VoxelOctree.cpp:46-53)


@pytest.mark.parametrize("N", [256, 64])
def test_tree_set_operations(orc, N):
    """union_tree / intersect_tree / remove_tree of item pairs (an edge's swept volume is the union of its samples')."""
    d = np.load(os.path.join(GOLD, "treenode_collides.npz"))
    Nb = N // 4
    ids, masks, off = _ids(d["N%d_item_bxyz" % N], Nb), d["N%d_item_val" % N].astype(np.uint64), d["N%d_item_offsets" % N]

    def dense(k):
        a = np.zeros(Nb ** 3, dtype=np.uint64)
        a[ids[off[k]:off[k + 1]]] = masks[off[k]:off[k + 1]]
        return a

    for name, fn in (("union", lambda a, b: a | b), ("inter", lambda a, b: a & b), ("remove", lambda a, b: a & ~b)):
        rb, rv, ro = _ids(d["N%d_%s_bxyz" % (N, name)], Nb), d["N%d_%s_val" % (N, name)], d["N%d_%s_offsets" % (N, name)]
        for j, (a, b) in enumerate(d["N%d_pairs" % N]):
            g = orc.Grid(N)
            g.blocks().reshape(-1)[:] = fn(dense(a), dense(b))
            gi, gm = g.export_blocks(leaf_order=True)
            assert np.array_equal(gi, rb[ro[j]:ro[j + 1]]) and np.array_equal(gm, rv[ro[j]:ro[j + 1]]), (name, j)


def test_backbone_voxel_sets_as_the_reference_stores_them(orc, irt, helpers):
    """The oracle's add_piecewise_line of its own shapes, exported in leaf order = the reference's leaves after
    set_cell -> union_block of the same cells: block layout, bit layout and serialisation order of a vertex cache."""
    d = np.load(os.path.join(GOLD, "treenode_backbones.npz"))
    robot = irt.workloads.robot_config2()
    orb = helpers.oracle_robot(orc, robot)
    grid = orc.Grid(256, tuple(d["limits"]))
    lo = d["leaf_offsets"]
    want_ids = _ids(d["leaves_bxyz"], 64)
    for i, st in enumerate(d["states"]):
        g = grid.empty_copy()
        g.add_piecewise_line(orb.shape(st)["p"])
        ids, masks = g.export_blocks(leaf_order=True)
        assert np.array_equal(ids, want_ids[lo[i]:lo[i + 1]]) and np.array_equal(masks, d["leaves_val"][lo[i]:lo[i + 1]]), i
        assert g.nblocks() == lo[i + 1] - lo[i]


def test_rmp_block_records_are_written_in_the_reference_order(irt, tmp_path):
    """serialize_inner (VoxelCachedLazyPRM.cpp:633-642) streams visit_leaves order; write_rmp must emit the same bytes
    whatever order the CSR holds the blocks in, and read them back as the same sets."""
    d = np.load(os.path.join(GOLD, "treenode_backbones.npz"))
    lo = d["leaf_offsets"][:9]
    ids = _ids(d["leaves_bxyz"][: lo[-1]], 64)
    masks = d["leaves_val"][: lo[-1]].astype(np.uint64)
    srt_ids, srt_masks = ids.copy(), masks.copy()
    for i in range(8):                                           # the engine's CSR: ascending block id per item
        o = np.argsort(ids[lo[i]:lo[i + 1]])
        srt_ids[lo[i]:lo[i + 1]], srt_masks[lo[i]:lo[i + 1]] = ids[lo[i]:lo[i + 1]][o], masks[lo[i]:lo[i + 1]][o]
    states = d["states"][:8]
    f = str(tmp_path / "a.rmp")
    irt.rmp.write_rmp(f, states, vertex_caches=dict(offsets=lo, block_ids=srt_ids, masks=srt_masks), N=256, limits=d["limits"])
    r = irt.rmp.read_rmp(f)
    assert np.array_equal(r["vertex_caches"]["block_ids"], ids) and np.array_equal(r["vertex_caches"]["masks"], masks)
    raw = open(f, "rb").read()
    first = raw.index(np.uint32(lo[1]).tobytes(), 9 + 49)        # u32 nblocks of vertex 0, then its 11-byte records
    rec = np.frombuffer(raw, dtype=irt.rmp._BLOCK, count=int(lo[1]), offset=first + 4)
    assert np.array_equal(np.stack([rec["bx"], rec["by"], rec["bz"]], 1), d["leaves_bxyz"][: lo[1]])


def test_live_against_the_reference_class_when_it_is_built(orc):
    """Where oracle/_ref/libref_treenode.so exists (the build container; it also travels to the GPU box): fresh random
    block sets through the reference class and through the oracle's dense grid, beyond the committed vectors."""
    from oracle import ref_treenode as ref
    ref.build()
    if not ref.available():
        pytest.skip("oracle/_ref is not built on this machine (needs /root/reference)")
    rng = np.random.default_rng(99)
    for N in (16, 128, 256):
        Nb = N // 4
        for _ in range(20):
            n = int(rng.integers(1, 300))
            b = rng.integers(0, Nb, size=(n, 3)).astype(np.uint32)
            m = rng.integers(1, 2 ** 63, size=n, dtype=np.uint64) & rng.integers(1, 2 ** 63, size=n, dtype=np.uint64)
            m[m == 0] = 1
            t, g = ref.RefTree(N), orc.Grid(N)
            prev = t.union_blocks(b, m)
            for k in range(n):
                assert g.union_block(*map(int, b[k]), int(m[k])) == int(prev[k])
            lb, lv = t.leaves()
            gi, gm = g.export_blocks(leaf_order=True)
            assert np.array_equal(gi, _ids(lb, Nb)) and np.array_equal(gm, lv) and t.nblocks() == g.nblocks()
            b2 = np.clip(b[: n // 2 + 1] + rng.integers(-1, 2, size=(n // 2 + 1, 3)), 0, Nb - 1).astype(np.uint32)
            m2 = rng.integers(1, 2 ** 63, size=len(b2), dtype=np.uint64) & rng.integers(1, 2 ** 63, size=len(b2), dtype=np.uint64)
            m2[m2 == 0] = 2
            t2, g2 = ref.RefTree(N), orc.Grid(N)
            t2.union_blocks(b2, m2)
            for k in range(len(b2)):
                g2.union_block(*map(int, b2[k]), int(m2[k]))
            assert t.collides(t2) == g.collides(g2) == g2.collides(g)
