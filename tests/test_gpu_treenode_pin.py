"""GPU legs of the reference-pinned vectors (tests/golden/treenode_*.npz, produced by the reference's own
`collision::detail::TreeNode<N>`, see tests/golden/make_treenode_golden.py): K4 `tr_check_cached` must give the
reference's octree-AND-octree verdicts (TreeNode.hxx:164-174,268) and K5 `tr_voxelize_batch` the block lists the
reference's set_cell -> union_block storage holds for the same cells (TreeNode.hxx:140-148, leaf :255-259)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _ids(bxyz, Nb):
    b = np.asarray(bxyz, dtype=np.int64).reshape(-1, 3)
    return ((b[:, 0] * Nb + b[:, 1]) * Nb + b[:, 2]).astype(np.uint32)


@pytest.mark.parametrize("N", [256, 64])
def test_k4_hit_bits_equal_the_reference_collides(irt, N):
    d = np.load(os.path.join(GOLD, "treenode_collides.npz"))
    Nb = N // 4
    vox = irt.VoxelOctree.from_sparse(N, (-0.25, 0.25) * 3, _ids(d["N%d_obst_bxyz" % N], Nb), d["N%d_obst_val" % N])
    chk = irt.VoxelBackboneValidityChecker(irt.workloads.robot_config2(), irt.VoxelEnvironment(), vox)
    ids, masks, off = _ids(d["N%d_item_bxyz" % N], Nb), d["N%d_item_val" % N].astype(np.uint64), d["N%d_item_offsets" % N]
    want = d["N%d_hit" % N]
    got = chk.engine.check_cached(ids, masks, off)
    assert np.array_equal(got, want)
    # the items as a roadmap file would hand them over (visit_leaves order) and in the engine's ascending order: same bits
    srt_ids, srt_masks = ids.copy(), masks.copy()
    for i in range(len(want)):
        o = np.argsort(ids[off[i]:off[i + 1]])
        srt_ids[off[i]:off[i + 1]], srt_masks[off[i]:off[i + 1]] = ids[off[i]:off[i + 1]][o], masks[off[i]:off[i + 1]][o]
    assert np.array_equal(chk.engine.check_cached(srt_ids, srt_masks, off), want)
    # the union of two items collides iff one of them does (an edge's swept volume, union_tree :97-108)
    ub, uv, uo = _ids(d["N%d_union_bxyz" % N], Nb), d["N%d_union_val" % N].astype(np.uint64), d["N%d_union_offsets" % N]
    pairs = d["N%d_pairs" % N]
    assert np.array_equal(chk.engine.check_cached(ub, uv, uo), want[pairs[:, 0]] | want[pairs[:, 1]])


def test_k5_block_lists_equal_the_reference_storage(irt):
    """tr_voxelize_batch on the fixture's states against the leaves the reference's octree holds after set_cell of the
    oracle's cells.  The GPU integrates its own backbone (<= 2e-15 m from the oracle's): a point within that of a voxel
    face may move one cell, so at most two bits may differ in at most two of the 160 sets; all others are identical,
    list for list, in the reference's serialisation order too."""
    d = np.load(os.path.join(GOLD, "treenode_backbones.npz"))
    robot = irt.workloads.robot_config2()
    vox = irt.VoxelOctree(256)
    vox.set_xlim(*d["limits"][0:2]); vox.set_ylim(*d["limits"][2:4]); vox.set_zlim(*d["limits"][4:6])
    chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
    out = chk.engine.voxelize_batch(d["states"])
    assert out["shape_valid"].all()
    lo = d["leaf_offsets"]
    want_ids, want_val = _ids(d["leaves_bxyz"], 64), d["leaves_val"].astype(np.uint64)
    off = out["offsets"]
    n_diff = 0
    for i in range(len(d["states"])):
        gi, gm = out["block_ids"][off[i]:off[i + 1]], out["masks"][off[i]:off[i + 1]]
        o = irt.collision.leaf_order_of(gi, 64)
        wi, wm = want_ids[lo[i]:lo[i + 1]], want_val[lo[i]:lo[i + 1]]
        if np.array_equal(gi[o], wi) and np.array_equal(gm[o], wm):
            continue
        a = {int(k): int(v) for k, v in zip(gi, gm)}
        b = {int(k): int(v) for k, v in zip(wi, wm)}
        assert sum(bin(a.get(k, 0) ^ b.get(k, 0)).count("1") for k in set(a) | set(b)) <= 2, i
        n_diff += 1
    assert n_diff <= 2
