"""createRoadmap's vertex phase on the device (tr_candidate_states*, tr_validate_candidates_dev, tr_compact_rows_dev,
tr_sample_valid_vertices*): motion-planning/VoxelCachedLazyPRM.cpp:1415-1455 (sample -> fk -> valid -> voxelize -> collides,
repeat until N accepted) as generator + fk_verdict + order-preserving compaction, all in HBM.  The accepted set must be the
first N valid candidates of the counter-based sequence -- for every batch split -- with the ORACLE's verdicts."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(irt, robot):
    W = irt.workloads
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    return irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)


@pytest.mark.parametrize("which", ["config2", "config3", "rot_ret"])
def test_device_generator_equals_the_host_mirror(irt, which):
    W, D = irt.workloads, irt.distributed
    robot = {"config2": W.robot_config2, "config3": W.robot_config3}.get(which, W.robot_config3)()
    if which == "rot_ret":
        robot.enable_rotation = robot.enable_retraction = True
    chk = _setup(irt, robot)
    e = chk.engine
    for seed, first, count in ((0, 0, 1000), (2024, 2 ** 32 - 300, 777), (2 ** 63 + 12345, 2 ** 40 + 1, 4096), (7, 5, 1)):
        got = e.candidate_states(seed, first, count)
        want = D.candidate_states(robot, seed, first, count)
        assert got.shape == (count, robot.state_size()) and np.array_equal(got, want)
    box = D.sampling_box(robot, tau_max=3.5)
    assert np.array_equal(e.candidate_states(11, 64, 500, box=box), D.candidate_states(robot, 11, 64, 500, tau_max=3.5))
    lo, hi = D.sampling_box(robot)
    big = e.candidate_states(3, 0, 200000)
    assert (big >= lo).all() and (big < hi + 1e-12).all()
    assert np.abs(big.mean(axis=0) - (lo + hi) / 2).max() < 0.01 * (hi - lo).max()      # uniform on the box
    assert np.abs(np.corrcoef(big.T) - np.eye(big.shape[1])).max() < 0.01               # coordinates independent
    # the default box is the planner's state space
    assert np.array_equal(e.candidate_states(3, 10, 50), D.candidate_states(robot, 3, 10, 50, box=(lo, hi)))


def test_sampled_vertices_are_the_first_valid_candidates_with_oracle_verdicts(irt, orc, helpers):
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config2()
    for t in robot.tendons:
        t.max_tension = 30.0
    chk = _setup(irt, robot)
    e = chk.engine
    out = e.sample_valid_vertices(3000, seed=5, want_index=True)
    assert out["accepted"] == 3000
    tried = out["tried"]
    cand = D.candidate_states(robot, 5, 0, tried)
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    want, tips, _ = orc.validate_batch(helpers.oracle_robot(orc, robot), helpers.oracle_grid(orc, vox), cand, nthreads=0, lib=orc.omp_lib())
    assert 0.2 < want.mean() < 0.95
    idx = np.flatnonzero(want)
    assert len(idx) == 3000 and idx[-1] == tried - 1            # the 3000th valid candidate is the last one consumed
    assert np.array_equal(out["index"], idx)
    assert np.array_equal(out["states"], cand[idx])
    assert np.abs(out["tips"] - tips[idx]).max() <= 1e-9


@pytest.mark.parametrize("which", ["config3", "retract", "spheres"])
def test_accepted_set_does_not_depend_on_the_batching(irt, which):
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3()
    if which == "retract":
        robot.enable_rotation = robot.enable_retraction = True
    chk = _setup(irt, robot)
    e = chk.engine
    if which == "spheres":
        e.set_checker(True)
    n_want = 5000
    a = e.sample_valid_vertices(n_want, seed=17, first_candidate=128, want_index=True)      # first call: rate unknown, two or three batches
    b = e.sample_valid_vertices(n_want, seed=17, first_candidate=128, want_index=True)      # second call: rate known, one batch
    for k in ("states", "tips", "index"):
        assert np.array_equal(a[k], b[k]), k
    assert a["tried"] == b["tried"] and a["accepted"] == n_want
    # against the plain batch call on host-generated candidates
    cand = D.candidate_states(robot, 17, 128, a["tried"])
    ok = chk.is_valid_detail(cand)
    idx = np.flatnonzero(ok["valid"])
    assert len(idx) == n_want and np.array_equal(a["index"], idx + 128)
    assert np.array_equal(a["states"], cand[idx]) and np.array_equal(a["tips"], ok["tips"][idx])
    # a prefix request is a prefix of the answer
    c = e.sample_valid_vertices(777, seed=17, first_candidate=128, want_index=True)
    assert np.array_equal(c["index"], a["index"][:777]) and c["tried"] == a["index"][776] - 128 + 1
    # max_candidates: fewer vertices than asked for, all candidates consumed
    d = e.sample_valid_vertices(n_want, seed=17, first_candidate=128, max_candidates=1024, want_index=True)
    k = int((idx < 1024).sum())
    assert d["accepted"] == k and d["tried"] == 1024 and np.array_equal(d["index"], a["index"][:k])
    # ... also when the bound is not a whole number of waves (round 3 consumed up to 63 candidates beyond it)
    d = e.sample_valid_vertices(n_want, seed=17, first_candidate=128, max_candidates=1000, want_index=True)
    k = int((idx < 1000).sum())
    assert d["accepted"] == k and d["tried"] == 1000 and np.array_equal(d["index"], a["index"][:k])


def test_device_form_and_shard_form(irt):
    import torch
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3()
    chk = _setup(irt, robot)
    e = chk.engine
    S = e.state_size
    n_want = 4000
    d_states = torch.empty(n_want * S, dtype=torch.float64, device="cuda")
    d_tips = torch.empty(n_want * 3, dtype=torch.float64, device="cuda")
    d_idx = torch.empty(n_want, dtype=torch.int64, device="cuda")
    side = torch.cuda.Stream()
    n_acc, n_tried = e.sample_valid_vertices_dev(n_want, d_states, d_tips, d_idx, seed=23, stream=side)
    side.synchronize()
    host = e.sample_valid_vertices(n_want, seed=23, want_index=True)
    assert (n_acc, n_tried) == (n_want, host["tried"])
    assert np.array_equal(d_states.cpu().numpy().reshape(n_want, S), host["states"])
    assert np.array_equal(d_tips.cpu().numpy().reshape(n_want, 3), host["tips"])
    assert np.array_equal(d_idx.cpu().numpy(), host["index"])
    # config 4's shard form: two "ranks" validate their halves of M candidates on the device, the concatenated mask and
    # the regenerate-and-compact step give the same vertices as one rank
    M = 20000 + 37                                             # not a whole number of mask words
    masks = []
    for rank in range(2):
        v = D.ShardedVertexValidator(robot, seed=23, device="cuda", validate_candidates=D.device_candidate_validator(e, 23))
        start, stop, shard = D.shard_bounds(M, 2, rank)
        n_real = max(0, min(stop, M) - start)
        masks.append(v.validate_candidates(start, n_real, shard // 64))
    full = torch.cat(masks)
    one = D.ShardedVertexValidator(robot, seed=23, device="cuda", validate_candidates=D.device_candidate_validator(e, 23)).run(
        M, rank=0, world_size=1, keep_on_device=True)
    nw = (M + 63) // 64
    assert torch.equal(full[:nw], one[:nw]) and int(full[nw:].abs().sum()) == 0
    mask = irt.unpack_bits(full.cpu().numpy().view(np.uint64), M)
    cand = D.candidate_states(robot, 23, 0, M)
    assert np.array_equal(mask, chk.is_valid_detail(cand)["valid"])
    verts, idx = D.gather_valid_vertices_dev(e, 23, M, full)
    assert np.array_equal(idx.cpu().numpy(), np.flatnonzero(mask)) and np.array_equal(verts.cpu().numpy(), cand[mask])
    # compaction edge cases: empty mask, full mask, capacity smaller than the set bits
    rows = torch.arange(130 * 2, dtype=torch.float64, device="cuda")
    out = torch.zeros(130 * 2, dtype=torch.float64, device="cuda")
    zero = torch.zeros(3, dtype=torch.int64, device="cuda")
    assert e.compact_rows_dev(zero, 130, rows, 2, out, 130) == 0
    ones = torch.full((3,), -1, dtype=torch.int64, device="cuda")
    assert e.compact_rows_dev(ones, 130, rows, 2, out, 130) == 130 and torch.equal(out, rows)
    out.zero_()
    assert e.compact_rows_dev(ones, 130, rows, 2, out, 50) == 130
    assert torch.equal(out[:100], rows[:100]) and int(out[100:].abs().sum()) == 0


@pytest.mark.parametrize("which", ["config3", "config2_rotating"])
def test_signature_rows_survive_the_wire_coding(irt, which):
    """tr_pack_signatures_dev / tr_unpack_signatures_dev: the accepted candidates' signature rows delta-coded for the all-gather of a
    sharded build (first cell + 6 bits per further point: 104 instead of 576 bytes at 129 points) come back word for word; rows that
    cannot be coded -- an invalid candidate whose backbone leaves the voxel domain (SIG_BAD), a row with a jump of two cells -- are
    counted, so that the caller sends them as they are.  Through distributed.signature_wire_codec + run_with_rows: same rows."""
    import torch
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3() if which == "config3" else W.robot_config2()
    if which != "config3":
        robot.enable_rotation = True
    chk = _setup(irt, robot)
    e = chk.engine
    sw, pw, P = e.signature_words(), e.signature_packed_words(), e.num_points
    assert sw == 144 and P == 129 and pw == 26 and pw % 2 == 0
    M = 30000 + 13
    validate = D.device_candidate_validator(e, 31, signatures=True)
    bits, sig = validate(0, M, (M + 63) // 64)
    mine = D.device_row_compactor(e)(bits, M, sig)
    n = mine.shape[0]
    assert 0.3 * M < n < M
    packed, bad = e.pack_signatures_dev(mine)
    assert bad == 0 and packed.shape == (n, pw)
    back = e.unpack_signatures_dev(packed)
    assert torch.equal(back[:, :P], mine[:, :P])
    # the coding itself, on the host: first word, then (d + 1) per axis in 2-bit fields, five points to a word, the sixteenth in the top bits
    row = mine[7, :P].cpu().numpy().astype(np.int64)
    cell = np.stack([row & 1023, (row >> 10) & 1023, (row >> 20) & 1023], 1)
    code = ((np.diff(cell, axis=0) + 1) * np.array([1, 4, 16])).sum(1)
    pk = packed[7].cpu().numpy().view(np.uint32).astype(np.uint64)
    assert pk[0] == row[0]
    for j, c in enumerate(code):
        g, q = divmod(j, 16)
        w3 = pk[1 + 3 * g: 4 + 3 * g]
        got = (int(w3[q // 5]) >> (6 * (q % 5))) & 63 if q < 15 else (int(w3[0]) >> 30) | ((int(w3[1]) >> 30) << 2) | ((int(w3[2]) >> 30) << 4)
        assert got == c, (j, got, c)
    # rows that cannot be coded are reported: a two-cell jump, a point outside the domain
    broken = mine[:64].clone()
    broken[3, 40] = (broken[3, 40] + 2) & 0x3FFFFFFF if int(broken[3, 40] & 1023) < 1000 else broken[3, 40] - 2
    broken[9, 100] = broken[9, 100] | (1 << 30)
    _, bad = e.pack_signatures_dev(broken)
    assert bad >= 2
    # through the sharded validator: world size 1 packs and unpacks on the way (what every rank does around the all-gather)
    v = D.ShardedVertexValidator(robot, seed=31, device="cuda", validate_candidates=validate)
    full, rows = v.run_with_rows(M, D.device_row_compactor(e), rank=0, world_size=1, codec=D.signature_wire_codec(e))
    assert v.rows_on_the_wire == "packed" and torch.equal(rows[:, :P], mine[:, :P]) and torch.equal(full[: bits.numel()], bits)
    empty, bad = D.signature_wire_codec(e).pack(mine[:0])
    assert empty.shape == (0, pw) and bad == 0


def test_roadmap_builder_uses_the_device_phase(irt):
    W, D = irt.workloads, irt.distributed
    robot = W.robot_config3()
    chk = _setup(irt, robot)
    mv = irt.VoxelBackboneMotionValidator(chk)
    rb = irt.RoadmapBuilder(chk, mv, seed=5)
    states, tips = rb.sample_valid_vertices(2000)
    t = rb.timing["vertices"]
    cand = D.candidate_states(robot, 5, 0, t["candidates"])
    ok = chk.is_valid_detail(cand)["valid"]
    assert ok.sum() == 2000 and ok[-1] and np.array_equal(states, cand[ok])


def test_builder_samples_the_same_vertices_on_the_other_schedules(irt):
    """RoadmapBuilder.sample_valid_vertices on a context that runs the stored-point schedules (TENDON_HIP_FUSED=0 / 1, where the device
    sampler reports TR_ERR_UNSUPPORTED): the host rejection loop over the same candidate sequence accepts the same vertices."""
    import os
    W = irt.workloads
    robot = W.robot_config3()
    vox, _ = W.reach_environment(seed=7, n_spheres=64)
    got = {}
    for fused in ("2", "1", "0"):
        old = os.environ.get("TENDON_HIP_FUSED")
        os.environ["TENDON_HIP_FUSED"] = fused
        try:
            chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
            rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=5)
            got[fused] = rb.sample_valid_vertices(3000, batch=4096)
            tried = rb.timing["vertices"]["candidates"]
            assert fused == "2" or tried == got["tried"], (fused, tried, got["tried"])
            got["tried"] = tried
        finally:
            if old is None:
                os.environ.pop("TENDON_HIP_FUSED", None)
            else:
                os.environ["TENDON_HIP_FUSED"] = old
    for fused in ("1", "0"):
        assert np.array_equal(got[fused][0], got["2"][0])
        assert np.abs(got[fused][1] - got["2"][1]).max() <= 1e-13          # tips: another kernel's rounding
