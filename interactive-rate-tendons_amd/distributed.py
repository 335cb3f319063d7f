"""Multi-GPU sharding of roadmap vertex validation (BASELINE config 4): one process per GPU,
contiguous vertex shards, no data-path collective inside the FK/collision kernels, and ONE
all-gather of the packed validity bitmask (RCCL over xGMI; `gloo` in CPU tests) before the
host connects edges -- the exchange step the reference's single-process createRoadmap gets for
free from shared memory (motion-planning/VoxelCachedLazyPRM.cpp:1446-1483: parallel vertex
validation, then serial addMilestone over ALL vertices).

Candidate vertices come from a counter-based generator keyed by (seed, global candidate index) -- Philox-4x32-10, the same
function on the device (csrc/sample.hip) and here -- so the candidate set and therefore the gathered mask are identical for
every world size and every batch size.
"""
import numpy as np

WORD = 64


def shard_bounds(M, world_size, rank):
    """Contiguous equal shards of M items padded so every shard is a whole number of 64-bit mask
    words (all-gather needs equal counts).  Returns (start, stop, shard_len); stop may exceed M
    for the padding tail, whose verdict bits are forced to 0."""
    words = (M + WORD - 1) // WORD
    words_per_rank = (words + world_size - 1) // world_size
    shard = words_per_rank * WORD
    start = rank * shard
    return start, start + shard, shard


_PHILOX_M0, _PHILOX_M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_PHILOX_W0, _PHILOX_W1 = 0x9E3779B9, 0xBB67AE85
_M32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox-4x32-10 (Salmon et al., SC'11) on arrays of 32-bit counter words held in uint64; key words are Python ints.
    The device generator (csrc/sample.hip) is the same function; tests hold the paper's known-answer vectors."""
    c0, c1, c2, c3 = (np.asarray(x, dtype=np.uint64) & _M32 for x in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = _PHILOX_M0 * c0, _PHILOX_M1 * c2                      # 32 x 32 -> 64 bit products
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c0, c1, c2, c3 = n0, p1 & _M32, n2, p0 & _M32
        k0, k1 = (k0 + _PHILOX_W0) & 0xFFFFFFFF, (k1 + _PHILOX_W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def sampling_box(robot, tau_max=None):
    """(lo, hi) of the planner's state space (motion-planning/Problem.cpp:101-163): tensions [0, max_tension] (or a common
    tau_max), rotation [-pi, pi), retraction [0, L]."""
    lo = [0.0] * len(robot.tendons)
    hi = [float(t.max_tension if tau_max is None else tau_max) for t in robot.tendons]
    if robot.enable_rotation:
        lo.append(-np.pi); hi.append(np.pi)
    if robot.enable_retraction:
        lo.append(0.0); hi.append(float(robot.specs.L))
    return np.array(lo), np.array(hi)


def candidate_states(robot, seed, start, count, tau_max=None, box=None):
    """States [start, start + count) of the global candidate sequence: a pure function of (seed, candidate index) -- the host
    mirror of the device generator (tr_candidate_states / tr_sample_valid_vertices, include/tendon_hip.h): Philox counter
    (index lo, index hi, coordinate pair, 0), key (seed lo, seed hi); 53-bit uniforms u, state[d] = lo[d] + u * (hi[d] - lo[d])
    with the product and the sum rounded separately.  Bit-identical to the device (tests/test_gpu_sampling.py)."""
    lo, hi = box if box is not None else sampling_box(robot, tau_max)
    S = len(lo)
    span = hi - lo
    idx = np.uint64(int(start)) + np.arange(int(count), dtype=np.uint64)
    out = np.empty((int(count), S))
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    for j in range((S + 1) // 2):
        r0, r1, r2, r3 = philox4x32_10(idx & _M32, idx >> np.uint64(32), np.full(len(idx), j, dtype=np.uint64), 0,
                                       seed & 0xFFFFFFFF, seed >> 32)
        ua = (((r0 << np.uint64(32)) | r1) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
        out[:, 2 * j] = lo[2 * j] + ua * span[2 * j]
        if 2 * j + 1 < S:
            ub = (((r2 << np.uint64(32)) | r3) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
            out[:, 2 * j + 1] = lo[2 * j + 1] + ub * span[2 * j + 1]
    return out


def allgather_mask(local_words, group=None):
    """All-gather equal-length int64 mask shards into the global mask (rank-major)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_words.clone()
    world = dist.get_world_size(group)
    if local_words.is_cuda and dist.get_backend(group) == "gloo":
        # rehearsal only (several ranks sharing one GPU, bench.py TENDON_BENCH_SHARED_GPU=1): gloo gathers through host memory
        out = torch.empty(world * local_words.numel(), dtype=local_words.dtype)
        dist.all_gather_into_tensor(out, local_words.cpu().contiguous(), group=group)
        return out.to(local_words.device)
    out = torch.empty(world * local_words.numel(), dtype=local_words.dtype, device=local_words.device)
    dist.all_gather_into_tensor(out, local_words.contiguous(), group=group)
    return out


def allgather_rows(local_rows, counts, group=None):
    """All-gather row blocks of different lengths: rank r contributes counts[r] rows (counts is known to every rank, e.g. from
    an all-gathered mask); shorter blocks are padded to the longest for the collective.  Returns the rows in rank order."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_rows
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    assert len(counts) == world and local_rows.shape[0] == counts[rank]
    w = int(np.prod(local_rows.shape[1:])) if local_rows.dim() > 1 else 1
    most = int(max(counts))
    buf = torch.zeros(most * w, dtype=local_rows.dtype, device=local_rows.device)
    buf[: counts[rank] * w] = local_rows.reshape(-1)
    full = allgather_mask(buf, group=group)
    parts = [full[r * most * w: r * most * w + int(counts[r]) * w] for r in range(world)]
    return torch.cat(parts).reshape((-1,) + tuple(local_rows.shape[1:]))


class ShardedVertexValidator:
    """Validate M candidate vertices across the ranks of the default process group.

    Production (one rank per GPU): `validate_candidates(first, count, n_words)` -> int64 tensor of n_words mask words ON THE
    DEVICE for candidates [first, first + count) -- `device_candidate_validator(engine, seed, box)` builds it from
    Engine.validate_candidates_dev, which generates the candidates in HBM, so neither states nor mask touch the host before
    the all-gather.  CPU tests (gloo): `validate_local(states)` -> uint64/int64 mask words for host states from the host
    mirror of the generator (the oracle plugs in there).  Bit i & 63 of word i >> 6; padding bits zero.
    """

    def __init__(self, robot, validate_local=None, seed=0, tau_max=None, device="cpu", validate_candidates=None, box=None):
        self.robot, self.validate_local, self.validate_candidates = robot, validate_local, validate_candidates
        self.seed, self.tau_max, self.device, self.box = seed, tau_max, device, box

    def run(self, M, rank=None, world_size=None, keep_on_device=False):
        import torch
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        start, stop, shard = shard_bounds(M, world_size, rank)
        n_real = max(0, min(stop, M) - start)
        if self.validate_candidates is not None:
            local = self.validate_candidates(start, n_real, shard // WORD)
        else:
            words = np.zeros(shard // WORD, dtype=np.uint64)
            if n_real > 0:
                states = candidate_states(self.robot, self.seed, start, n_real, self.tau_max, box=self.box)
                w = np.asarray(self.validate_local(states)).view(np.uint64)
                words[: w.size] = w
                if n_real % WORD:                              # padding bits stay zero
                    words[n_real // WORD] &= np.uint64((1 << (n_real % WORD)) - 1)
            local = torch.from_numpy(words.view(np.int64)).to(self.device)
        full = allgather_mask(local)
        # world_size * shard/64 words; bits of items >= M are zero.  unpack_bits(words, M) is the mask.
        return full if keep_on_device else full.cpu().numpy().view(np.uint64)

    def run_with_rows(self, M, compact, rank=None, world_size=None, codec=None):
        """run(), and a row of data per ACCEPTED candidate gathered with the mask: validate_candidates(first, count, n_words) returns
        (mask words, rows [count, w]) here -- device_candidate_validator(..., signatures=True): the candidates' backbone signatures,
        which tr_validate_edges_indexed_sig_dev takes instead of integrating the vertices again on every rank -- and
        compact(mask words, count, rows) -> the accepted candidates' rows in order.  Returns (mask words tensor, rows of all
        accepted candidates in candidate order, the same on every rank)."""
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        start, stop, shard = shard_bounds(M, world_size, rank)
        n_real = max(0, min(stop, M) - start)
        local, rows = self.validate_candidates(start, n_real, shard // WORD)
        full = allgather_mask(local)
        words = full.cpu().numpy().view(np.uint64).reshape(world_size, shard // WORD)
        counts = [int(np.unpackbits(words[r].view(np.uint8)).sum()) for r in range(world_size)]
        mine = compact(local, n_real, rows)
        if codec is not None:
            # the rows travel packed (signature_wire_codec: 104 instead of 576 bytes per vertex at 129 backbone points) unless some
            # rank holds a row it cannot code -- then every rank sends its rows as they are (one small all-reduce decides)
            packed, bad = codec.pack(mine)
            if dist.is_available() and dist.is_initialized() and world_size > 1:
                import torch
                flag = torch.tensor([bad], dtype=torch.int64, device=mine.device if dist.get_backend() != "gloo" else "cpu")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                bad = int(flag.item())
            self.rows_on_the_wire = "raw" if bad else "packed"
            if not bad:
                return full, codec.unpack(allgather_rows(packed, counts))
        return full, allgather_rows(mine, counts)


def device_candidate_validator(engine, seed, box=None, tips=None, signatures=False):
    """validate_candidates for ShardedVertexValidator on `engine`'s GPU (tr_validate_candidates_dev).  tips (optional): a
    dict that receives the shard's tip tensor under "tips" (n x 3, on the device).  signatures: (mask, signature rows) for
    run_with_rows (tr_validate_candidates_sig_dev)."""
    def validate(first, count, n_words):
        import torch
        dev = "cuda:%d" % engine.device
        bits = torch.zeros(n_words, dtype=torch.int64, device=dev)
        if signatures:
            sig = torch.empty((max(count, 0), engine.signature_words()), dtype=torch.int32, device=dev)
            if count > 0:
                engine.validate_candidates_sig_dev(seed, first, count, bits, sig, box=box)
            return bits, sig
        if count > 0:
            d_tips = torch.empty(count * 3, dtype=torch.float64, device=dev) if tips is not None else None
            engine.validate_candidates_dev(seed, first, count, bits, d_tips=d_tips, box=box)
            if tips is not None:
                tips["tips"] = d_tips.view(count, 3)
        return bits
    return validate


def signature_wire_codec(engine):
    """codec for ShardedVertexValidator.run_with_rows on `engine`'s GPU: the accepted candidates' signature rows are delta-coded
    before the all-gather and restored after it (tr_pack_signatures_dev / tr_unpack_signatures_dev).  None when the context has
    no signatures to hand over or TENDON_HIP_SIG_WIRE=raw asks for the rows as they are (A/B)."""
    import os
    if os.environ.get("TENDON_HIP_SIG_WIRE", "") == "raw" or engine.signature_words() == 0:
        return None

    class Codec:
        @staticmethod
        def pack(rows):
            if rows.shape[0] == 0:
                import torch
                return torch.empty((0, engine.signature_packed_words()), dtype=torch.int32, device=rows.device), 0
            return engine.pack_signatures_dev(rows.contiguous())

        @staticmethod
        def unpack(packed):
            return engine.unpack_signatures_dev(packed.contiguous())
    return Codec


def device_row_compactor(engine):
    """compact for ShardedVertexValidator.run_with_rows on `engine`'s GPU: int32 rows of an even number of words
    (tr_compact_rows_dev on them as doubles)."""
    def compact(d_mask, count, rows):
        import torch
        w = rows.shape[1]
        out = torch.empty((max(count, 1), w), dtype=torch.int32, device=rows.device)
        if count == 0:
            return out[:0]
        n = engine.compact_rows_dev(d_mask, count, rows.view(torch.float64).reshape(-1), w // 2, out.view(torch.float64).reshape(-1), count)
        return out[:n]
    return compact


def gather_valid_vertices_dev(engine, seed, M, d_mask, box=None):
    """Every rank's copy of the accepted vertices after the all-gather: the M candidates are regenerated in HBM
    (tr_candidate_states_dev, ~10 us per 10^6) and compacted by the gathered mask in candidate order (tr_compact_rows_dev).
    Returns (states tensor [n_valid, S] on the device, candidate indices [n_valid])."""
    import torch
    dev = "cuda:%d" % engine.device
    S = engine.state_size
    cand = torch.empty(M * S, dtype=torch.float64, device=dev)
    engine.candidate_states_dev(seed, 0, M, cand, box=box)
    out = torch.empty(M * S, dtype=torch.float64, device=dev)
    idx = torch.empty(M, dtype=torch.int64, device=dev)
    n = engine.compact_rows_dev(d_mask, M, cand, S, out, M, d_index_out=idx)
    return out[: n * S].view(n, S), idx[:n]


class ShardedEdgeValidator:
    """The second phase of SURVEY section 8(e): every rank holds the same edge list (the host connects
    edges from the gathered vertex mask, VoxelCachedLazyPRM.cpp:1463-1502), validates its contiguous
    shard of it (checkMotion per edge, :1520-1542) and the packed verdicts are all-gathered.

    validate_local(a, b) -> uint64/int64 mask words for the shard's edges; in production
    Engine.validate_edges on the rank's GPU (packed with pack_bits), the oracle in the CPU tests.
    """

    def __init__(self, validate_local, device="cpu"):
        self.validate_local, self.device = validate_local, device

    def run(self, a, b, rank=None, world_size=None):
        import torch
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        M = len(a)
        start, stop, shard = shard_bounds(M, world_size, rank)
        n_real = max(0, min(stop, M) - start)
        words = np.zeros(shard // WORD, dtype=np.uint64)
        if n_real > 0:
            w = np.asarray(self.validate_local(a[start:start + n_real], b[start:start + n_real])).view(np.uint64)
            words[: w.size] = w
            if n_real % WORD:
                words[n_real // WORD] &= np.uint64((1 << (n_real % WORD)) - 1)
        local = torch.from_numpy(words.view(np.int64)).to(self.device)
        return allgather_mask(local).cpu().numpy().view(np.uint64)

    def run_indexed(self, states, edges, rank=None, world_size=None):
        """The roadmap form: every rank holds all vertex states (they follow from the gathered vertex mask) and the whole edge list as
        index pairs; validate_local(states, edge_shard) validates the rank's contiguous shard -- in production
        Engine.validate_edges_indexed, which integrates every vertex ONCE per rank for all of its edges in the shard and bisects
        the shard as two lanes -- and the packed verdicts are all-gathered."""
        import torch
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        edges = np.asarray(edges)
        M = len(edges)
        start, stop, shard = shard_bounds(M, world_size, rank)
        n_real = max(0, min(stop, M) - start)
        words = np.zeros(shard // WORD, dtype=np.uint64)
        if n_real > 0:
            w = np.asarray(self.validate_local(states, edges[start:start + n_real])).view(np.uint64)
            words[: w.size] = w
            if n_real % WORD:
                words[n_real // WORD] &= np.uint64((1 << (n_real % WORD)) - 1)
        local = torch.from_numpy(words.view(np.int64)).to(self.device)
        return allgather_mask(local).cpu().numpy().view(np.uint64)


class ShardedNeighbours:
    """The step between the two: the connection loop (connectionStrategy_(v) for every vertex, :1491-1502) spread over the
    ranks.  Every rank holds all vertex states (they follow from the gathered mask), computes the k-nearest rows of its
    contiguous shard of the vertices against ALL vertices and the rows are all-gathered (int32, equal-sized shards,
    padding rows = -1); the table is then the same on every rank and each derives the same edge list from it.

    knn_local(first, count) -> (count, k) int32 neighbour indices; in production Engine.knn(states, k,
    query_range=(first, count)) on the rank's GPU, a brute-force numpy table in the CPU tests.
    """

    def __init__(self, knn_local, k, device="cpu"):
        self.knn_local, self.k, self.device = knn_local, int(k), device

    def run(self, n, rank=None, world_size=None):
        import torch
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        start, stop, shard = shard_bounds(n, world_size, rank)
        n_real = max(0, min(stop, n) - start)
        rows = np.full((shard, self.k), -1, dtype=np.int32)
        if n_real > 0:
            rows[:n_real] = np.asarray(self.knn_local(start, n_real), dtype=np.int32).reshape(n_real, self.k)
        local = torch.from_numpy(rows.reshape(-1)).to(self.device)
        full = allgather_mask(local).cpu().numpy().reshape(-1, self.k)
        return full[:n]


def sharded_knn_edges_dev(engine, d_states, k, d_edges, rank=None, world_size=None):
    """ShardedNeighbours with everything in HBM: this rank's rows of the k-nearest table (tr_knn_range_dev) into a device tensor, one
    all-gather of the int32 rows (equal shards, padding rows -1), the edge list of the whole table into d_edges (int32, capacity x 2;
    tr_knn_table_edges_dev) on every rank.  Returns the number of edges."""
    import torch
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    n = d_states.shape[0]
    start, stop, shard = shard_bounds(n, world_size, rank)
    n_real = max(0, min(stop, n) - start)
    rows = torch.full((shard, int(k)), -1, dtype=torch.int32, device=d_states.device)
    if n_real > 0:
        engine.knn_range_dev(d_states, n, start, n_real, k, rows)           # (the first n_real rows of the shard's block)
    table = allgather_mask(rows.reshape(-1))[: n * int(k)].contiguous()
    return engine.edges_from_knn_dev(table, n, k, d_edges)


def sharded_edge_verdicts_dev(engine, d_states, d_edges, n_edges, space_params, d_vertex_sig=None, rank=None, world_size=None):
    """ShardedEdgeValidator.run_indexed with the vertices and the edge list in HBM: this rank's contiguous shard of d_edges through
    tr_validate_edges_indexed_dev (with d_vertex_sig: _sig_dev, no vertex pass), the verdict words all-gathered as device tensors.
    Returns the mask words (int64 tensor on the device; unpack_bits(words, n_edges) is the mask)."""
    import torch
    import torch.distributed as dist
    if rank is None:
        rank = dist.get_rank() if dist.is_initialized() else 0
    if world_size is None:
        world_size = dist.get_world_size() if dist.is_initialized() else 1
    start, stop, shard = shard_bounds(n_edges, world_size, rank)
    n_real = max(0, min(stop, n_edges) - start)
    words = torch.zeros(shard // WORD, dtype=torch.int64, device=d_states.device)
    if n_real > 0:
        engine.validate_edges_indexed_dev(d_states, d_states.shape[0], d_edges[start:start + n_real], n_real, words, None, *space_params,
                                          d_vertex_sig=d_vertex_sig)
        if n_real % WORD:                                  # padding bits stay zero
            words[n_real // WORD] &= (1 << (n_real % WORD)) - 1
    return allgather_mask(words)


def pack_bits(mask):
    """bool[n] -> uint64 words, bit i & 63 of word i >> 6 (the layout of every verdict mask here)."""
    mask = np.asarray(mask, dtype=bool)
    pad = (-len(mask)) % WORD
    bits = np.concatenate([mask, np.zeros(pad, dtype=bool)]).reshape(-1, WORD)
    return (bits.astype(np.uint64) << np.arange(WORD, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)
