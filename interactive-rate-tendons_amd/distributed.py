"""Multi-GPU sharding of roadmap vertex validation (BASELINE config 4): one process per GPU,
contiguous vertex shards, no data-path collective inside the FK/collision kernels, and ONE
all-gather of the packed validity bitmask (RCCL over xGMI; `gloo` in CPU tests) before the
host connects edges -- the exchange step the reference's single-process createRoadmap gets for
free from shared memory (motion-planning/VoxelCachedLazyPRM.cpp:1446-1483: parallel vertex
validation, then serial addMilestone over ALL vertices).

Candidate vertices come from a counter-keyed generator (chunk index -> stream), so the candidate
set and therefore the gathered mask are identical for every world size.
"""
import numpy as np

WORD = 64
RNG_CHUNK = 1 << 16


def shard_bounds(M, world_size, rank):
    """Contiguous equal shards of M items padded so every shard is a whole number of 64-bit mask
    words (all-gather needs equal counts).  Returns (start, stop, shard_len); stop may exceed M
    for the padding tail, whose verdict bits are forced to 0."""
    words = (M + WORD - 1) // WORD
    words_per_rank = (words + world_size - 1) // world_size
    shard = words_per_rank * WORD
    start = rank * shard
    return start, start + shard, shard


def candidate_states(robot, seed, start, count, tau_max=None):
    """States [start, start+count) of the global candidate sequence keyed by (seed, chunk)."""
    S = robot.state_size()
    out = np.empty((count, S))
    pos = start
    while pos < start + count:
        chunk = pos // RNG_CHUNK
        lo = chunk * RNG_CHUNK
        rng = np.random.default_rng([int(seed), int(chunk)])
        block = np.empty((RNG_CHUNK, S))
        k = 0
        for t in robot.tendons:
            block[:, k] = rng.uniform(0.0, t.max_tension if tau_max is None else tau_max, RNG_CHUNK)
            k += 1
        if robot.enable_rotation:
            block[:, k] = rng.uniform(-np.pi, np.pi, RNG_CHUNK)
            k += 1
        if robot.enable_retraction:
            block[:, k] = rng.uniform(0.0, robot.specs.L, RNG_CHUNK)
        a = pos - lo
        b = min(RNG_CHUNK, start + count - lo)
        out[pos - start: pos - start + (b - a)] = block[a:b]
        pos += b - a
    return out


def allgather_mask(local_words, group=None):
    """All-gather equal-length int64 mask shards into the global mask (rank-major)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return local_words.clone()
    world = dist.get_world_size(group)
    out = torch.empty(world * local_words.numel(), dtype=local_words.dtype, device=local_words.device)
    dist.all_gather_into_tensor(out, local_words.contiguous(), group=group)
    return out


class ShardedVertexValidator:
    """Validate M candidate vertices across the ranks of the default process group.

    validate_local(states) -> uint64/int64 mask words for that shard (bit i&63 of word i>>6).
    In production this is Engine.validate_batch_dev on the rank's GPU; the CPU tests plug in the
    oracle to exercise the sharding and the collective with the `gloo` backend.
    """

    def __init__(self, robot, validate_local, seed=0, tau_max=None, device="cpu"):
        self.robot, self.validate_local = robot, validate_local
        self.seed, self.tau_max, self.device = seed, tau_max, device

    def run(self, M, rank=None, world_size=None):
        import torch
        import torch.distributed as dist
        if rank is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
        if world_size is None:
            world_size = dist.get_world_size() if dist.is_initialized() else 1
        start, stop, shard = shard_bounds(M, world_size, rank)
        n_real = max(0, min(stop, M) - start)
        words = np.zeros(shard // WORD, dtype=np.uint64)
        if n_real > 0:
            states = candidate_states(self.robot, self.seed, start, n_real, self.tau_max)
            w = np.asarray(self.validate_local(states)).view(np.uint64)
            words[: w.size] = w
            if n_real % WORD:                              # padding bits stay zero
                words[n_real // WORD] &= np.uint64((1 << (n_real % WORD)) - 1)
        local = torch.from_numpy(words.view(np.int64)).to(self.device)
        full = allgather_mask(local)
        # world_size * shard/64 words; bits of items >= M are zero.  unpack_bits(words, M) is the mask.
        return full.cpu().numpy().view(np.uint64)


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


def pack_bits(mask):
    """bool[n] -> uint64 words, bit i & 63 of word i >> 6 (the layout of every verdict mask here)."""
    mask = np.asarray(mask, dtype=bool)
    pad = (-len(mask)) % WORD
    bits = np.concatenate([mask, np.zeros(pad, dtype=bool)]).reshape(-1, WORD)
    return (bits.astype(np.uint64) << np.arange(WORD, dtype=np.uint64)).sum(axis=1, dtype=np.uint64)
