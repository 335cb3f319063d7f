"""Batched forms of the roadmap-building loops of motion_planning::VoxelCachedLazyPRM that sit on
the hot path (motion-planning/VoxelCachedLazyPRM.cpp): rejection-sampled
vertex generation (createRoadmap phase 1, :1446-1455), k-nearest-neighbour edge lists (phase 3,
:1491-1502 -- on the host here; the GPU version is a "next" row), edge validation (phase 4,
:1508-1554), voxel caches (:1704-1713, :1751-1775) and their re-validation after the environment
changed (:2397-2411, :2497-2509).  Graph bookkeeping, A*, file formats stay with the reference.
"""
import time

import numpy as np

from . import distributed as D
from .engine import unpack_bits


class RoadmapBuilder:
    def __init__(self, checker, motion_validator, seed=0, tau_max=None):
        self.checker, self.mv = checker, motion_validator
        self.robot, self.engine = checker.robot(), checker.engine
        self.seed, self.tau_max = seed, tau_max
        self.timing = {}

    # ---- createRoadmap phase 1: rejection sampling until N valid vertices ----------------------------
    def sample_valid_vertices(self, N, batch=1 << 16):
        """Candidates come from the counter-keyed sequence of distributed.candidate_states, so the
        accepted set is a deterministic prefix-filter of it (independent of batch size)."""
        t0 = time.perf_counter()
        states, tips, pos, tried = [], [], 0, 0
        have = 0
        while have < N:
            cand = D.candidate_states(self.robot, self.seed, pos, batch, self.tau_max)
            out = self.engine.validate_batch(cand, True, False)
            ok = out["valid"]
            take = np.flatnonzero(ok)
            if have + take.size > N:
                take = take[: N - have]
                tried += int(take[-1]) + 1
            else:
                tried += batch
            states.append(cand[take]); tips.append(out["tips"][take])
            have += take.size
            pos += batch
        self.timing["vertices"] = dict(seconds=time.perf_counter() - t0, candidates=tried, accepted=N)
        return np.concatenate(states), np.concatenate(tips)

    # ---- phase 3: k nearest neighbours in state space (host) -------------------------------------------
    def state_space_metric_scale(self):
        """Per-dimension weights of the reference's compound space (Problem.cpp:112-152): tension 1,
        retraction 2*extent/L.  (Rotation is an SO2 component: handled by the caller if enabled.)"""
        ext = np.sqrt(sum(t.max_tension ** 2 for t in self.robot.tendons))
        w = [1.0] * len(self.robot.tendons)
        if self.robot.enable_rotation:
            w.append(ext / (4 * np.pi))
        if self.robot.enable_retraction:
            w.append(2 * ext / self.robot.specs.L)
        return np.array(w)

    def knn_edges_gpu(self, states, k, max_distance=np.inf):
        """connectionStrategy_(v) for every vertex on the GPU (tr_knn; the reference's metric incl.
        rotation / retraction weights), then the undirected edge set the reference's
        `if (!getEdge(v, n)) connectVertices(v, n)` loop builds (:1491-1502): k counts v itself."""
        t0 = time.perf_counter()
        idx, dist = self.engine.knn(states, k, max_distance)
        src = np.repeat(np.arange(len(states), dtype=np.int64), k)
        dst = idx.reshape(-1).astype(np.int64)
        keep = (dst >= 0) & (dst != src)
        lo, hi = np.minimum(src[keep], dst[keep]), np.maximum(src[keep], dst[keep])
        key = np.unique(lo * len(states) + hi)
        e = np.stack([key // len(states), key % len(states)], 1)
        self.timing["knn_gpu"] = dict(seconds=time.perf_counter() - t0, edges=len(e))
        return e

    def knn_edges(self, states, k):
        """Undirected k-NN edge list (i < j).  Distances are the compound-space sums of per-subspace
        norms; for tension-only robots that is the Euclidean norm, which cKDTree handles exactly."""
        from scipy.spatial import cKDTree
        t0 = time.perf_counter()
        if self.robot.enable_rotation or self.robot.enable_retraction:
            raise NotImplementedError("host k-NN is provided for tension-only state spaces")
        tree = cKDTree(states)
        _, idx = tree.query(states, k=k + 1)
        src = np.repeat(np.arange(len(states)), k)
        dst = idx[:, 1:].reshape(-1)
        e = np.stack([np.minimum(src, dst), np.maximum(src, dst)], 1)
        e = np.unique(e[e[:, 0] != e[:, 1]], axis=0)
        self.timing["knn"] = dict(seconds=time.perf_counter() - t0, edges=len(e))
        return e

    # ---- phase 4: edge validity --------------------------------------------------------------------------
    def validate_edges(self, states, edges):
        t0 = time.perf_counter()
        out = self.mv.check_motion_indexed(states, edges)          # vertices evaluated once for all their edges
        self.timing["edges"] = dict(seconds=time.perf_counter() - t0, edges=len(edges), fk_samples=int(out["n_fk"].sum()))
        return out["valid"], out["n_fk"]

    # ---- voxel caches and their re-validation --------------------------------------------------------------
    def vertex_caches(self, states):
        t0 = time.perf_counter()
        out = self.engine.voxelize_batch(states)
        self.timing["vertex_caches"] = dict(seconds=time.perf_counter() - t0, items=len(states), blocks=int(out["offsets"][-1]))
        return out

    def edge_caches(self, states, edges):
        t0 = time.perf_counter()
        out = self.engine.voxelize_edges(states[edges[:, 0]], states[edges[:, 1]], self.mv.min_tension_change,
                                         self.mv.min_rotation_change, self.mv.min_retraction_change)
        self.timing["edge_caches"] = dict(seconds=time.perf_counter() - t0, items=len(edges), blocks=int(out["offsets"][-1]))
        return out

    def revalidate(self, caches, new_obstacles, env=None):
        """fromRoadmapParser's loops: cached voxel sets vs a (changed) obstacle grid -> hit mask."""
        inv = None if env is None else env.inv_rotation
        self.engine.set_grid(new_obstacles.Nx(), new_obstacles.limits(), new_obstacles.blocks, inv)
        t0 = time.perf_counter()
        hit = self.engine.check_cached(caches["block_ids"], caches["masks"], caches["offsets"])
        self.timing["revalidate"] = dict(seconds=time.perf_counter() - t0, items=len(caches["offsets"]) - 1)
        return hit


class DeviceCaches:
    """Roadmap voxel caches resident in HBM (CSR of block ids / masks): uploaded once, re-validated
    with one K4 launch each time the obstacle grid changes -- the interactive loop of BASELINE config 5."""

    def __init__(self, engine, caches):
        import torch
        self.engine = engine
        dev = "cuda:%d" % engine.device
        self.n = len(caches["offsets"]) - 1
        self.ids = torch.from_numpy(caches["block_ids"].view(np.int32)).to(dev)
        self.masks = torch.from_numpy(caches["masks"].view(np.int64)).to(dev)
        self.offsets = torch.from_numpy(np.ascontiguousarray(caches["offsets"], dtype=np.int64)).to(dev)
        self.bits = torch.zeros((self.n + 63) // 64, dtype=torch.int64, device=dev)

    def revalidate(self, new_obstacles=None, env=None, sync=True):
        """hit mask (bool[n]) of every cached set against the current (or a new) obstacle grid."""
        import torch
        if new_obstacles is not None:
            self.engine.set_grid(new_obstacles.Nx(), new_obstacles.limits(), new_obstacles.blocks,
                                 None if env is None else env.inv_rotation)
        self.engine.check_cached_dev(self.ids, self.masks, self.offsets, self.n, self.bits)
        if not sync:
            return None
        torch.cuda.synchronize()
        return unpack_bits(self.bits.cpu().numpy().view(np.uint64), self.n)


def gathered_vertex_mask(robot, validate_bits_dev, M, seed, tau_max, device):
    """BASELINE config 4: validate M candidate vertices sharded over the ranks of the default process
    group and all-gather the validity bitmask (distributed.ShardedVertexValidator with the GPU engine
    as the local validator)."""
    v = D.ShardedVertexValidator(robot, validate_bits_dev, seed=seed, tau_max=tau_max, device=device)
    return unpack_bits(v.run(M), M)
