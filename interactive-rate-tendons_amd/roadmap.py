"""Batched forms of the roadmap-building loops of motion_planning::VoxelCachedLazyPRM that sit on
the hot path (motion-planning/VoxelCachedLazyPRM.cpp): rejection-sampled
vertex generation (createRoadmap phase 1, :1446-1455), k-nearest-neighbour edge lists (phase 3,
:1491-1502 -- on the host here; the GPU version is a "next" row), edge validation (phase 4,
:1508-1554), voxel caches (:1704-1713, :1751-1775) and their re-validation after the environment
changed (:2397-2411, :2497-2509).  Graph bookkeeping, A*, file formats stay with the reference.
"""
import time

import numpy as np

from . import _lib as L
from . import distributed as D
from .engine import unpack_bits


class RoadmapBuilder:
    def __init__(self, checker, motion_validator, seed=0, tau_max=None):
        self.checker, self.mv = checker, motion_validator
        self.robot, self.engine = checker.robot(), checker.engine
        self.seed, self.tau_max = seed, tau_max
        self.timing = {}

    # ---- createRoadmap phase 1: rejection sampling until N valid vertices ----------------------------
    def sample_valid_vertices(self, N, batch=None):
        """createRoadmap's vertex phase on the device (tr_sample_valid_vertices): the candidates are a counter-based sequence
        keyed by (seed, candidate index) -- distributed.candidate_states is its host mirror -- generated in HBM, validated by
        fk_verdict and compacted in candidate order on the GPU; only the accepted states and tips come back.  The accepted set
        is a deterministic prefix-filter of the sequence (independent of the batch sizes; `batch` is accepted and ignored)."""
        t0 = time.perf_counter()
        box = D.sampling_box(self.robot, self.tau_max)
        try:
            out = self.engine.sample_valid_vertices(N, seed=self.seed, box=box)
        except L.Unsupported:
            # a context on another schedule than the verdict-only one (TENDON_HIP_FUSED=0 / 1: A/B runs): the rejection loop over
            # batches of the SAME candidate sequence through validate_candidates -- the same accepted set, candidate by candidate
            out = self._sample_valid_vertices_batches(N, box, batch or (1 << 16))
        if out["accepted"] < N:
            raise RuntimeError("only %d of %d valid vertices after %d candidates" % (out["accepted"], N, out["tried"]))
        states, tips, tried = out["states"], out["tips"], out["tried"]
        if self.robot.enable_retraction:
            # Milestones are numbered by backbone length.  The numbering of a roadmap's vertices is free (the reference's is the
            # order of its addMilestone calls), and with this one every later batch over the roadmap is homogeneous wave by
            # wave: vertex v's neighbours in state space have similar retractions (the retraction term of the metric is the
            # heaviest, Problem.cpp:144-152), the edge list comes out ordered by vertex number, and the bisection hands its
            # samples on in groups of 64 neighbours -- so a wave of the retraction kernels (which runs from its LONGEST
            # backbone's base to the tip, shorter ones idling) holds backbones of one length in the stored-point forms
            # (voxel caches, connect) too, where no device-side ordering is applied.
            # (65 536 levels of the retraction range: a radix sort of 16-bit keys, 1 ms per 10^5 against 12 for the doubles)
            L_ = self.robot.specs.L
            order = np.argsort(np.clip(states[:, -1] * (65535.0 / L_), 0.0, 65535.0).astype(np.uint16), kind="stable")
            states, tips = np.ascontiguousarray(states[order]), np.ascontiguousarray(tips[order])
        self.timing["vertices"] = dict(seconds=time.perf_counter() - t0, candidates=tried, accepted=N)
        return states, tips

    def _sample_valid_vertices_batches(self, N, box, batch):
        states, tips, pos = [], [], 0
        have, tried, limit = 0, 0, 64 * N + (1 << 20)
        while have < N and pos < limit:
            m = min(batch, limit - pos)
            cand = D.candidate_states(self.robot, self.seed, pos, m, box=box)
            det = self.checker.is_valid_detail(cand)
            idx = np.flatnonzero(det["valid"])[: N - have]
            states.append(cand[idx]); tips.append(det["tips"][idx])
            tried = pos + (int(idx[-1]) + 1 if have + len(idx) >= N and len(idx) else m)
            have += len(idx); pos += m
        return dict(states=np.concatenate(states) if states else np.zeros((0, len(box[0]))), tips=np.concatenate(tips) if tips else np.zeros((0, 3)),
                    accepted=have, tried=tried if have >= N else pos)

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
        e = self.engine.knn_edges(states, k, max_distance).astype(np.int64)
        self.timing["knn_gpu"] = dict(seconds=time.perf_counter() - t0, edges=len(e))
        return e

    def build_on_device(self, n_vertices, k, max_distance=np.inf):
        """createRoadmap's phases 1 - 3 without leaving HBM (:1431-1560): n_vertices valid vertices from the device sampler, with their
        backbone signatures where the context can hand them over (tr_sample_valid_vertices_sig_dev), the k-nearest edge list
        (tr_knn_edges_dev; k counts the vertex itself) and checkMotion on every candidate edge (tr_validate_edges_indexed_sig_dev /
        _dev) -- only counts cross PCIe.  Returns torch tensors on the engine's GPU: d_states [n, S], d_edges [n_edges, 2] int32,
        d_valid_bits (int64 words; unpack_bits(words, n_edges)), and n_domain_errors, candidates_tried.  Same vertices, edges and
        verdicts as sample_valid_vertices -> knn_edges_gpu -> validate_edges -- except for retraction robots, whose vertices the host
        builder renumbers by backbone length (sample_valid_vertices) and this call leaves in acceptance order: the same vertex SET and
        the same edges between them, under different indices."""
        import torch
        t0 = time.perf_counter()
        eng, dev = self.engine, "cuda:%d" % self.engine.device
        n, S, sw = int(n_vertices), eng.state_size, eng.signature_words()
        d_states = torch.empty(n * S, dtype=torch.float64, device=dev)
        d_sig = torch.empty((n, sw), dtype=torch.int32, device=dev) if sw else None
        acc, tried = eng.sample_valid_vertices_dev(n, d_states, seed=self.seed, box=D.sampling_box(self.robot, self.tau_max), d_sig=d_sig)
        if acc < n:
            raise RuntimeError("only %d of %d valid vertices after %d candidates" % (acc, n, tried))
        d_edges = torch.empty((max(1, n * int(k)), 2), dtype=torch.int32, device=dev)
        ne = eng.knn_edges_dev(d_states, n, k, d_edges, max_distance)
        d_bits = torch.zeros((ne + 63) // 64, dtype=torch.int64, device=dev)
        nd = eng.validate_edges_indexed_dev(d_states, n, d_edges, ne, d_bits, None, self.mv.min_tension_change, self.mv.min_rotation_change,
                                            self.mv.min_retraction_change, d_vertex_sig=d_sig) if ne else 0
        torch.cuda.synchronize()
        self.timing["build_on_device"] = dict(seconds=time.perf_counter() - t0, vertices=n, edges=ne, candidates=tried)
        return dict(d_states=d_states.view(n, S), d_edges=d_edges[:ne], d_valid_bits=d_bits, n_edges=ne, n_domain_errors=nd,
                    candidates_tried=tried, signatures_handed_over=bool(sw))

    def knn_edges_star(self, states):
        """The PRM* connection strategy (setStarConnectionStrategy, VoxelCachedLazyPRM.cpp:1346-1356): k grows with the
        roadmap, k = ceil((e + e/dim) ln n); in createRoadmap all n vertices are in place before connecting."""
        return self.knn_edges_gpu(states, self.engine.kstar_k(len(states)) + 1)

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
    def vertex_caches(self, states, device=False):
        """voxelizeVertex for every vertex.  device=True: block ids / masks stay in HBM as torch tensors (for
        VoxelCachedLazyPRM.set_caches / DeviceCaches on the same GPU); offsets and flags are host arrays either way."""
        t0 = time.perf_counter()
        out = self.engine.voxelize_batch(states, device=device)
        self.timing["vertex_caches"] = dict(seconds=time.perf_counter() - t0, items=len(states), blocks=int(out["offsets"][-1]))
        return out

    def validate_edges_sharded(self, states, edges, device=None, d_states=None, d_vertex_sig=None):
        """checkMotion of the candidate edges over the ranks of the default process group (second phase): this rank
        validates its contiguous shard of the edge list with the indexed form (every vertex once per rank, two to four lanes), one
        all-gather of the verdict words; every rank returns the whole mask.  d_states (the vertex array already on this rank's GPU)
        and d_vertex_sig (the vertices' signature rows gathered with the vertex mask, ShardedVertexValidator.run_with_rows) select
        the device-resident form: with the signatures no rank integrates the vertices again."""
        import torch
        dev = device if device is not None else ("cuda:%d" % self.engine.device if torch.cuda.is_available() else "cpu")
        mv = self.mv
        if d_states is not None:
            nv = d_states.shape[0]

            def local(_, e_):
                d_e = torch.from_numpy(np.ascontiguousarray(e_, dtype=np.int32)).to(d_states.device)
                d_bits = torch.zeros((len(e_) + 63) // 64, dtype=torch.int64, device=d_states.device)
                self.engine.validate_edges_indexed_dev(d_states, nv, d_e, len(e_), d_bits, None, mv.min_tension_change, mv.min_rotation_change,
                                                       mv.min_retraction_change, d_vertex_sig=d_vertex_sig)
                return d_bits.cpu().numpy().view(np.uint64)
            return unpack_bits(D.ShardedEdgeValidator(local, device=dev).run_indexed(None, np.asarray(edges)), len(edges))
        st = np.ascontiguousarray(states, dtype=np.float64)
        sh = D.ShardedEdgeValidator(lambda s_, e_: D.pack_bits(self.engine.validate_edges_indexed(
            s_, e_, mv.min_tension_change, mv.min_rotation_change, mv.min_retraction_change)["valid"]), device=dev)
        return unpack_bits(sh.run_indexed(st, np.asarray(edges)), len(edges))

    def knn_edges_sharded(self, states, k, device=None):
        """knn_edges_gpu with the neighbour search spread over the ranks of the default process group (config 4's sizes:
        10^6 vertices are 10^12 pair distances): this rank's rows of the table (tr_knn_range), one all-gather of the rows,
        then the edge set from the whole table on every rank (tr_knn_table_edges).  Same edges as knn_edges_gpu."""
        t0 = time.perf_counter()
        st = np.ascontiguousarray(states, dtype=np.float64)
        dev = ("cuda:%d" % self.engine.device) if device is None else device
        sh = D.ShardedNeighbours(lambda first, count: self.engine.knn(st, k, query_range=(first, count))[0], k, device=dev)
        table = sh.run(len(st))
        edges = self.engine.edges_from_knn(table)
        self.timing["knn_sharded"] = dict(seconds=time.perf_counter() - t0, n=len(st), k=k, edges=len(edges))
        return edges

    def connect(self, states, edges, device=False):
        """createRoadmap's edge phase in one pass (tr_connect_edges_indexed): checkMotion on every candidate edge and the voxel
        sets of the accepted ones -> (accepted edges, their caches as CSR over the accepted edges only)."""
        t0 = time.perf_counter()
        out = self.engine.voxelize_edges_indexed(states, edges, self.mv.min_tension_change, self.mv.min_rotation_change,
                                                 self.mv.min_retraction_change, device=device, validate=True)
        keep = np.flatnonzero(out["fully_valid"])                      # (one index list for all four selections: boolean masks re-scan per use)
        off = out["offsets"]
        new_off = np.empty(len(keep) + 1, dtype=off.dtype)              # rejected edges own nothing: dropping them leaves the lists as they are
        np.take(off, keep, out=new_off[:-1])
        new_off[-1] = off[-1]
        out["offsets"] = new_off
        out["n_fk"] = np.take(out["n_fk"], keep)
        out["fully_valid"] = np.ones(len(keep), dtype=bool)
        e = np.asarray(edges).reshape(-1, 2)
        kept = np.take(e, keep, axis=0)
        self.timing["connect"] = dict(seconds=time.perf_counter() - t0, items=len(e), accepted=len(keep), blocks=int(off[-1]))
        return kept, out

    def create_roadmap(self, n_vertices, k=None, batch=1 << 17, device=True, n_landmarks=16):
        """createRoadmap (motion-planning/VoxelCachedLazyPRM.cpp:1431-1560) as one call: n_vertices valid milestones (:1446-1483),
        the connection loop with k nearest (k = None: the PRM* strategy's k, :1346-1356) keeping the edges checkMotion accepts
        (:1491-1551), every vertex's and every kept edge's voxel set (:1704-1713, :1751-1775), and the result attached to a
        VoxelCachedLazyPRM ready for solveWithRoadmap.  device=True keeps the block lists in HBM from the voxelisation to the
        query loop.  Returns (prm, dict(states, tips, edges, vertex_caches, edge_caches)); self.timing has the phases."""
        t0 = time.perf_counter()
        states, tips = self.sample_valid_vertices(n_vertices, batch=batch)
        kk = (self.engine.kstar_k(len(states)) if k is None else int(k)) + 1            # the tables count the vertex itself
        cand = self.knn_edges_gpu(states, kk)
        self.engine.reserve_edges(len(cand))
        edges, ec = self.connect(states, cand, device=device)
        vc = self.vertex_caches(states, device=device)
        prm = VoxelCachedLazyPRM(self.checker, states, edges)
        prm.set_caches(vc, ec)
        if n_landmarks:
            prm.prepare(n_landmarks)
        self.timing["create_roadmap"] = dict(seconds=time.perf_counter() - t0, vertices=len(states), candidate_edges=len(cand), edges=len(edges))
        return prm, dict(states=states, tips=tips, edges=edges, vertex_caches=vc, edge_caches=ec)

    def save_rmp(self, path, roadmap, weights=None):
        """Write the dict create_roadmap returns (states, tips, edges, vertex_caches, edge_caches; lists on the host or on the
        device) as a `.rmp` file the reference's loaders read (rmp.write_rmp)."""
        from . import rmp

        def host(c):
            if type(c["block_ids"]).__module__.startswith("torch"):
                c = dict(c, block_ids=c["block_ids"].cpu().numpy().view(np.uint32), masks=c["masks"].cpu().numpy().view(np.uint64))
            pres = next((c[k] for k in ("present", "shape_valid", "fully_valid") if k in c and c[k] is not None), None)
            return dict(offsets=c["offsets"], block_ids=c["block_ids"], masks=c["masks"],
                        present=np.ones(len(c["offsets"]) - 1, bool) if pres is None else np.asarray(pres, bool))
        st, e = roadmap["states"], roadmap["edges"]
        if weights is None:                                  # connectVertices stores the motion cost = state-space distance (:2857-2861)
            weights = self.engine.state_distance(st[e[:, 0]], st[e[:, 1]])
        vox = self.checker._voxels
        rmp.write_rmp(path, st, roadmap.get("tips"), e, weights, host(roadmap["vertex_caches"]), host(roadmap["edge_caches"]),
                      N=vox.Nx(), limits=vox.limits())

    def edge_caches(self, states, edges, device=False):
        t0 = time.perf_counter()
        out = self.engine.voxelize_edges_indexed(states, edges, self.mv.min_tension_change,
                                                 self.mv.min_rotation_change, self.mv.min_retraction_change, device=device)
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
        on_dev = isinstance(caches["block_ids"], torch.Tensor)
        self.ids = caches["block_ids"].to(dev) if on_dev else torch.from_numpy(caches["block_ids"].view(np.int32)).to(dev)
        self.masks = caches["masks"].to(dev) if on_dev else torch.from_numpy(caches["masks"].view(np.int64)).to(dev)
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


class _Paths:
    """The paths of a batch of queries as a sequence: paths[q] is query q's vertex indices, start ... goal (empty when it has none).
    A view on the packed arrays -- ten thousand slices are only made when someone asks for them."""

    def __init__(self, vertices, offsets):
        self._v, self._o = vertices, offsets

    def __len__(self):
        return len(self._o) - 1

    def __getitem__(self, q):
        if isinstance(q, slice):
            return [self[i] for i in range(*q.indices(len(self)))]
        q = int(q)
        if q < 0:
            q += len(self)
        if not 0 <= q < len(self):
            raise IndexError(q)
        return self._v[self._o[q]:self._o[q + 1]]

    def __iter__(self):
        return (self[q] for q in range(len(self)))

    def tolist(self):
        """The paths as a list of int32 arrays (what solve() returned before round 4)."""
        return [self[q] for q in range(len(self))]

    def __eq__(self, other):
        try:
            return len(self) == len(other) and all(np.array_equal(a, b) for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    __hash__ = None


class VoxelCachedLazyPRM:
    """The query side of motion_planning::VoxelCachedLazyPRM on a roadmap with voxel caches (BASELINE config 5):
    `solveWithRoadmap` (motion-planning/VoxelCachedLazyPRM.cpp:1977-2096 -> constructSolution :2689-2771) for a batch
    of (start, goal) roadmap vertices, `clearValidity` (:1656-1663), and `revalidate`, the eager form of the loading
    loops (:2397-2411, :2486-2526).  Graph search runs in the native library on the host cores, validity comes from
    the cached voxel sets resident in HBM through K4 (tr_roadmap_* in include/tendon_hip.h)."""

    def __init__(self, checker, states, edges, weights=None):
        import ctypes as C
        from . import _lib as L
        self._C, self._L = C, L
        self.engine = checker.engine
        self.checker = checker
        self.lib = L.lib()
        self.states = np.ascontiguousarray(states, dtype=np.float64)
        self.edges = np.ascontiguousarray(np.asarray(edges).reshape(-1, 2), dtype=np.int32)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float64)
        rm = C.c_void_p()
        st = self.lib.tr_roadmap_create(self.engine._ctx, self.states.ctypes.data_as(C.POINTER(C.c_double)), len(self.states),
                                        self.edges.ctypes.data_as(C.POINTER(C.c_int32)),
                                        w.ctypes.data_as(C.POINTER(C.c_double)) if w is not None else None, len(self.edges),
                                        C.byref(rm))
        if st != L.TR_OK:
            raise L._EXC.get(st, L.TendonHipError)("tr_roadmap_create failed (status %d)" % st)
        self._rm = rm
        self.stats = None
        self.search_stats = None
        self.search_profile = None

    @classmethod
    def from_rmp(cls, checker, path, n_landmarks=16):
        """A roadmap file with voxel caches (`.rmp`, rmp.read_rmp: the layout of RmpStreamer / LazyRmpParser) as a query
        object: what fromRoadmapParser builds (motion-planning/VoxelCachedLazyPRM.cpp:2357-2580), without the octrees."""
        from . import rmp
        d = rmp.read_rmp(path)
        if d["vertex_caches"] is None or d["edge_caches"] is None:
            raise ValueError("%s holds no voxel caches" % path)
        prm = cls(checker, d["states"], d["edges"], weights=d["weights"])
        prm.set_caches(d["vertex_caches"], d["edge_caches"])
        if n_landmarks:
            prm.prepare(n_landmarks)
        return prm

    def _check(self, st):
        if st != self._L.TR_OK:
            msg = self.lib.tr_roadmap_last_error(self._rm)
            raise self._L._EXC.get(st, self._L.TendonHipError)(msg.decode() if msg else "status %d" % st)

    def close(self):
        if getattr(self, "_rm", None) is not None:
            self.lib.tr_roadmap_destroy(self._rm)
            self._rm = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_caches(self, vertex_caches, edge_caches):
        """Upload vertexVoxelsProperty_ / edgeVoxelsProperty_ as CSR (dicts of offsets, block_ids, masks as returned by
        Engine.voxelize_batch / voxelize_edges or rmp.read_rmp; optional boolean 'shape_valid' / 'fully_valid' /
        'present' = which items have a cache at all)."""
        C = self._C
        from .distributed import pack_bits

        def on_device(c):
            return type(c["block_ids"]).__module__.startswith("torch")

        def arrs(c, keys):
            off = np.ascontiguousarray(c["offsets"], dtype=np.int64)
            present = next((c[k] for k in keys if k in c and c[k] is not None), None)
            pb = None if present is None else np.ascontiguousarray(pack_bits(np.asarray(present, dtype=bool)))
            if on_device(c):
                import torch
                ids, mk = c["block_ids"], c["masks"]
                if not (ids.is_cuda and mk.is_cuda and ids.device.index == self.engine.device and mk.device.index == self.engine.device
                        and ids.dtype == torch.int32 and mk.dtype == torch.int64 and ids.is_contiguous() and mk.is_contiguous()
                        and ids.numel() >= off[-1] and mk.numel() >= off[-1]):
                    raise self._L.InvalidArgument("device caches must be contiguous int32 / int64 tensors on the checker's GPU")
            else:
                ids = np.ascontiguousarray(c["block_ids"], dtype=np.uint32)
                mk = np.ascontiguousarray(c["masks"], dtype=np.uint64)
            return off, ids, mk, pb
        vo, vi, vm, vp = arrs(vertex_caches, ("present", "shape_valid"))
        eo, ei, em, ep = arrs(edge_caches, ("present", "fully_valid"))
        if len(vo) != len(self.states) + 1 or len(eo) != len(self.edges) + 1:
            raise self._L.InvalidArgument("cache offsets do not match the roadmap")
        u32, u64, i64 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
        if on_device(vertex_caches) != on_device(edge_caches):
            raise self._L.InvalidArgument("vertex and edge caches must both be host arrays or both device tensors")
        if on_device(vertex_caches):
            import torch
            torch.cuda.synchronize(self.engine.device)
            self._check(self.lib.tr_roadmap_set_caches_dev(
                self._rm, vo.ctypes.data_as(i64), C.c_void_p(vi.data_ptr()), C.c_void_p(vm.data_ptr()),
                vp.ctypes.data_as(u64) if vp is not None else None, eo.ctypes.data_as(i64), C.c_void_p(ei.data_ptr()),
                C.c_void_p(em.data_ptr()), ep.ctypes.data_as(u64) if ep is not None else None))
            return
        self._check(self.lib.tr_roadmap_set_caches(
            self._rm, vo.ctypes.data_as(i64), vi.ctypes.data_as(u32), vm.ctypes.data_as(u64),
            vp.ctypes.data_as(u64) if vp is not None else None, eo.ctypes.data_as(i64), ei.ctypes.data_as(u32),
            em.ctypes.data_as(u64), ep.ctypes.data_as(u64) if ep is not None else None))

    def set_obstacles(self, voxels, env=None):
        """A changed environment: new obstacle grid for the checker's engine, every validity unknown again."""
        self.engine.set_grid(voxels.Nx(), voxels.limits(), voxels.blocks, None if env is None else env.inv_rotation)
        self.clearValidity()

    def prepare(self, n_landmarks=16, n_threads=0):
        """Landmark lower bounds for the searches (tr_roadmap_prepare); 0 = the reference's heuristic alone."""
        self._check(self.lib.tr_roadmap_prepare(self._rm, int(n_landmarks), int(n_threads)))

    def clearValidity(self):
        self._check(self.lib.tr_roadmap_clear_validity(self._rm))

    def revalidate(self):
        """Every cached set against the current grid (one K4 launch) -> (#invalid vertices, #invalid edges)."""
        C = self._C
        nv, ne = C.c_int64(0), C.c_int64(0)
        self._check(self.lib.tr_roadmap_revalidate(self._rm, C.byref(nv), C.byref(ne)))
        return nv.value, ne.value

    def validity(self):
        """(vertex status, edge status): 0 unknown, 1 valid, 2 invalid / removed."""
        C = self._C
        v, e = np.zeros(len(self.states), dtype=np.uint8), np.zeros(len(self.edges), dtype=np.uint8)
        self._check(self.lib.tr_roadmap_get_validity(self._rm, v.ctypes.data_as(C.POINTER(C.c_uint8)), e.ctypes.data_as(C.POINTER(C.c_uint8))))
        return v, e

    def set_validity(self, vertex_status=None, edge_status=None):
        """What a builder already knows (tr_roadmap_set_validity): createRoadmap with ValidateVertices / ValidateEdges leaves its
        items VALIDITY_TRUE (motion-planning/VoxelCachedLazyPRM.cpp:1476, :2621-2631).  None = leave as it is."""
        C = self._C
        u8 = C.POINTER(C.c_uint8)
        v = None if vertex_status is None else np.ascontiguousarray(vertex_status, dtype=np.uint8)
        e = None if edge_status is None else np.ascontiguousarray(edge_status, dtype=np.uint8)
        if (v is not None and len(v) != len(self.states)) or (e is not None and len(e) != len(self.edges)):
            raise self._L.InvalidArgument("status arrays do not match the roadmap")
        self._check(self.lib.tr_roadmap_set_validity(self._rm, v.ctypes.data_as(u8) if v is not None else None,
                                                     e.ctypes.data_as(u8) if e is not None else None))

    def solveWithRoadmap(self, starts, goals, n_threads=0):
        """Batch of queries -> dict(status, cost, paths): paths[q] = vertex indices start ... goal (empty unless solved)."""
        C, L = self._C, self._L
        s = np.ascontiguousarray(starts, dtype=np.int32).reshape(-1)
        g = np.ascontiguousarray(goals, dtype=np.int32).reshape(-1)
        if s.shape != g.shape:
            raise L.InvalidArgument("starts and goals differ in length")
        n = len(s)
        status = np.zeros(n, dtype=np.int32)
        cost = np.zeros(n)
        off = np.zeros(n + 1, dtype=np.int64)
        st = L.TrRoadmapStats()
        i32 = C.POINTER(C.c_int32)
        self._check(self.lib.tr_roadmap_solve(self._rm, s.ctypes.data_as(i32), g.ctypes.data_as(i32), n, int(n_threads),
                                              status.ctypes.data_as(i32), cost.ctypes.data_as(C.POINTER(C.c_double)),
                                              off.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(st)))
        pv = np.zeros(int(off[-1]), dtype=np.int32)
        self._check(self.lib.tr_roadmap_fetch_paths(self._rm, pv.ctypes.data_as(i32), len(pv)))
        self.stats = dict(rounds=st.rounds, items_checked=st.items_checked, astar_runs=st.astar_runs, expanded=st.expanded)
        ss = (C.c_int64 * 8)()
        self._check(self.lib.tr_roadmap_search_stats(self._rm, ss))
        # where the searches ran (tr_roadmap_search_stats): finished by the kernel / handed back by it / on the host threads meanwhile
        self.search_stats = dict(device=int(ss[0]), handed_back=int(ss[1]), host_meanwhile=int(ss[2]), list_moves=int(ss[3]),
                                 expanded_device=int(ss[4]), expanded_host=int(ss[5]), answered_by_components=int(ss[6]), table_growths=int(ss[7]))
        pr = (C.c_double * 4)()
        self._check(self.lib.tr_roadmap_profile(self._rm, pr))
        # the roadmap_astar launches of this solve (HIP events): ms, launches, expansions, algorithmic bytes per expansion
        self.search_profile = dict(kernel_ms=float(pr[0]), launches=int(pr[1]), expansions=int(pr[2]), bytes_per_expansion=float(pr[3]))
        return dict(status=status, cost=cost, path_offsets=off, path_vertices=pv, paths=_Paths(pv, off))

    def search_state_bytes(self):
        """Device memory the graph searches hold for this roadmap between calls (tr_roadmap_search_state_bytes)."""
        b = self._C.c_int64(0)
        self._check(self.lib.tr_roadmap_search_state_bytes(self._rm, self._C.byref(b)))
        return int(b.value)

    def search_sweeps(self):
        """Searches of the last solve answered by the device sweep instead of A* (tr_roadmap_search_sweeps)."""
        b = self._C.c_int64(0)
        self._check(self.lib.tr_roadmap_search_sweeps(self._rm, self._C.byref(b)))
        return int(b.value)

    def reserve_search_state(self, n_queries):
        """Sets the device searches up for rounds of up to n_queries queries now (tr_roadmap_reserve_search_state) instead of inside the first solve."""
        self._check(self.lib.tr_roadmap_reserve_search_state(self._rm, int(n_queries)))

    def release_search_state(self):
        """Hands the searches' tables back to the device (tr_roadmap_release_search_state); the next large round allocates them again."""
        b = self._C.c_int64(0)
        self._check(self.lib.tr_roadmap_release_search_state(self._rm, self._C.byref(b)))
        return int(b.value)
