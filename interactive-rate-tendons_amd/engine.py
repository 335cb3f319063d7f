"""Thin object wrapper over one tr_ctx (robot constants + obstacle grid resident on one GPU).

Device buffers are torch tensors (torch is the memory / stream plumbing, nothing more); every
compute call goes through the C ABI in include/tendon_hip.h.
"""
import ctypes as C

import numpy as np

from . import _lib as L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _torch():
    import torch
    return torch


def unpack_bits(words, n):
    """uint64 validity words -> bool[n] (bit i&63 of word i>>6)."""
    w = np.ascontiguousarray(words, dtype=np.uint64)
    return np.unpackbits(w.view(np.uint8), bitorder="little")[:n].astype(bool)


class Engine:
    """One context of libtendon_hip.so: tr_create ... tr_destroy."""

    def __init__(self, robot, device=0):
        self._ctx = None
        lib = L.lib()
        n = len(robot.tendons)
        if n == 0:
            raise L.OutOfRange("robot has no tendons")
        n_a, n_m = len(robot.tendons[0].C), len(robot.tendons[0].D)
        for t in robot.tendons:
            if len(t.C) != n_a or len(t.D) != n_m:
                raise L.InvalidArgument("all tendons must share C.size() and D.size() (get_r_info.cpp:112-115)")
        self._C = _f64([t.C for t in robot.tendons]).reshape(n, n_a)
        self._D = _f64([t.D for t in robot.tendons]).reshape(n, n_m)
        self._maxt = _f64([t.max_tension for t in robot.tendons])
        self._minl = _f64([t.min_length for t in robot.tendons])
        self._maxl = _f64([t.max_length for t in robot.tendons])
        d = L.TrRobotDesc()
        s = robot.specs
        d.r, d.L, d.dL, d.ro, d.ri, d.E, d.nu = robot.r, s.L, s.dL, s.ro, s.ri, s.E, s.nu
        d.n_tendons, d.n_a, d.n_m = n, n_a, n_m
        d.C, d.D = _dp(self._C), _dp(self._D)
        d.max_tension, d.min_length, d.max_length = _dp(self._maxt), _dp(self._minl), _dp(self._maxl)
        d.enable_rotation, d.enable_retraction = int(robot.enable_rotation), int(robot.enable_retraction)
        d.residual_threshold = robot.residual_threshold
        ctx = C.c_void_p()
        st = lib.tr_create(C.byref(d), int(device), C.byref(ctx))
        if st != L.TR_OK:
            L.check(None, st)
        self._ctx = ctx
        self.lib = lib
        self.device = int(device)
        self.n_tendons = n
        self.state_size = lib.tr_state_size(ctx)
        self.num_points = lib.tr_num_points(ctx)
        self.has_grid = False

    def close(self):
        if self._ctx is not None:
            self.lib.tr_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- setup -------------------------------------------------------------------------------
    def home_lengths(self):
        out = np.zeros(self.n_tendons)
        L.check(self._ctx, self.lib.tr_home_lengths(self._ctx, _dp(out)))
        return out

    def set_grid(self, N, limits, blocks, inv_rot=None):
        lim = _f64(limits).reshape(6)
        blk = np.ascontiguousarray(blocks, dtype=np.uint64).reshape(-1)
        if blk.size != (N // 4) ** 3:
            raise L.InvalidArgument("voxel dimension mismatch (%d blocks for N=%d)" % (blk.size, N))
        rot = None if inv_rot is None else _f64(inv_rot).reshape(9)
        L.check(self._ctx, self.lib.tr_set_grid(self._ctx, int(N), _dp(lim),
                                                blk.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                _dp(rot) if rot is not None else None))
        self.has_grid = True
        self._grid_nb = (int(N) // 4) ** 3

    # ---- environment preparation on the resident grid (collision::VoxelOctree's editing operations) ----
    def grid_add_spheres(self, spheres):
        sp = _f64(spheres).reshape(-1, 4)
        L.check(self._ctx, self.lib.tr_grid_add_spheres(self._ctx, _dp(sp), sp.shape[0]))

    def grid_add_capsules(self, capsules):
        cp = _f64(capsules).reshape(-1, 7)
        L.check(self._ctx, self.lib.tr_grid_add_capsules(self._ctx, _dp(cp), cp.shape[0]))

    def grid_remove_interior(self, keep_diagonal=True):
        L.check(self._ctx, self.lib.tr_grid_remove_interior(self._ctx, int(bool(keep_diagonal))))

    def grid_dilate(self, num=1, use_diagonal=False):
        L.check(self._ctx, self.lib.tr_grid_dilate(self._ctx, int(num), int(bool(use_diagonal))))

    def grid_dilate_sphere(self, r):
        L.check(self._ctx, self.lib.tr_grid_dilate_sphere(self._ctx, float(r)))

    def get_grid(self):
        """uint64[Nb^3]: the obstacle blocks as they are on the device now."""
        blk = np.empty(self._grid_nb, dtype=np.uint64)
        L.check(self._ctx, self.lib.tr_get_grid(self._ctx, blk.ctypes.data_as(C.POINTER(C.c_uint64))))
        return blk

    def set_checker(self, spheres):
        """False: VoxelBackboneValidityChecker (default); True: VoxelValidityChecker (sphere-swept robot)."""
        L.check(self._ctx, self.lib.tr_set_checker(self._ctx, L.TR_CHECKER_SPHERES if spheres else L.TR_CHECKER_BACKBONE))

    def reserve(self, n):
        L.check(self._ctx, self.lib.tr_reserve(self._ctx, int(n)))

    def reserve_edges(self, n_edges):
        """Pre-size the sample pool / device frontier of the edge calls."""
        L.check(self._ctx, self.lib.tr_reserve_edges(self._ctx, int(n_edges)))

    def set_debug(self, bits):
        L.check(self._ctx, self.lib.tr_set_debug(self._ctx, int(bits)))

    def edge_schedule_last(self):
        """How the last indexed edge call was scheduled: dict(samples, rounds, exact_sweep, flags) of the edge queue (samples == 0:
        the level-synchronous lanes took the call)."""
        st = (C.c_uint32 * 4)()
        L.check(self._ctx, self.lib.tr_edge_schedule_last(self._ctx, st))
        return dict(samples=int(st[0]), rounds=int(st[1]), exact_sweep=int(st[2]), flags=int(st[3]))

    # ---- host-buffer calls ---------------------------------------------------------------------
    def _states(self, states):
        st = _f64(states)
        if st.ndim == 1:
            st = st.reshape(1, -1)
        if st.ndim != 2 or st.shape[1] != self.state_size:
            raise L.InvalidArgument("State is not the right size")       # TendonRobot.h:107-109
        return st

    def fk_batch(self, states, want_R=False):
        st = self._states(states)
        n, P, N = st.shape[0], self.num_points, self.n_tendons
        p = np.empty((n, P, 3))
        R = np.empty((n, P, 9)) if want_R else None
        Lb, Li = np.empty(n), np.empty((n, N))
        conv = np.empty(n, dtype=np.uint8)
        npts = np.empty(n, dtype=np.int32)
        L.check(self._ctx, self.lib.tr_fk_batch(self._ctx, _dp(st), n, _dp(p), _dp(R) if want_R else None,
                                                _dp(Lb), _dp(Li), conv.ctypes.data_as(C.POINTER(C.c_uint8)),
                                                npts.ctypes.data_as(C.POINTER(C.c_int32))))
        return dict(p=p, R=R, L=Lb, L_i=Li, converged=conv.astype(bool), n_points=npts)

    def validate_batch(self, states, want_tips=True, want_flags=True):
        st = self._states(states)
        n = st.shape[0]
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        tips = np.empty((n, 3)) if want_tips else None
        flags = np.empty(n, dtype=np.uint8) if want_flags else None
        L.check(self._ctx, self.lib.tr_validate_batch(
            self._ctx, _dp(st), n, bits.ctypes.data_as(C.POINTER(C.c_uint64)),
            _dp(tips) if want_tips else None,
            flags.ctypes.data_as(C.POINTER(C.c_uint8)) if want_flags else None))
        return dict(valid=unpack_bits(bits, n), bits=bits, tips=tips, flags=flags)

    def validate_edges(self, a, b, min_tension_change=0.02, min_rotation_change=0.01,
                       min_retraction_change=0.0001):
        a, b = self._states(a), self._states(b)
        if a.shape != b.shape:
            raise L.InvalidArgument("start and end are different sizes")   # VoxelEnvironment.cpp:227-229
        n = a.shape[0]
        sp = L.TrSpaceParams(min_tension_change, min_rotation_change, min_retraction_change)
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        nfk = np.zeros(n, dtype=np.int32)
        nde = C.c_int64(0)
        L.check(self._ctx, self.lib.tr_validate_edges(
            self._ctx, C.byref(sp), _dp(a), _dp(b), n, bits.ctypes.data_as(C.POINTER(C.c_uint64)),
            nfk.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nde)))
        return dict(valid=unpack_bits(bits, n), bits=bits, n_fk=nfk, n_domain_errors=nde.value)

    def validate_edges_indexed(self, states, edges, min_tension_change=0.02, min_rotation_change=0.01, min_retraction_change=0.0001):
        """Roadmap form: edges (n_edges, 2) index rows of states; every vertex is evaluated once for all its edges."""
        st = self._states(states)
        e = np.ascontiguousarray(np.asarray(edges).reshape(-1, 2), dtype=np.int32)
        n = e.shape[0]
        sp = L.TrSpaceParams(min_tension_change, min_rotation_change, min_retraction_change)
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        nfk = np.zeros(n, dtype=np.int32)
        nd = C.c_int64(0)
        L.check(self._ctx, self.lib.tr_validate_edges_indexed(
            self._ctx, C.byref(sp), _dp(st), st.shape[0], e.ctypes.data_as(C.POINTER(C.c_int32)), n,
            bits.ctypes.data_as(C.POINTER(C.c_uint64)), nfk.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(nd)))
        return dict(valid=unpack_bits(bits, n), n_fk=nfk, n_domain_errors=int(nd.value))

    def validate_edges_last_valid(self, a, b, min_tension_change=0.02, min_rotation_change=0.01,
                                  min_retraction_change=0.0001):
        a, b = self._states(a), self._states(b)
        if a.shape != b.shape:
            raise L.InvalidArgument("start and end are different sizes")
        n = a.shape[0]
        sp = L.TrSpaceParams(min_tension_change, min_rotation_change, min_retraction_change)
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        lvt = np.zeros(n)
        nfk = np.zeros(n, dtype=np.int32)
        L.check(self._ctx, self.lib.tr_validate_edges_last_valid(
            self._ctx, C.byref(sp), _dp(a), _dp(b), n, bits.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(lvt),
            nfk.ctypes.data_as(C.POINTER(C.c_int32))))
        return dict(valid=unpack_bits(bits, n), last_valid_t=lvt, n_fk=nfk)

    def validate_edges_discrete(self, a, b, min_tension_change=0.02, min_rotation_change=0.01,
                                min_retraction_change=0.0001, last_valid=True):
        """last_valid = True: checkMotion(s1, s2, last_valid) (the installed state checker judges every sample);
        False: checkMotion(s1, s2) (shape validity + swept backbone volume); the two differ under the sphere checker only."""
        a, b = self._states(a), self._states(b)
        if a.shape != b.shape:
            raise L.InvalidArgument("start and end are different sizes")
        n = a.shape[0]
        sp = L.TrSpaceParams(min_tension_change, min_rotation_change, min_retraction_change)
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        lvt = np.zeros(n)
        nfk = np.zeros(n, dtype=np.int32)
        L.check(self._ctx, self.lib.tr_validate_edges_discrete(
            self._ctx, C.byref(sp), _dp(a), _dp(b), n, bits.ctypes.data_as(C.POINTER(C.c_uint64)),
            _dp(lvt) if last_valid else None, nfk.ctypes.data_as(C.POINTER(C.c_int32))))
        return dict(valid=unpack_bits(bits, n), last_valid_t=lvt if last_valid else None, n_fk=nfk)

    def check_cached(self, block_ids, masks, offsets):
        ids = np.ascontiguousarray(block_ids, dtype=np.uint32)
        mk = np.ascontiguousarray(masks, dtype=np.uint64)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        n = off.size - 1
        bits = np.zeros((max(n, 0) + 63) // 64, dtype=np.uint64)
        L.check(self._ctx, self.lib.tr_check_cached(
            self._ctx, ids.ctypes.data_as(C.POINTER(C.c_uint32)), mk.ctypes.data_as(C.POINTER(C.c_uint64)),
            off.ctypes.data_as(C.POINTER(C.c_int64)), n, bits.ctypes.data_as(C.POINTER(C.c_uint64))))
        return unpack_bits(bits, n)

    def knn(self, states, k, max_distance=np.inf, query_range=None):
        """k nearest (self included, as OMPL's nearestK on a populated structure) in the state-space metric.
        query_range = (first, count): only those rows of the table (all states stay candidates) -- one rank's share."""
        st = self._states(states)
        n = st.shape[0]
        q0, nq = (0, n) if query_range is None else (int(query_range[0]), int(query_range[1]))
        idx = np.empty((nq, k), dtype=np.int32)
        dist = np.empty((nq, k))
        if query_range is None:
            L.check(self._ctx, self.lib.tr_knn(self._ctx, _dp(st), n, int(k), float(max_distance),
                                               idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(dist)))
        else:
            L.check(self._ctx, self.lib.tr_knn_range(self._ctx, _dp(st), n, q0, nq, int(k), float(max_distance),
                                                     idx.ctypes.data_as(C.POINTER(C.c_int32)), _dp(dist)))
        return idx, dist

    def edges_from_knn(self, idx):
        """Undirected, deduplicated, ordered edge list of a k-nearest table given as an (n, k) index array (tr_knn_table_edges)."""
        t = np.ascontiguousarray(idx, dtype=np.int32)
        if t.ndim != 2:
            raise L.InvalidArgument("idx must be (n, k)")
        n, k = t.shape
        cap = max(1, n * k)
        e = np.empty((cap, 2), dtype=np.int32)
        ne = C.c_int64(0)
        L.check(self._ctx, self.lib.tr_knn_table_edges(self._ctx, t.ctypes.data_as(C.POINTER(C.c_int32)), n, k,
                                                       e.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(ne)))
        return e[: min(cap, ne.value)].copy()

    def knn_edges(self, states, k, max_distance=np.inf):
        """Undirected edge list (n_edges, 2), lo < hi, ordered, of the k-nearest table (k counts the vertex itself):
        the connection loop of createRoadmap, deduplicated on the device."""
        st = self._states(states)
        n = st.shape[0]
        cap = max(1, n * int(k))              # n (k - 1) unless k or more states coincide (a row then need not hold its own vertex)
        e = np.empty((cap, 2), dtype=np.int32)
        ne = C.c_int64(0)
        L.check(self._ctx, self.lib.tr_knn_edges(self._ctx, _dp(st), n, int(k), float(max_distance),
                                                 e.ctypes.data_as(C.POINTER(C.c_int32)), cap, C.byref(ne)))
        return e[: min(cap, ne.value)].copy()

    def space_weights(self):
        """(w_rotation, w_retraction) of the planner's compound state space (tr_space_weights; Problem.cpp:131-152)."""
        wr, ws = C.c_double(0), C.c_double(0)
        L.check(self._ctx, self.lib.tr_space_weights(self._ctx, C.byref(wr), C.byref(ws)))
        return wr.value, ws.value

    def state_distance(self, a, b):
        """OMPL's CompoundStateSpace::distance for rows of a and b as Problem.cpp:101-163 wires it: Euclidean norm of the
        tensions (weight 1) + w_rotation x shortest SO2 arc + w_retraction x |delta s| -- the cost connectVertices stores on an
        edge (VoxelCachedLazyPRM.cpp:2857-2861) and the metric of tr_knn."""
        a, b = _f64(a).reshape(-1, self.state_size), _f64(b).reshape(-1, self.state_size)
        n = self.n_tendons
        _, rot, ret = self.state_layout()
        wr, ws = self.space_weights()
        d = np.sqrt(((a[:, :n] - b[:, :n]) ** 2).sum(axis=1))
        k = n
        if rot:
            arc = np.abs(a[:, k] - b[:, k])
            d = d + wr * np.where(arc > np.pi, 2.0 * np.pi - arc, arc)
            k += 1
        if ret:
            d = d + ws * np.abs(a[:, k] - b[:, k])
        return d

    def state_layout(self):
        n, rot, ret = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        L.check(self._ctx, self.lib.tr_state_layout(self._ctx, C.byref(n), C.byref(rot), C.byref(ret)))
        return n.value, bool(rot.value), bool(ret.value)

    def kstar_k(self, n_milestones):
        """k of og::KStarStrategy for a roadmap of n_milestones vertices (PRM*)."""
        k = self.lib.tr_kstar_k(self._ctx, int(n_milestones))
        if k < 0:
            raise L.InvalidArgument("n_milestones must be positive")
        return k

    def _fetch_lists(self, nnz, device=False):
        """Block lists of the last voxelize call: numpy arrays, or (device=True) torch tensors on this engine's GPU (int32 /
        int64 views of the uint32 ids / uint64 masks) that never crossed PCIe."""
        if device:
            torch = _torch()
            dev = "cuda:%d" % self.device
            ids = torch.empty(nnz, dtype=torch.int32, device=dev)
            masks = torch.empty(nnz, dtype=torch.int64, device=dev)
            L.check(self._ctx, self.lib.tr_voxelize_fetch_dev(self._ctx, C.c_void_p(ids.data_ptr()), C.c_void_p(masks.data_ptr()), nnz,
                                                              self._stream_ptr(None)))
            return ids, masks
        ids = np.empty(nnz, dtype=np.uint32)
        masks = np.empty(nnz, dtype=np.uint64)
        L.check(self._ctx, self.lib.tr_voxelize_fetch(self._ctx, ids.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                      masks.ctypes.data_as(C.POINTER(C.c_uint64)), nnz))
        return ids, masks

    def voxelize_batch(self, states, device=False):
        """voxelizeVertex for a batch: CSR (offsets, block_ids, masks), shape validity, tips."""
        st = self._states(states)
        n = st.shape[0]
        offsets = np.zeros(n + 1, dtype=np.int64)
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        tips = np.empty((n, 3))
        L.check(self._ctx, self.lib.tr_voxelize_batch(self._ctx, _dp(st), n, offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                                      bits.ctypes.data_as(C.POINTER(C.c_uint64)), _dp(tips)))
        ids, masks = self._fetch_lists(int(offsets[-1]), device)
        return dict(offsets=offsets, block_ids=ids, masks=masks, shape_valid=unpack_bits(bits, n), tips=tips)

    def voxelize_edges(self, a, b, min_tension_change=0.02, min_rotation_change=0.01, min_retraction_change=0.0001, device=False):
        """voxelizeEdge for a batch: swept-volume block lists of the fully valid edges."""
        a, b = self._states(a), self._states(b)
        if a.shape != b.shape:
            raise L.InvalidArgument("start and end are different sizes")
        n = a.shape[0]
        sp = L.TrSpaceParams(min_tension_change, min_rotation_change, min_retraction_change)
        offsets = np.zeros(n + 1, dtype=np.int64)
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        nfk = np.zeros(n, dtype=np.int32)
        L.check(self._ctx, self.lib.tr_voxelize_edges(self._ctx, C.byref(sp), _dp(a), _dp(b), n,
                                                      offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                                      bits.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                      nfk.ctypes.data_as(C.POINTER(C.c_int32))))
        ids, masks = self._fetch_lists(int(offsets[-1]), device)
        return dict(offsets=offsets, block_ids=ids, masks=masks, fully_valid=unpack_bits(bits, n), n_fk=nfk)

    def voxelize_edges_indexed(self, states, edges, min_tension_change=0.02, min_rotation_change=0.01, min_retraction_change=0.0001,
                               device=False, validate=False):
        """voxelizeEdge for roadmap edges given as index pairs: every vertex integrated and voxelised once.  validate=True is
        tr_connect_edges_indexed: the same samples are also tested against the obstacles (checkMotion), 'fully_valid' is then
        checkMotion's verdict and only valid edges own a voxel set."""
        st = self._states(states)
        e = np.ascontiguousarray(np.asarray(edges).reshape(-1, 2), dtype=np.int32)
        n = e.shape[0]
        sp = L.TrSpaceParams(min_tension_change, min_rotation_change, min_retraction_change)
        offsets = np.zeros(n + 1, dtype=np.int64)
        bits = np.zeros((n + 63) // 64, dtype=np.uint64)
        nfk = np.zeros(n, dtype=np.int32)
        fn = self.lib.tr_connect_edges_indexed if validate else self.lib.tr_voxelize_edges_indexed
        L.check(self._ctx, fn(
            self._ctx, C.byref(sp), _dp(st), st.shape[0], e.ctypes.data_as(C.POINTER(C.c_int32)), n,
            offsets.ctypes.data_as(C.POINTER(C.c_int64)), bits.ctypes.data_as(C.POINTER(C.c_uint64)),
            nfk.ctypes.data_as(C.POINTER(C.c_int32))))
        ids, masks = self._fetch_lists(int(offsets[-1]), device)
        return dict(offsets=offsets, block_ids=ids, masks=masks, fully_valid=unpack_bits(bits, n), n_fk=nfk)

    # ---- device-buffer calls (torch tensors on this engine's GPU) -------------------------------
    @staticmethod
    def _stream_ptr(stream):
        if stream is None:
            stream = _torch().cuda.current_stream()
        return C.c_void_p(stream.cuda_stream)

    def _check_dev(self, t, dtype, min_numel, name):
        torch = _torch()
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == dtype and t.is_contiguous()
                and t.device.index == self.device and t.numel() >= min_numel):
            raise L.InvalidArgument("%s must be a contiguous %s tensor on cuda:%d with >= %d elements"
                                    % (name, dtype, self.device, min_numel))
        return C.c_void_p(t.data_ptr())

    def validate_batch_dev(self, d_states, n, d_bits, d_tips=None, d_flags=None, stream=None):
        torch = _torch()
        ps = self._check_dev(d_states, torch.float64, n * self.state_size, "d_states")
        pb = self._check_dev(d_bits, torch.int64, (n + 63) // 64, "d_bits")
        pt = self._check_dev(d_tips, torch.float64, 3 * n, "d_tips") if d_tips is not None else None
        pf = self._check_dev(d_flags, torch.uint8, n, "d_flags") if d_flags is not None else None
        L.check(self._ctx, self.lib.tr_validate_batch_dev(self._ctx, ps, int(n), pb, pt, pf, self._stream_ptr(stream)))

    # ---- createRoadmap phase 1 on the device (tr_sample_valid_vertices & co, include/tendon_hip.h) ----------------------
    @staticmethod
    def _box(box):
        if box is None:
            return None, None, None
        lo, hi = _f64(box[0]), _f64(box[1])
        return (lo, hi), _dp(lo), _dp(hi)

    def candidate_states(self, seed, first, count, box=None):
        """Candidates [first, first + count) of the sequence `seed`, generated on the device, as a host array."""
        keep, plo, phi = self._box(box)
        out = np.empty((int(count), self.state_size))
        L.check(self._ctx, self.lib.tr_candidate_states(self._ctx, int(seed), int(first), int(count), plo, phi, _dp(out)))
        return out

    def candidate_states_dev(self, seed, first, count, d_states, box=None, stream=None):
        keep, plo, phi = self._box(box)
        ps = self._check_dev(d_states, _torch().float64, count * self.state_size, "d_states")
        L.check(self._ctx, self.lib.tr_candidate_states_dev(self._ctx, int(seed), int(first), int(count), plo, phi, ps, self._stream_ptr(stream)))

    def validate_candidates_dev(self, seed, first, count, d_bits, d_tips=None, d_flags=None, box=None, stream=None):
        """tr_validate_batch_dev on device-generated candidates [first, first + count): nothing is uploaded, the mask stays in HBM."""
        torch = _torch()
        keep, plo, phi = self._box(box)
        pb = self._check_dev(d_bits, torch.int64, (count + 63) // 64, "d_bits")
        pt = self._check_dev(d_tips, torch.float64, 3 * count, "d_tips") if d_tips is not None else None
        pf = self._check_dev(d_flags, torch.uint8, count, "d_flags") if d_flags is not None else None
        L.check(self._ctx, self.lib.tr_validate_candidates_dev(self._ctx, int(seed), int(first), int(count), plo, phi, pb, pt, pf,
                                                               self._stream_ptr(stream)))

    def signature_words(self):
        """uint32 words per row of the vertex signature arrays (tr_signature_words); 0: this context cannot hand signatures over."""
        return int(self.lib.tr_signature_words(self._ctx))

    def signature_packed_words(self):
        """uint32 words of a signature row as it travels between ranks (tr_signature_packed_words: first cell + 6 bits per further point)."""
        return int(self.lib.tr_signature_packed_words(self._ctx))

    def pack_signatures_dev(self, d_sig, stream=None):
        """Signature rows [n, signature_words()] (int32, on this GPU) -> (packed rows [n, signature_packed_words()], rows that could not
        be coded: the caller sends the rows as they are if there is one).  tr_pack_signatures_dev; synchronises the stream."""
        torch = _torch()
        n = int(d_sig.shape[0])
        out = torch.empty((n, self.signature_packed_words()), dtype=torch.int32, device=d_sig.device)
        bad = C.c_int64(0)
        L.check(self._ctx, self.lib.tr_pack_signatures_dev(self._ctx, C.c_void_p(d_sig.data_ptr()), n, C.c_void_p(out.data_ptr()), C.byref(bad),
                                                           self._stream_ptr(stream)))
        return out, int(bad.value)

    def unpack_signatures_dev(self, d_packed, stream=None):
        """The inverse: packed rows [n, signature_packed_words()] -> signature rows [n, signature_words()] (tr_unpack_signatures_dev; the
        padding words of a row beyond the backbone's points are left as allocated: nothing reads them)."""
        torch = _torch()
        n = int(d_packed.shape[0])
        out = torch.empty((n, self.signature_words()), dtype=torch.int32, device=d_packed.device)
        L.check(self._ctx, self.lib.tr_unpack_signatures_dev(self._ctx, C.c_void_p(d_packed.data_ptr()), n, C.c_void_p(out.data_ptr()),
                                                             self._stream_ptr(stream)))
        return out

    def validate_candidates_sig_dev(self, seed, first, count, d_bits, d_sig, d_tips=None, box=None, stream=None):
        """validate_candidates_dev that also writes every candidate's backbone cell signature (d_sig: int32 tensor, count x
        signature_words()): compacted like the accepted states, the rows spare validate_edges_indexed_dev its vertex pass."""
        torch = _torch()
        keep, plo, phi = self._box(box)
        pb = self._check_dev(d_bits, torch.int64, (count + 63) // 64, "d_bits")
        pt = self._check_dev(d_tips, torch.float64, 3 * count, "d_tips") if d_tips is not None else None
        psig = self._check_dev(d_sig, torch.int32, count * max(1, self.signature_words()), "d_sig")
        L.check(self._ctx, self.lib.tr_validate_candidates_sig_dev(self._ctx, int(seed), int(first), int(count), plo, phi, pb, pt, psig,
                                                                   self._stream_ptr(stream)))

    def knn_edges_dev(self, d_states, n, k, d_edges, max_distance=np.inf):
        """tr_knn_edges with the states in HBM and the edge list left there (d_edges: int32 tensor, capacity x 2): returns n_edges."""
        torch = _torch()
        ps = self._check_dev(d_states, torch.float64, n * self.state_size, "d_states")
        cap = d_edges.numel() // 2
        pe = self._check_dev(d_edges, torch.int32, 2 * cap, "d_edges")
        ne = C.c_int64(0)
        L.check(self._ctx, self.lib.tr_knn_edges_dev(self._ctx, ps, int(n), int(k), float(max_distance), pe, cap, C.byref(ne)))
        return int(ne.value)

    def knn_range_dev(self, d_states, n, first_query, n_queries, k, d_idx, max_distance=np.inf):
        """tr_knn_range with the states in HBM: rows first_query .. of the k-nearest table into d_idx (int32, n_queries x k, device)."""
        torch = _torch()
        ps = self._check_dev(d_states, torch.float64, n * self.state_size, "d_states")
        pi = self._check_dev(d_idx, torch.int32, n_queries * int(k), "d_idx")
        L.check(self._ctx, self.lib.tr_knn_range_dev(self._ctx, ps, int(n), int(first_query), int(n_queries), int(k), float(max_distance), pi))

    def edges_from_knn_dev(self, d_table, n, k, d_edges):
        """tr_knn_table_edges on a table in HBM (int32, n x k), the edge list written into d_edges (int32, capacity x 2): returns n_edges."""
        torch = _torch()
        pt = self._check_dev(d_table, torch.int32, n * int(k), "d_table")
        cap = d_edges.numel() // 2
        pe = self._check_dev(d_edges, torch.int32, 2 * cap, "d_edges")
        ne = C.c_int64(0)
        L.check(self._ctx, self.lib.tr_knn_table_edges_dev(self._ctx, pt, int(n), int(k), pe, cap, C.byref(ne)))
        return int(ne.value)

    def validate_edges_indexed_dev(self, d_states, n_states, d_edges, n_edges, d_bits, d_n_fk=None, min_tension_change=0.02,
                                   min_rotation_change=0.01, min_retraction_change=0.0001, d_vertex_sig=None):
        """tr_validate_edges_indexed on device arrays (vertex states, index pairs, mask words, optional FK counts): returns the
        number of domain errors.  d_vertex_sig (int32, n_states x signature_words()): the vertices' signature rows from
        validate_candidates_sig_dev -- every vertex is then taken to be valid and none is integrated again."""
        torch = _torch()
        ps = self._check_dev(d_states, torch.float64, n_states * self.state_size, "d_states")
        pe = self._check_dev(d_edges, torch.int32, 2 * n_edges, "d_edges")
        pb = self._check_dev(d_bits, torch.int64, (n_edges + 63) // 64, "d_bits")
        pn = self._check_dev(d_n_fk, torch.int32, n_edges, "d_n_fk") if d_n_fk is not None else None
        sp = L.TrSpaceParams(min_tension_change, min_rotation_change, min_retraction_change)
        nd = C.c_int64(0)
        if d_vertex_sig is not None:
            pv = self._check_dev(d_vertex_sig, torch.int32, n_states * max(1, self.signature_words()), "d_vertex_sig")
            L.check(self._ctx, self.lib.tr_validate_edges_indexed_sig_dev(self._ctx, C.byref(sp), ps, int(n_states), pv, pe, int(n_edges), pb, pn, C.byref(nd)))
        else:
            L.check(self._ctx, self.lib.tr_validate_edges_indexed_dev(self._ctx, C.byref(sp), ps, int(n_states), pe, int(n_edges), pb, pn, C.byref(nd)))
        return int(nd.value)

    def compact_rows_dev(self, d_mask, count, d_rows, row_doubles, d_rows_out, capacity, d_index_out=None, stream=None):
        """Rows of d_rows whose mask bit is set, in order, into d_rows_out (at most capacity); returns the number of set bits."""
        torch = _torch()
        pm = self._check_dev(d_mask, torch.int64, (count + 63) // 64, "d_mask")
        pr = self._check_dev(d_rows, torch.float64, count * row_doubles, "d_rows") if row_doubles else None
        po = self._check_dev(d_rows_out, torch.float64, capacity * row_doubles, "d_rows_out") if row_doubles else None
        pi = self._check_dev(d_index_out, torch.int64, capacity, "d_index_out") if d_index_out is not None else None
        n_out = C.c_int64(0)
        L.check(self._ctx, self.lib.tr_compact_rows_dev(self._ctx, pm, int(count), pr, int(row_doubles), int(capacity), po, pi, C.byref(n_out),
                                                        self._stream_ptr(stream)))
        return n_out.value

    def sample_valid_vertices(self, n_want, seed=0, first_candidate=0, box=None, max_candidates=0, want_tips=True, want_index=False):
        """The first n_want valid candidates of the sequence, in candidate order: generated, validated and compacted on the GPU."""
        keep, plo, phi = self._box(box)
        n_want = int(n_want)
        states = np.empty((n_want, self.state_size))
        tips = np.empty((n_want, 3)) if want_tips else None
        index = np.empty(n_want, dtype=np.int64) if want_index else None
        n_acc, n_tried = C.c_int64(0), C.c_int64(0)
        L.check(self._ctx, self.lib.tr_sample_valid_vertices(
            self._ctx, int(seed), int(first_candidate), plo, phi, n_want, int(max_candidates), _dp(states),
            _dp(tips) if want_tips else None, index.ctypes.data_as(C.POINTER(C.c_int64)) if want_index else None,
            C.byref(n_acc), C.byref(n_tried)))
        k = n_acc.value
        return dict(states=states[:k], tips=tips[:k] if want_tips else None, index=index[:k] if want_index else None,
                    accepted=k, tried=n_tried.value)

    def sample_valid_vertices_dev(self, n_want, d_states, d_tips=None, d_index=None, seed=0, first_candidate=0, box=None,
                                  max_candidates=0, stream=None, d_sig=None):
        """tr_sample_valid_vertices_dev; with d_sig (int32, n_want x signature_words()) also the accepted vertices' signature rows
        (tr_sample_valid_vertices_sig_dev), which validate_edges_indexed_dev takes as d_vertex_sig."""
        torch = _torch()
        keep, plo, phi = self._box(box)
        ps = self._check_dev(d_states, torch.float64, n_want * self.state_size, "d_states")
        pt = self._check_dev(d_tips, torch.float64, 3 * n_want, "d_tips") if d_tips is not None else None
        pi = self._check_dev(d_index, torch.int64, n_want, "d_index") if d_index is not None else None
        n_acc, n_tried = C.c_int64(0), C.c_int64(0)
        if d_sig is not None:
            psig = self._check_dev(d_sig, torch.int32, n_want * max(1, self.signature_words()), "d_sig")
            L.check(self._ctx, self.lib.tr_sample_valid_vertices_sig_dev(self._ctx, int(seed), int(first_candidate), plo, phi, int(n_want),
                                                                         int(max_candidates), ps, pt, pi, psig, C.byref(n_acc), C.byref(n_tried),
                                                                         self._stream_ptr(stream)))
            return n_acc.value, n_tried.value
        L.check(self._ctx, self.lib.tr_sample_valid_vertices_dev(self._ctx, int(seed), int(first_candidate), plo, phi, int(n_want),
                                                                 int(max_candidates), ps, pt, pi, C.byref(n_acc), C.byref(n_tried),
                                                                 self._stream_ptr(stream)))
        return n_acc.value, n_tried.value

    def fk_batch_dev(self, d_states, n, ld, d_px, d_py, d_pz, d_L=None, d_Li=None, d_conv=None, d_R=None,
                     d_npts=None, stream=None):
        torch = _torch()
        P = self.num_points
        ps = self._check_dev(d_states, torch.float64, n * self.state_size, "d_states")
        px = self._check_dev(d_px, torch.float64, P * ld, "d_px")
        py = self._check_dev(d_py, torch.float64, P * ld, "d_py")
        pz = self._check_dev(d_pz, torch.float64, P * ld, "d_pz")
        pR = self._check_dev(d_R, torch.float64, 9 * P * ld, "d_R") if d_R is not None else None
        pL = self._check_dev(d_L, torch.float64, n, "d_L") if d_L is not None else None
        pLi = self._check_dev(d_Li, torch.float64, self.n_tendons * ld, "d_Li") if d_Li is not None else None
        pc = self._check_dev(d_conv, torch.uint8, n, "d_conv") if d_conv is not None else None
        pn = self._check_dev(d_npts, torch.int32, n, "d_npts") if d_npts is not None else None
        L.check(self._ctx, self.lib.tr_fk_batch_dev(self._ctx, ps, int(n), int(ld), px, py, pz, pR, pL, pLi, pc, pn,
                                                    self._stream_ptr(stream)))

    def fk_batch_retraction_dev(self, d_states, n, ld, d_px, d_py, d_pz, d_Li, d_conv, d_npts, d_home_Li, d_L=None, stream=None):
        """Retraction robots: FK with per-configuration point counts and home lengths (rows aligned at the tip)."""
        torch = _torch()
        P, N = self.num_points, self.n_tendons
        L.check(self._ctx, self.lib.tr_fk_batch_retraction_dev(
            self._ctx, self._check_dev(d_states, torch.float64, n * self.state_size, "d_states"), int(n), int(ld),
            self._check_dev(d_px, torch.float64, P * ld, "d_px"), self._check_dev(d_py, torch.float64, P * ld, "d_py"),
            self._check_dev(d_pz, torch.float64, P * ld, "d_pz"), None,
            self._check_dev(d_L, torch.float64, n, "d_L") if d_L is not None else None,
            self._check_dev(d_Li, torch.float64, N * ld, "d_Li"), self._check_dev(d_conv, torch.uint8, n, "d_conv"),
            self._check_dev(d_npts, torch.int32, n, "d_npts"), self._check_dev(d_home_Li, torch.float64, N * ld, "d_home_Li"),
            self._stream_ptr(stream)))

    def validate_shapes_retraction_dev(self, n, ld, d_px, d_py, d_pz, d_npts, d_Li, d_home_Li, d_conv, d_bits, d_flags=None,
                                       check_voxels=True, stream=None):
        torch = _torch()
        P, N = self.num_points, self.n_tendons
        L.check(self._ctx, self.lib.tr_validate_shapes_retraction_dev(
            self._ctx, int(n), int(ld), self._check_dev(d_px, torch.float64, P * ld, "d_px"),
            self._check_dev(d_py, torch.float64, P * ld, "d_py"), self._check_dev(d_pz, torch.float64, P * ld, "d_pz"),
            self._check_dev(d_npts, torch.int32, n, "d_npts"), self._check_dev(d_Li, torch.float64, N * ld, "d_Li"),
            self._check_dev(d_home_Li, torch.float64, N * ld, "d_home_Li"), self._check_dev(d_conv, torch.uint8, n, "d_conv"),
            int(bool(check_voxels)), self._check_dev(d_bits, torch.int64, (n + 63) // 64, "d_bits"),
            self._check_dev(d_flags, torch.uint8, n, "d_flags") if d_flags is not None else None, self._stream_ptr(stream)))

    def validate_shapes_dev(self, n, ld, d_px, d_py, d_pz, d_Li, d_conv, d_bits, d_flags=None, d_npts=None,
                            check_voxels=True, stream=None):
        torch = _torch()
        P = self.num_points
        px = self._check_dev(d_px, torch.float64, P * ld, "d_px")
        py = self._check_dev(d_py, torch.float64, P * ld, "d_py")
        pz = self._check_dev(d_pz, torch.float64, P * ld, "d_pz")
        pLi = self._check_dev(d_Li, torch.float64, self.n_tendons * ld, "d_Li")
        pc = self._check_dev(d_conv, torch.uint8, n, "d_conv")
        pb = self._check_dev(d_bits, torch.int64, (n + 63) // 64, "d_bits")
        pf = self._check_dev(d_flags, torch.uint8, n, "d_flags") if d_flags is not None else None
        pn = self._check_dev(d_npts, torch.int32, n, "d_npts") if d_npts is not None else None
        L.check(self._ctx, self.lib.tr_validate_shapes_dev(self._ctx, int(n), int(ld), px, py, pz, pn, pLi, pc,
                                                           int(bool(check_voxels)), pb, pf, self._stream_ptr(stream)))

    def check_cached_dev(self, d_ids, d_masks, d_offsets, n_items, d_bits, stream=None):
        torch = _torch()
        pi = self._check_dev(d_ids, torch.int32, 0, "d_ids")
        pm = self._check_dev(d_masks, torch.int64, 0, "d_masks")
        po = self._check_dev(d_offsets, torch.int64, n_items + 1, "d_offsets")
        pb = self._check_dev(d_bits, torch.int64, (n_items + 63) // 64, "d_bits")
        L.check(self._ctx, self.lib.tr_check_cached_dev(self._ctx, pi, pm, po, int(n_items), pb,
                                                        self._stream_ptr(stream)))

    # ---- instrumentation -----------------------------------------------------------------------
    def profile_begin(self):
        L.check(self._ctx, self.lib.tr_profile_begin(self._ctx))

    def profile_read(self):
        n = (C.c_int64 * L.TR_PROFILE_SLOTS)()
        ms = (C.c_double * L.TR_PROFILE_SLOTS)()
        L.check(self._ctx, self.lib.tr_profile_read(self._ctx, n, ms))
        return {name: dict(launches=int(n[i]), total_ms=float(ms[i])) for i, name in enumerate(L.PROFILE_SLOT_NAMES)}

    def profile_end(self):
        L.check(self._ctx, self.lib.tr_profile_end(self._ctx))
