"""Host-side mirror of collision::VoxelOctree (cpp/src/collision/VoxelOctree.h:68-330) as the
engine consumes it: a DENSE array of 64-bit blocks (4x4x4 voxels each, bit x*16+y*4+z,
VoxelOctree.cpp:1501-1503), block index ((bx*Nb)+by)*Nb+bz.  Only what the hot path needs on
the host: limits, cell/block access, rasterising an obstacle environment (points, spheres), the
sparse (block id, mask) export used for roadmap voxel caches.  Robot voxelisation and the
octree-vs-octree test run on the GPU (sweep_kernel.hpp).
"""
import numpy as np

from . import _lib as L

_SUPPORTED = (4, 8, 16, 32, 64, 128, 256, 512)


def leaf_order_of(block_ids, Nb):
    """Permutation that puts dense block ids ((bx * Nb + by) * Nb + bz) into the order `TreeNode::visit_leaves` reaches
    them (collision/detail/TreeNode.hxx:176-190: per octree level bx outermost, then by, then bz, i.e. ascending by the key
    that interleaves the bits of bx, by, bz from the top, bx first).  It is the order of the "data" array of the voxel
    files (VoxelOctree.cpp:1357-1363) and of a roadmap file's block records (VoxelCachedLazyPRM.cpp:633-642)."""
    ids = np.asarray(block_ids, dtype=np.int64)
    bx, by, bz = ids // (Nb * Nb), (ids // Nb) % Nb, ids % Nb
    key = np.zeros(len(ids), dtype=np.int64)
    for bit in range(max(int(Nb).bit_length() - 1, 0)):
        key |= (((bx >> bit) & 1) << (3 * bit + 2)) | (((by >> bit) & 1) << (3 * bit + 1)) | (((bz >> bit) & 1) << (3 * bit))
    return np.argsort(key, kind="stable")


class VoxelOctree:
    def __init__(self, Ndim=4):
        if Ndim not in _SUPPORTED:                         # VoxelOctree.cpp:98-116
            raise L.InvalidArgument("unsupported voxel dimension: %d" % Ndim)
        self._N = int(Ndim)
        nb = self._N // 4
        self.blocks = np.zeros((nb, nb, nb), dtype=np.uint64)
        self.set_xlim(0.0, 1.0)
        self.set_ylim(0.0, 1.0)
        self.set_zlim(0.0, 1.0)

    @staticmethod
    def to_supported_size(Ndim):                           # VoxelOctree.cpp:82-96
        for s in _SUPPORTED:
            if Ndim <= s:
                return s
        raise L.InvalidArgument("too large for supported voxel octree: %d" % Ndim)

    # sizes ------------------------------------------------------------------------------------
    def Nx(self): return self._N
    def Ny(self): return self._N
    def Nz(self): return self._N
    def N(self): return self._N ** 3
    def Nbx(self): return self._N // 4
    def Nb(self): return self._N ** 3 // 64

    # limits (VoxelOctree.cpp:152-177) ------------------------------------------------------------
    def _set(self, axis, lo, hi):
        if lo >= hi:
            raise L.LengthError("%slimits must be positive in size" % axis)
        setattr(self, "_%smin" % axis, float(lo))
        setattr(self, "_%smax" % axis, float(hi))
        setattr(self, "_d%s" % axis, (float(hi) - float(lo)) / self._N)

    def set_xlim(self, lo, hi=None):
        self._set("x", *(lo if hi is None else (lo, hi)))

    def set_ylim(self, lo, hi=None):
        self._set("y", *(lo if hi is None else (lo, hi)))

    def set_zlim(self, lo, hi=None):
        self._set("z", *(lo if hi is None else (lo, hi)))

    def xlim(self): return (self._xmin, self._xmax)
    def ylim(self): return (self._ymin, self._ymax)
    def zlim(self): return (self._zmin, self._zmax)
    def limits(self): return (self._xmin, self._xmax, self._ymin, self._ymax, self._zmin, self._zmax)
    def dx(self): return self._dx
    def dy(self): return self._dy
    def dz(self): return self._dz

    def copy_limits(self, other):
        for a in ("_xmin", "_xmax", "_ymin", "_ymax", "_zmin", "_zmax", "_dx", "_dy", "_dz"):
            setattr(self, a, getattr(other, a))

    def empty_copy(self):                                  # VoxelOctree.cpp:146-150
        c = VoxelOctree(self._N)
        c.copy_limits(self)
        return c

    def __eq__(self, other):
        return (isinstance(other, VoxelOctree) and self._N == other._N and self.limits() == other.limits()
                and np.array_equal(self.blocks, other.blocks))

    # cells / blocks ---------------------------------------------------------------------------------
    @staticmethod
    def bitmask(x, y, z):                                  # VoxelOctree.cpp:1501-1503
        return np.uint64(1) << np.uint64(x * 16 + y * 4 + z)

    def block(self, bx, by, bz):
        return int(self.blocks[bx, by, bz])

    def set_block(self, bx, by, bz, value):
        self.blocks[bx, by, bz] = np.uint64(value)

    def cell(self, ix, iy, iz):
        return bool(self.blocks[ix // 4, iy // 4, iz // 4] & self.bitmask(ix % 4, iy % 4, iz % 4))

    def set_cell(self, ix, iy, iz, value=True):            # VoxelOctree.cpp:256-265
        m = self.bitmask(ix % 4, iy % 4, iz % 4)
        old = self.blocks[ix // 4, iy // 4, iz // 4]
        self.blocks[ix // 4, iy // 4, iz // 4] = (old | m) if value else (old & ~m)
        return bool(old & m) if value else bool(old & ~m)

    def is_empty(self): return not self.blocks.any()
    def nblocks(self): return int(np.count_nonzero(self.blocks))

    def ncells(self):
        return int(np.unpackbits(self.blocks.view(np.uint8)).sum())

    def is_in_domain(self, x, y, z):                       # VoxelOctree.cpp:1505-1509 (closed)
        return (self._xmin <= x <= self._xmax) and (self._ymin <= y <= self._ymax) and (self._zmin <= z <= self._zmax)

    def nearest_cell(self, x, y, z):                       # VoxelOctree.cpp:295-307 (truncate, clamp)
        m = self._N - 1
        f = lambda v, lo, d: min(m, max(0, int((v - lo) / d)))
        return f(x, self._xmin, self._dx), f(y, self._ymin, self._dy), f(z, self._zmin, self._dz)

    def find_cell(self, x, y, z):                          # VoxelOctree.cpp:309-317, 1511-1521
        for v, lo, hi, name in ((x, self._xmin, self._xmax, "x"), (y, self._ymin, self._ymax, "y"),
                                (z, self._zmin, self._zmax, "z")):
            if v < lo or hi < v:
                raise L.DomainError("%s is out of the voxel dimensions" % name)
        return (int((x - self._xmin) / self._dx), int((y - self._ymin) / self._dy), int((z - self._zmin) / self._dz))

    def _nearest_block_idx(self, x, y, z):                 # VoxelOctree.cpp:281-293
        m = self._N // 4 - 1
        f = lambda v, lo, d: min(m, max(0, int(int((v - lo) / d) / 4)))
        return f(x, self._xmin, self._dx), f(y, self._ymin, self._dy), f(z, self._zmin, self._dz)

    # rasterisation of the obstacle environment -----------------------------------------------------
    def add_point(self, p):                                # VoxelOctree.cpp:319-323
        x, y, z = map(float, p)
        if self.is_in_domain(x, y, z):
            self.set_cell(*self.nearest_cell(x, y, z))

    def add_sphere(self, c, r):
        """Voxel centres inside the sphere (VoxelOctree.cpp:434-469; |c - p|^2 <= r^2, collision.hxx:65-68)."""
        c = np.asarray(c, dtype=np.float64)
        r = float(r)
        self.add_point(c)
        lo = self._nearest_block_idx(*(c - r))
        hi = self._nearest_block_idx(*(c + r))
        ix = np.arange(lo[0] * 4, hi[0] * 4 + 4)
        iy = np.arange(lo[1] * 4, hi[1] * 4 + 4)
        iz = np.arange(lo[2] * 4, hi[2] * 4 + 4)
        x = self._xmin + self._dx * (ix + 0.5)
        y = self._ymin + self._dy * (iy + 0.5)
        z = self._zmin + self._dz * (iz + 0.5)
        ddx, ddy, ddz = c[0] - x, c[1] - y, c[2] - z
        d2 = ((ddx * ddx)[:, None, None] + (ddy * ddy)[None, :, None]) + (ddz * ddz)[None, None, :]
        inside = d2 <= r * r
        nbx, nby, nbz = len(ix) // 4, len(iy) // 4, len(iz) // 4
        cells = inside.reshape(nbx, 4, nby, 4, nbz, 4).transpose(0, 2, 4, 1, 3, 5).reshape(nbx, nby, nbz, 64)
        w = (np.uint64(1) << np.arange(64, dtype=np.uint64))
        masks = (cells.astype(np.uint64) * w).sum(axis=-1, dtype=np.uint64)
        self.blocks[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] |= masks

    def add_capsule(self, a, b, r):
        """Voxel centres within r of the segment a-b, plus the end points' cells (VoxelOctree.cpp:471-515;
        collides(Capsule, Point), collision.hxx:83-87)."""
        a, b, r = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), float(r)
        self.add_point(a); self.add_point(b)
        lo = self._nearest_block_idx(*(np.minimum(a, b) - r))
        hi = self._nearest_block_idx(*(np.maximum(a, b) + r))
        ix, iy, iz = (np.arange(lo[d] * 4, hi[d] * 4 + 4) for d in range(3))
        X = (self._xmin + self._dx * (ix + 0.5))[:, None, None]
        Y = (self._ymin + self._dy * (iy + 0.5))[None, :, None]
        Z = (self._zmin + self._dz * (iz + 0.5))[None, None, :]
        d = b - a
        dsq = (d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]
        eps = np.finfo(np.float64).eps
        if dsq <= eps * eps:
            t = np.zeros(np.broadcast_shapes(X.shape, Y.shape, Z.shape))
        else:
            t = ((d[0] * (X - a[0]) + d[1] * (Y - a[1])) + d[2] * (Z - a[2])) / dsq
        t = np.maximum(0.0, np.minimum(1.0, t))
        e0, e1, e2 = (a[0] + d[0] * t) - X, (a[1] + d[1] * t) - Y, (a[2] + d[2] * t) - Z
        inside = ((e0 * e0 + e1 * e1) + e2 * e2) <= r * r
        nbx, nby, nbz = len(ix) // 4, len(iy) // 4, len(iz) // 4
        cells = inside.reshape(nbx, 4, nby, 4, nbz, 4).transpose(0, 2, 4, 1, 3, 5).reshape(nbx, nby, nbz, 64)
        w = (np.uint64(1) << np.arange(64, dtype=np.uint64))
        masks = (cells.astype(np.uint64) * w).sum(axis=-1, dtype=np.uint64)
        self.blocks[lo[0]:hi[0] + 1, lo[1]:hi[1] + 1, lo[2]:hi[2] + 1] |= masks

    def add_voxels(self, other):
        if other._N != self._N:
            raise L.InvalidArgument("voxel dimension mismatch (%d != %d)" % (self._N, other._N))
        self.blocks |= other.blocks

    # set operations on whole blocks / voxel sets (VoxelOctree.cpp:224-249, 980-996); the block forms return the OLD value
    def union_block(self, bx, by, bz, value):
        old = int(self.blocks[bx, by, bz]); self.blocks[bx, by, bz] = np.uint64(old | int(value)); return old

    def intersect_block(self, bx, by, bz, value):
        old = int(self.blocks[bx, by, bz]); self.blocks[bx, by, bz] = np.uint64(old & int(value)); return old

    def subtract_block(self, bx, by, bz, value):
        return self.intersect_block(bx, by, bz, ~int(value) & 0xFFFFFFFFFFFFFFFF)

    def remove_point(self, p):
        x, y, z = map(float, p)
        if self.is_in_domain(x, y, z):
            self.set_cell(*self.nearest_cell(x, y, z), False)

    def _same_size(self, other):
        if other._N != self._N:
            raise L.InvalidArgument("voxel dimension mismatch (%d != %d)" % (self._N, other._N))

    def remove_voxels(self, other):
        self._same_size(other)
        self.blocks &= ~other.blocks

    def intersect_voxels(self, other):
        self._same_size(other)
        self.blocks &= other.blocks

    def visit_leaves(self, visitor):
        """visitor(bx, by, bz, block) for every non-empty block in the reference's traversal order
        (VoxelOctree.cpp:998-1001 -> detail/TreeNode.hxx:176-190; pinned by tests/golden/treenode_*.npz)."""
        ids, masks = self.to_sparse(leaf_order=True)
        Nb = self.Nbx()
        for i, m in zip(ids.tolist(), masks.tolist()):
            visitor(i // (Nb * Nb), (i // Nb) % Nb, i % Nb, int(m))

    def occupied_voxels(self):
        """(n, 3) cell indices of the occupied voxels (visit_occupied_voxels, :1013-1017)."""
        ids = np.argwhere(self.blocks != 0)
        out = []
        for bx, by, bz in ids:
            v = int(self.blocks[bx, by, bz])
            bits = np.flatnonzero([(v >> k) & 1 for k in range(64)])
            out.append(np.stack([4 * bx + bits // 16, 4 * by + (bits // 4) % 4, 4 * bz + bits % 4], 1))
        return np.concatenate(out) if out else np.zeros((0, 3), dtype=np.int64)

    def collides(self, other):
        """Host form of VoxelOctree::collides (VoxelOctree.cpp:967-978) for small checks in tests."""
        if isinstance(other, VoxelOctree):
            if other._N != self._N:
                raise L.InvalidArgument("voxel dimension mismatch (%d != %d)" % (self._N, other._N))
            return bool((self.blocks & other.blocks).any())
        x, y, z = map(float, other)
        return self.is_in_domain(x, y, z) and self.cell(*self.nearest_cell(x, y, z))

    # sparse export: what a roadmap voxel cache stores (VoxelCachedLazyPRM.cpp:986-1114 .rmp blocks) ---
    def to_sparse(self, leaf_order=False):
        """(dense block ids, masks) of the non-empty blocks, ascending by id -- or, with leaf_order, in the order the
        reference visits (and serialises) them."""
        ids = np.flatnonzero(self.blocks.reshape(-1)).astype(np.uint32)
        if leaf_order:
            ids = ids[leaf_order_of(ids, self.Nbx())]
        return ids, self.blocks.reshape(-1)[ids].copy()

    # ---- the reference's obstacle-set file formats (collision/VoxelOctree.cpp:1357-1497): what a maintainer hands to
    # tr_set_grid comes from these files.  .nrrd needs ITK and is not read here. ------------------------------------------
    def to_json(self):
        """VoxelOctree::to_json (:1357-1376): {"VoxelOctree": {dimension, x/y/zlimits, data: [[bx, by, bz, block], ...]}}."""
        Nb = self.Nbx()
        ids, masks = self.to_sparse(leaf_order=True)
        data = [[int(i) // (Nb * Nb), (int(i) // Nb) % Nb, int(i) % Nb, int(m)] for i, m in zip(ids, masks)]
        return {"VoxelOctree": {"dimension": int(self._N), "xlimits": [self._xmin, self._xmax], "ylimits": [self._ymin, self._ymax],
                                "zlimits": [self._zmin, self._zmax], "data": data}}

    @classmethod
    def from_json(cls, obj):
        """VoxelOctree::from_json (:1378-1394)."""
        o = obj["VoxelOctree"]
        v = cls(int(o["dimension"]))
        v.set_xlim(*o["xlimits"]); v.set_ylim(*o["ylimits"]); v.set_zlim(*o["zlimits"])
        for bx, by, bz, val in o["data"]:
            v.set_block(int(bx), int(by), int(bz), int(val))
        return v

    def to_toml(self):
        """VoxelOctree::to_toml (:1396-1430) as TOML text: blocks as [bx, by, bz, upper 32 bits, lower 32 bits]."""
        d = self.to_json()["VoxelOctree"]
        rows = ",\n  ".join("[%d, %d, %d, %d, %d]" % (bx, by, bz, val >> 32, val & 0xffffffff) for bx, by, bz, val in d["data"])
        lim = lambda a: "[%r, %r]" % (float(a[0]), float(a[1]))
        return ("[VoxelOctree]\ndimension = %d\nxlimits = %s\nylimits = %s\nzlimits = %s\ndata = [\n  %s\n]\n"
                % (d["dimension"], lim(d["xlimits"]), lim(d["ylimits"]), lim(d["zlimits"]), rows))

    @classmethod
    def from_toml(cls, tbl):
        """VoxelOctree::from_toml (:1432-1473): `tbl` is the parsed table, with or without the "VoxelOctree" container."""
        o = tbl.get("VoxelOctree", tbl)
        v = cls(int(o["dimension"]))
        v.set_xlim(*o["xlimits"]); v.set_ylim(*o["ylimits"]); v.set_zlim(*o["zlimits"])
        for bx, by, bz, hi, lo in o["data"]:
            v.set_block(int(bx), int(by), int(bz), (int(hi) << 32) | int(lo))
        return v

    def to_file(self, fname):
        """VoxelOctree::to_file (:1475-1485): .toml[.gz], else the JSON family by extension (util/json_io.cpp: .json,
        .msgpack, each optionally .gz)."""
        import gzip
        import json
        gz = fname.endswith(".gz")
        base = fname[:-3] if gz else fname
        if base.endswith(".nrrd"):
            raise L.Unsupported(".nrrd needs ITK: not available in this build")
        if base.endswith(".toml"):
            raw = self.to_toml().encode()
        elif base.endswith(".msgpack"):
            import msgpack
            raw = msgpack.packb(self.to_json())
        else:
            raw = json.dumps(self.to_json()).encode()
        with (gzip.open(fname, "wb") if gz else open(fname, "wb")) as f:
            f.write(raw)

    @classmethod
    def from_file(cls, fname):
        """VoxelOctree::from_file (:1487-1497)."""
        import gzip
        import json
        gz = fname.endswith(".gz")
        base = fname[:-3] if gz else fname
        if base.endswith(".nrrd"):
            raise L.Unsupported(".nrrd needs ITK: not available in this build")
        with (gzip.open(fname, "rb") if gz else open(fname, "rb")) as f:
            raw = f.read()
        if base.endswith(".toml"):
            import tomli
            return cls.from_toml(tomli.loads(raw.decode()))
        if base.endswith(".msgpack"):
            import msgpack
            return cls.from_json(msgpack.unpackb(raw))
        return cls.from_json(json.loads(raw.decode()))

    @classmethod
    def from_sparse(cls, N, limits, ids, masks):
        v = cls(N)
        v.set_xlim(limits[0], limits[1]); v.set_ylim(limits[2], limits[3]); v.set_zlim(limits[4], limits[5])
        v.blocks.reshape(-1)[np.asarray(ids, dtype=np.int64)] = np.asarray(masks, dtype=np.uint64)
        return v
