"""ctypes front end of oracle/_ref/libref_treenode.so -- the REFERENCE's own `collision::detail::TreeNode<N>`
(/root/reference/cpp/src/collision/detail/TreeNode.h, .hxx) compiled as it lies, behind oracle/ref_treenode_driver.cpp.
TEST INFRASTRUCTURE ONLY (tests/ and tests/golden/make_treenode_golden.py); the product never loads it.

`build()` compiles it where /root/reference exists (the build container); on the GPU box only the prebuilt file is
used, and `available()` says whether there is one.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libref_treenode.so")
REFERENCE = os.environ.get("TENDON_REFERENCE_ROOT", "/root/reference")
_lib = None


def reference_present():
    return os.path.exists(os.path.join(REFERENCE, "cpp", "src", "collision", "detail", "TreeNode.h"))


def build():
    """make -C oracle ref; returns the library path, or None when the reference's sources are not on this machine."""
    if not reference_present():
        return LIB_PATH if os.path.exists(LIB_PATH) else None
    subprocess.check_call(["make", "-C", _HERE, "-s", "ref", "REFERENCE=" + REFERENCE])
    return LIB_PATH


def available():
    return os.path.exists(LIB_PATH)


def _load():
    global _lib
    if _lib is not None:
        return _lib
    if not available():
        raise RuntimeError("oracle/_ref/libref_treenode.so is not built (make -C oracle ref needs /root/reference)")
    lib = C.CDLL(LIB_PATH)
    u32, u64, vp = C.c_uint32, C.c_uint64, C.c_void_p
    lib.ref_tree_new.argtypes, lib.ref_tree_new.restype = [C.c_int], vp
    lib.ref_tree_free.argtypes, lib.ref_tree_free.restype = [vp], None
    lib.ref_tree_copy.argtypes, lib.ref_tree_copy.restype = [vp], vp
    lib.ref_tree_N.argtypes = [vp]
    lib.ref_tree_block.argtypes, lib.ref_tree_block.restype = [vp, u32, u32, u32], u64
    lib.ref_tree_set_block.argtypes, lib.ref_tree_set_block.restype = [vp, u32, u32, u32, u64], None
    lib.ref_tree_union_block.argtypes, lib.ref_tree_union_block.restype = [vp, u32, u32, u32, u64], u64
    lib.ref_tree_intersect_block.argtypes, lib.ref_tree_intersect_block.restype = [vp, u32, u32, u32, u64], u64
    lib.ref_tree_nblocks.argtypes, lib.ref_tree_nblocks.restype = [vp], u64
    lib.ref_tree_is_empty.argtypes = [vp]
    for name in ("collides", "equals"):
        getattr(lib, "ref_tree_" + name).argtypes = [vp, vp]
    for name in ("union_tree", "intersect_tree", "remove_tree"):
        f = getattr(lib, "ref_tree_" + name)
        f.argtypes, f.restype = [vp, vp], None
    lib.ref_tree_leaves.argtypes, lib.ref_tree_leaves.restype = [vp, C.POINTER(u32), C.POINTER(u64), C.c_long], C.c_long
    lib.ref_tree_blocks_visited.argtypes, lib.ref_tree_blocks_visited.restype = [vp], C.c_long
    lib.ref_tree_union_blocks.argtypes = [vp, C.POINTER(u32), C.POINTER(u64), C.c_long, C.POINTER(u64)]
    lib.ref_tree_union_blocks.restype = None
    lib.ref_tree_set_blocks.argtypes, lib.ref_tree_set_blocks.restype = [vp, C.POINTER(u32), C.POINTER(u64), C.c_long], None
    _lib = lib
    return lib


def _u32(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class RefTree:
    """One `TreeNode<N>` of the reference (N voxels per axis, N/4 blocks per axis)."""

    def __init__(self, N, _ptr=None):
        self.lib = _load()
        self.N = int(N)
        self.Nb = self.N // 4
        self.ptr = _ptr if _ptr is not None else self.lib.ref_tree_new(self.N)
        if not self.ptr:
            raise ValueError("TreeNode<%r> is not one of VoxelOctree's variants" % N)

    def __del__(self):
        if getattr(self, "ptr", None):
            self.lib.ref_tree_free(self.ptr)
            self.ptr = None

    def copy(self):
        return RefTree(self.N, _ptr=self.lib.ref_tree_copy(self.ptr))

    def _same(self, other):
        if other.N != self.N:
            raise ValueError("voxel dimension mismatch")

    def block(self, bx, by, bz):
        return int(self.lib.ref_tree_block(self.ptr, bx, by, bz))

    def set_block(self, bx, by, bz, v):
        self.lib.ref_tree_set_block(self.ptr, bx, by, bz, int(v))

    def union_block(self, bx, by, bz, v):
        return int(self.lib.ref_tree_union_block(self.ptr, bx, by, bz, int(v)))

    def intersect_block(self, bx, by, bz, v):
        return int(self.lib.ref_tree_intersect_block(self.ptr, bx, by, bz, int(v)))

    def nblocks(self):
        return int(self.lib.ref_tree_nblocks(self.ptr))

    def is_empty(self):
        return bool(self.lib.ref_tree_is_empty(self.ptr))

    def collides(self, other):
        self._same(other)
        return bool(self.lib.ref_tree_collides(self.ptr, other.ptr))

    def __eq__(self, other):
        self._same(other)
        return bool(self.lib.ref_tree_equals(self.ptr, other.ptr))

    def union_tree(self, other):
        self._same(other)
        self.lib.ref_tree_union_tree(self.ptr, other.ptr)

    def intersect_tree(self, other):
        self._same(other)
        self.lib.ref_tree_intersect_tree(self.ptr, other.ptr)

    def remove_tree(self, other):
        self._same(other)
        self.lib.ref_tree_remove_tree(self.ptr, other.ptr)

    def leaves(self):
        """(bxyz (n, 3) uint32, vals (n,) uint64) in the reference's visit_leaves order."""
        n = self.lib.ref_tree_leaves(self.ptr, None, None, 0)
        bxyz, vals = np.empty((n, 3), dtype=np.uint32), np.empty(n, dtype=np.uint64)
        self.lib.ref_tree_leaves(self.ptr, bxyz.ctypes.data_as(C.POINTER(C.c_uint32)),
                                 vals.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        return bxyz, vals

    def blocks_visited(self):
        return int(self.lib.ref_tree_blocks_visited(self.ptr))

    def union_blocks(self, bxyz, vals):
        """union_block over rows in order; returns the previous values."""
        bxyz, vals = _u32(bxyz).reshape(-1, 3), _u64(vals)
        assert len(bxyz) == len(vals) and (bxyz < self.Nb).all()
        prev = np.empty(len(vals), dtype=np.uint64)
        self.lib.ref_tree_union_blocks(self.ptr, bxyz.ctypes.data_as(C.POINTER(C.c_uint32)),
                                       vals.ctypes.data_as(C.POINTER(C.c_uint64)), len(vals),
                                       prev.ctypes.data_as(C.POINTER(C.c_uint64)))
        return prev

    def set_blocks(self, bxyz, vals):
        bxyz, vals = _u32(bxyz).reshape(-1, 3), _u64(vals)
        assert len(bxyz) == len(vals) and (bxyz < self.Nb).all()
        self.lib.ref_tree_set_blocks(self.ptr, bxyz.ctypes.data_as(C.POINTER(C.c_uint32)),
                                     vals.ctypes.data_as(C.POINTER(C.c_uint64)), len(vals))
