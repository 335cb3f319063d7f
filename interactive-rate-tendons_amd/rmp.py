"""Reader / writer for the reference's binary roadmap format `.rmp`
(motion-planning/VoxelCachedLazyPRM.cpp: RmpStreamer :986-1114, LazyRmpParser :863-967, the
binary_write / binary_read helpers :583-659, serialize_inner :557-578) that goes straight between the
file and the CSR voxel caches the engine consumes (tr_check_cached), without building octrees
(SURVEY.md 8f rank 2).  Little-endian, packed:

    u32 n_verts, u32 n_edges, u8 has_voxels, [u8 Nb, f64 x 6 limits (xmin,xmax,ymin,ymax,zmin,zmax)]
    vertex: u32 index, u32 n + f64 x n state, u8 has_tip [+ f64 x 3], [u8 has_voxels_here [+ u32 nblocks + nblocks x (u8 bx, u8 by, u8 bz, u64 mask)]]
    edge:   u32 source, u32 target, f64 weight, [voxels as above]

The per-item voxel fields exist only when the header's has_voxels is set.
"""
import struct

import numpy as np

from .collision import leaf_order_of

_BLOCK = np.dtype([("bx", "u1"), ("by", "u1"), ("bz", "u1"), ("mask", "<u8")])     # 11 bytes, packed


def _pack_voxels(out, caches, i, Nb):
    if caches is None:
        return
    a, b = int(caches["offsets"][i]), int(caches["offsets"][i + 1])
    present = caches.get("present")
    has = bool(present[i]) if present is not None else True
    out.append(struct.pack("<B", int(has)))
    if not has:
        return
    ids = np.asarray(caches["block_ids"][a:b], dtype=np.int64)
    order = leaf_order_of(ids, Nb)          # serialize_inner writes visit_leaves order (:633-642), not ascending ids
    ids = ids[order]
    rec = np.empty(b - a, dtype=_BLOCK)
    rec["bx"], rec["by"], rec["bz"] = ids // (Nb * Nb), (ids // Nb) % Nb, ids % Nb
    rec["mask"] = np.asarray(caches["masks"][a:b])[order]
    out.append(struct.pack("<I", b - a))
    out.append(rec.tobytes())


def write_rmp(path, states, tips=None, edges=None, weights=None, vertex_caches=None, edge_caches=None,
              N=None, limits=None, indices=None):
    """states (n, S); tips (n, 3) or None; edges (m, 2) int; weights (m,); caches: dict(offsets, block_ids,
    masks[, present]) in the engine's CSR form; N = voxels per axis and limits when caches are given."""
    states = np.ascontiguousarray(states, dtype="<f8")
    n = len(states)
    edges = np.zeros((0, 2), dtype=np.int64) if edges is None else np.asarray(edges)
    m = len(edges)
    weights = np.zeros(m) if weights is None else np.asarray(weights, dtype=np.float64)
    has_vox = vertex_caches is not None or edge_caches is not None
    out = [struct.pack("<IIB", n, m, int(has_vox))]
    Nb = 0
    if has_vox:
        if N is None or limits is None:
            raise ValueError("voxel caches need the reference grid size and limits")
        Nb = N // 4
        if Nb > 255:
            raise ValueError("the .rmp header stores the block count per axis in one byte")
        out.append(struct.pack("<B6d", Nb, *map(float, limits)))
        empty = dict(offsets=np.zeros(max(n, m) + 1, dtype=np.int64), block_ids=np.zeros(0, np.uint32),
                     masks=np.zeros(0, np.uint64), present=np.zeros(max(n, m), bool))
        vertex_caches = vertex_caches or empty
        edge_caches = edge_caches or empty
    for i in range(n):
        idx = i if indices is None else int(indices[i])
        out.append(struct.pack("<II", idx, states.shape[1]))
        out.append(states[i].tobytes())
        if tips is None:
            out.append(b"\x00")
        else:
            out.append(b"\x01" + np.ascontiguousarray(tips[i], dtype="<f8").tobytes())
        _pack_voxels(out, vertex_caches, i, Nb)
    for j in range(m):
        out.append(struct.pack("<IId", int(edges[j, 0]), int(edges[j, 1]), float(weights[j])))
        _pack_voxels(out, edge_caches, j, Nb)
    with open(path, "wb") as f:
        f.write(b"".join(out))


def _read_voxels(buf, pos, Nb, ids, masks, offsets, present):
    has = buf[pos]
    pos += 1
    present.append(bool(has))
    if has:
        (nb,) = struct.unpack_from("<I", buf, pos)
        pos += 4
        rec = np.frombuffer(buf, dtype=_BLOCK, count=nb, offset=pos)
        pos += nb * _BLOCK.itemsize
        ids.append((rec["bx"].astype(np.uint32) * Nb + rec["by"]) * Nb + rec["bz"])
        masks.append(rec["mask"].astype(np.uint64))
        offsets.append(offsets[-1] + nb)
    else:
        offsets.append(offsets[-1])
    return pos


def read_rmp(path):
    """-> dict(states, indices, tips (NaN where absent), edges, weights, N, limits, vertex_caches, edge_caches)."""
    buf = memoryview(open(path, "rb").read())
    n, m, has_vox = struct.unpack_from("<IIB", buf, 0)
    pos = 9
    Nb, limits = 0, None
    if has_vox:
        Nb = buf[pos]
        limits = struct.unpack_from("<6d", buf, pos + 1)
        pos += 1 + 48
    states, indices, tips = [], np.empty(n, dtype=np.uint32), np.full((n, 3), np.nan)
    v = dict(ids=[], masks=[], offsets=[0], present=[])
    e = dict(ids=[], masks=[], offsets=[0], present=[])
    for i in range(n):
        idx, cnt = struct.unpack_from("<II", buf, pos)
        pos += 8
        indices[i] = idx
        states.append(np.frombuffer(buf, dtype="<f8", count=cnt, offset=pos))
        pos += 8 * cnt
        has_tip = buf[pos]
        pos += 1
        if has_tip:
            tips[i] = np.frombuffer(buf, dtype="<f8", count=3, offset=pos)
            pos += 24
        if has_vox:
            pos = _read_voxels(buf, pos, Nb, v["ids"], v["masks"], v["offsets"], v["present"])
    edges, weights = np.empty((m, 2), dtype=np.int64), np.empty(m)
    for j in range(m):
        s, t, w = struct.unpack_from("<IId", buf, pos)
        pos += 16
        edges[j], weights[j] = (s, t), w
        if has_vox:
            pos = _read_voxels(buf, pos, Nb, e["ids"], e["masks"], e["offsets"], e["present"])
    if pos != len(buf):
        raise ValueError("trailing bytes in %s" % path)

    def csr(d):
        return dict(offsets=np.array(d["offsets"], dtype=np.int64),
                    block_ids=np.concatenate(d["ids"]) if d["ids"] else np.zeros(0, np.uint32),
                    masks=np.concatenate(d["masks"]) if d["masks"] else np.zeros(0, np.uint64),
                    present=np.array(d["present"], dtype=bool))
    return dict(states=np.array(states).reshape(n, -1) if n else np.zeros((0, 0)), indices=indices, tips=tips,
                edges=edges, weights=weights, N=4 * Nb if has_vox else None, limits=limits,
                vertex_caches=csr(v) if has_vox else None, edge_caches=csr(e) if has_vox else None)
