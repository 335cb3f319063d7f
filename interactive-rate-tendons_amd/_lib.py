"""ctypes binding of libtendon_hip.so (the C ABI declared in include/tendon_hip.h).

There is no CPU fallback: if the shared library is missing or no HIP device is present the
functions that need it raise, loudly.  `build()` compiles the library in-tree with hipcc for
gfx950 (cross-compiles without a GPU).
"""
import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)
LIB_PATH = os.environ.get("TENDON_HIP_LIB") or os.path.join(_PKG, "libtendon_hip.so")   # env override: A/B builds
SRC_DIR = os.path.join(_PKG, "csrc")
HEADER = os.path.join(_ROOT, "include", "tendon_hip.h")

TR_OK, TR_ERR_INVALID_ARG, TR_ERR_OUT_OF_RANGE, TR_ERR_DOMAIN, TR_ERR_LENGTH, TR_ERR_RUNTIME, \
    TR_ERR_HIP, TR_ERR_UNSUPPORTED = range(8)

TR_FLAG_CONVERGED, TR_FLAG_LENGTH_OK, TR_FLAG_NO_SELFCOL, TR_FLAG_NO_VOXCOL, TR_FLAG_DOMAIN = 1, 2, 4, 8, 16
TR_CHECKER_BACKBONE, TR_CHECKER_SPHERES = 0, 1
TR_PROFILE_SLOTS = 6
PROFILE_SLOT_NAMES = ("fk_rk4_batch", "backbone_voxel_sweep", "cached_blocks_vs_grid", "edge_helpers", "fk_sweep_fused", "fk_verdict")

# every symbol include/tendon_hip.h declares (tests check the .so exports exactly these)
ABI_SYMBOLS = (
    "tr_create", "tr_destroy", "tr_last_error", "tr_state_size", "tr_num_points", "tr_device",
    "tr_home_lengths", "tr_set_grid", "tr_set_checker", "tr_grid_add_spheres", "tr_grid_add_capsules", "tr_grid_remove_interior", "tr_grid_dilate",
    "tr_grid_dilate_sphere", "tr_get_grid", "tr_reserve", "tr_reserve_edges", "tr_fk_batch", "tr_fk_batch_dev", "tr_fk_batch_retraction_dev",
    "tr_validate_shapes_retraction_dev",
    "tr_validate_batch", "tr_validate_batch_dev", "tr_validate_shapes_dev", "tr_validate_edges", "tr_validate_edges_indexed", "tr_validate_edges_last_valid",
    "tr_validate_edges_discrete",
    "tr_check_cached", "tr_check_cached_dev", "tr_check_cached_subset_dev", "tr_state_layout", "tr_space_weights", "tr_kstar_k",
    "tr_roadmap_create", "tr_roadmap_destroy", "tr_roadmap_last_error", "tr_roadmap_set_caches", "tr_roadmap_set_caches_dev", "tr_roadmap_prepare", "tr_roadmap_clear_validity",
    "tr_roadmap_revalidate", "tr_roadmap_get_validity", "tr_roadmap_set_validity", "tr_roadmap_solve", "tr_roadmap_fetch_paths", "tr_roadmap_search_stats", "tr_roadmap_profile", "tr_roadmap_release_search_state", "tr_roadmap_reserve_search_state", "tr_roadmap_search_state_bytes", "tr_roadmap_search_sweeps", "tr_voxelize_batch", "tr_voxelize_edges", "tr_voxelize_edges_indexed", "tr_connect_edges_indexed", "tr_voxelize_fetch", "tr_voxelize_fetch_dev", "tr_voxelize_count", "tr_knn", "tr_knn_range", "tr_knn_table_edges", "tr_knn_edges", "tr_knn_edges_dev", "tr_knn_range_dev", "tr_knn_table_edges_dev", "tr_validate_edges_indexed_dev", "tr_signature_words", "tr_signature_packed_words", "tr_pack_signatures_dev", "tr_unpack_signatures_dev", "tr_validate_candidates_sig_dev", "tr_validate_edges_indexed_sig_dev", "tr_profile_begin", "tr_profile_read", "tr_profile_end",
    "tr_set_debug", "tr_edge_schedule_last",
    "tr_candidate_states", "tr_candidate_states_dev", "tr_validate_candidates_dev", "tr_compact_rows_dev",
    "tr_sample_valid_vertices", "tr_sample_valid_vertices_dev", "tr_sample_valid_vertices_sig_dev",
)


class TrRobotDesc(C.Structure):
    _fields_ = [
        ("r", C.c_double),
        ("L", C.c_double), ("dL", C.c_double), ("ro", C.c_double), ("ri", C.c_double),
        ("E", C.c_double), ("nu", C.c_double),
        ("n_tendons", C.c_int32), ("n_a", C.c_int32), ("n_m", C.c_int32),
        ("C", C.POINTER(C.c_double)), ("D", C.POINTER(C.c_double)),
        ("max_tension", C.POINTER(C.c_double)), ("min_length", C.POINTER(C.c_double)),
        ("max_length", C.POINTER(C.c_double)),
        ("enable_rotation", C.c_int32), ("enable_retraction", C.c_int32),
        ("residual_threshold", C.c_double),
    ]


class TrSpaceParams(C.Structure):
    _fields_ = [("min_tension_change", C.c_double), ("min_rotation_change", C.c_double),
                ("min_retraction_change", C.c_double)]


class TrRoadmapStats(C.Structure):
    _fields_ = [("rounds", C.c_int64), ("items_checked", C.c_int64), ("astar_runs", C.c_int64), ("expanded", C.c_int64)]


TR_QUERY_SOLVED, TR_QUERY_NO_PATH, TR_QUERY_INVALID_START, TR_QUERY_INVALID_GOAL = range(4)


class TendonHipError(RuntimeError):
    """Base for errors reported by libtendon_hip (status + tr_last_error text)."""
    status = TR_ERR_RUNTIME


# status -> Python exception mirroring the C++ exception type the reference throws
class InvalidArgument(TendonHipError, ValueError):      # std::invalid_argument
    status = TR_ERR_INVALID_ARG


class OutOfRange(TendonHipError, IndexError):           # std::out_of_range
    status = TR_ERR_OUT_OF_RANGE


class DomainError(TendonHipError, ArithmeticError):     # std::domain_error
    status = TR_ERR_DOMAIN


class LengthError(TendonHipError, ValueError):          # std::length_error
    status = TR_ERR_LENGTH


class HipError(TendonHipError):
    status = TR_ERR_HIP


class Unsupported(TendonHipError, NotImplementedError):
    status = TR_ERR_UNSUPPORTED


_EXC = {TR_ERR_INVALID_ARG: InvalidArgument, TR_ERR_OUT_OF_RANGE: OutOfRange, TR_ERR_DOMAIN: DomainError,
        TR_ERR_LENGTH: LengthError, TR_ERR_RUNTIME: TendonHipError, TR_ERR_HIP: HipError,
        TR_ERR_UNSUPPORTED: Unsupported}


HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]
OBJ_DIR = os.path.join(SRC_DIR, "_obj")


def _units():
    """(object name, source, extra flags): the C ABI + K2..K6, the cache merge, and K1 once per
    (tendon count, kernel: shared grid / retraction / fused with K2 / verdict-only / verdict-only with retraction / edge queue) so its
    instantiations compile in parallel."""
    fk_deps = ["fk_inst.hip", "fk_launch.hpp", "fk_kernel.hpp", "fk_retract_kernel.hpp", "fused_kernel.hpp", "verdict_kernel.hpp",
               "sweep_kernel.hpp", "sphere_kernel.hpp", "tr_types.hpp"]
    fk_only = ["fk_inst.hip", "fk_kernel.hpp", "fk_retract_kernel.hpp", "cache_merge.hip", "roadmap.hip", "sample.hip", "edge_queue_kernel.hpp", "search_kernel.hpp"]
    main_deps = [f for f in os.listdir(SRC_DIR) if not f.startswith("_") and f not in fk_only]
    u = [("tendon_hip.o", "tendon_hip.hip", [], main_deps + [HEADER]),
         ("cache_merge.o", "cache_merge.hip", [], ["cache_merge.hip", "cache_merge.hpp"]),
         ("roadmap.o", "roadmap.hip", ["-pthread"], ["roadmap.hip", "search_kernel.hpp", HEADER]),
         ("sample.o", "sample.hip", [], ["sample.hip", "sample.hpp", "tr_types.hpp"])]
    q_deps = fk_deps + ["edge_queue_kernel.hpp", "edge_kernel.hpp"]
    for n in range(1, 9):
        for kind, tag in ((0, "u"), (1, "r"), (2, "f"), (3, "v"), (4, "w"), (5, "q")):
            u.append(("fk_%s%d.o" % (tag, n), "fk_inst.hip", ["-DTRK_INST_N=%d" % n, "-DTRK_INST_KIND=%d" % kind],
                      q_deps if kind == 5 else fk_deps))
    return u


def _sources():
    return [os.path.join(SRC_DIR, f) for f in sorted(os.listdir(SRC_DIR)) if not f.startswith("_")] + [HEADER]


def source_hash():
    """sha1 over the kernel / ABI sources: ties a profile (profiles/traffic_latest.json) to the code it was taken on."""
    import hashlib
    h = hashlib.sha1()
    for f in _sources():
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def build(force=False, verbose=False, jobs=None):
    """Compile libtendon_hip.so in-tree for gfx950 if it is missing or older than its sources.
    Objects go to csrc/_obj/ (git-ignored); a unit is recompiled when any source or header is newer."""
    # up to date = the library was linked from exactly these sources: their hash is recorded next to it at link time
    # (modification times do not survive a checkout or a copy to another machine)
    stamp = LIB_PATH + ".srchash"
    want = source_hash()
    if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == want:
        return LIB_PATH
    os.makedirs(OBJ_DIR, exist_ok=True)
    # one builder at a time: the ranks of a `torch.distributed.run` launch all arrive here when the library is stale, and
    # would compile and link into the same files (bench.py's own launcher builds once in the parent; this covers the rest)
    import fcntl
    with open(os.path.join(OBJ_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(LIB_PATH) and os.path.exists(stamp) and open(stamp).read().strip() == want:
            return LIB_PATH                                      # another process built it while this one waited
        return _build_locked(force, verbose, jobs, stamp, want)


def _build_locked(force, verbose, jobs, stamp, want):
    import hashlib
    todo, stamps = [], []
    for obj, src, extra, deps in _units():
        o = os.path.join(OBJ_DIR, obj)
        # an object is current when it was compiled from exactly these dependencies with exactly these flags: the hash of
        # both is kept next to it (modification times lie after a copy or a checkout)
        h = hashlib.sha1(" ".join(HIPCC_FLAGS + extra + [src]).encode())
        for f in sorted(set(deps)):
            h.update(os.path.basename(f).encode())
            h.update(open(os.path.join(SRC_DIR, f), "rb").read())
        key = h.hexdigest()
        okey = o + ".dephash"
        if force or not os.path.exists(o) or not os.path.exists(okey) or open(okey).read().strip() != key:
            if os.path.exists(okey):
                os.remove(okey)
            todo.append(["hipcc"] + HIPCC_FLAGS + extra + ["-c", os.path.join(SRC_DIR, src), "-o", o])
            stamps.append((okey, key))
    jobs = jobs or max(1, min(len(todo), os.cpu_count() or 1))
    running, failed = [], None
    while (todo or running) and failed is None:
        while todo and len(running) < jobs:
            cmd = todo.pop(0)
            if verbose:
                print(" ".join(cmd), flush=True)
            running.append((cmd, subprocess.Popen(cmd)))
        cmd, p = running.pop(0)
        if p.wait() != 0:
            failed = cmd
    for _, p in running:
        p.wait()
    if failed is not None:
        raise subprocess.CalledProcessError(1, failed)
    for okey, key in stamps:
        with open(okey, "w") as f:
            f.write(key + "\n")
    link = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", LIB_PATH] + \
           [os.path.join(OBJ_DIR, u[0]) for u in _units()]
    if verbose:
        print(" ".join(link), flush=True)
    subprocess.check_call(link)
    with open(stamp, "w") as f:
        f.write(want + "\n")
    return LIB_PATH


_lib = None


def lib():
    """Load libtendon_hip.so (after torch, so both share one HIP runtime). Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libtendon_hip.so is not built (%s). Run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback for this engine." % LIB_PATH)
    try:
        import torch  # noqa: F401  (loads the HIP runtime this process will share)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    P = C.POINTER
    vp, i64, dp = C.c_void_p, C.c_int64, C.POINTER(C.c_double)
    L.tr_create.argtypes = [P(TrRobotDesc), C.c_int, P(vp)]
    L.tr_destroy.argtypes = [vp]
    L.tr_destroy.restype = None
    L.tr_last_error.argtypes = [vp]
    L.tr_last_error.restype = C.c_char_p
    for f in ("tr_state_size", "tr_num_points", "tr_device"):
        getattr(L, f).argtypes = [vp]
    L.tr_home_lengths.argtypes = [vp, dp]
    L.tr_set_grid.argtypes = [vp, C.c_uint32, dp, P(C.c_uint64), dp]
    L.tr_set_checker.argtypes = [vp, C.c_int32]
    L.tr_grid_add_spheres.argtypes = [vp, dp, i64]
    L.tr_grid_add_capsules.argtypes = [vp, dp, i64]
    L.tr_grid_remove_interior.argtypes = [vp, C.c_int32]
    L.tr_grid_dilate.argtypes = [vp, C.c_int32, C.c_int32]
    L.tr_grid_dilate_sphere.argtypes = [vp, C.c_double]
    L.tr_get_grid.argtypes = [vp, P(C.c_uint64)]
    L.tr_reserve.argtypes = [vp, i64]
    L.tr_reserve_edges.argtypes = [vp, i64]
    L.tr_fk_batch.argtypes = [vp, dp, i64, dp, dp, dp, dp, P(C.c_uint8), P(C.c_int32)]
    L.tr_fk_batch_dev.argtypes = [vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.tr_fk_batch_retraction_dev.argtypes = [vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.tr_validate_shapes_retraction_dev.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]
    L.tr_validate_batch.argtypes = [vp, dp, i64, P(C.c_uint64), dp, P(C.c_uint8)]
    L.tr_validate_batch_dev.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.tr_validate_shapes_dev.argtypes = [vp, i64, i64, vp, vp, vp, vp, vp, vp, C.c_int, vp, vp, vp]
    L.tr_validate_edges.argtypes = [vp, P(TrSpaceParams), dp, dp, i64, P(C.c_uint64), P(C.c_int32), P(i64)]
    L.tr_validate_edges_indexed.argtypes = [vp, P(TrSpaceParams), dp, i64, P(C.c_int32), i64, P(C.c_uint64), P(C.c_int32), P(i64)]
    L.tr_validate_edges_last_valid.argtypes = [vp, P(TrSpaceParams), dp, dp, i64, P(C.c_uint64), dp, P(C.c_int32)]
    L.tr_validate_edges_discrete.argtypes = [vp, P(TrSpaceParams), dp, dp, i64, P(C.c_uint64), dp, P(C.c_int32)]
    L.tr_check_cached.argtypes = [vp, P(C.c_uint32), P(C.c_uint64), P(i64), i64, P(C.c_uint64)]
    L.tr_check_cached_dev.argtypes = [vp, vp, vp, vp, i64, vp, vp]
    L.tr_check_cached_subset_dev.argtypes = [vp, vp, vp, vp, i64, vp, i64, vp, vp]
    L.tr_state_layout.argtypes = [vp, P(C.c_int32), P(C.c_int32), P(C.c_int32)]
    L.tr_space_weights.argtypes = [vp, dp, dp]
    L.tr_kstar_k.argtypes = [vp, i64]
    L.tr_roadmap_create.argtypes = [vp, dp, i64, P(C.c_int32), dp, i64, P(vp)]
    L.tr_roadmap_destroy.argtypes = [vp]
    L.tr_roadmap_destroy.restype = None
    L.tr_roadmap_last_error.argtypes = [vp]
    L.tr_roadmap_last_error.restype = C.c_char_p
    L.tr_roadmap_set_caches.argtypes = [vp, P(i64), P(C.c_uint32), P(C.c_uint64), P(C.c_uint64), P(i64), P(C.c_uint32),
                                        P(C.c_uint64), P(C.c_uint64)]
    L.tr_roadmap_set_caches_dev.argtypes = [vp, P(i64), vp, vp, P(C.c_uint64), P(i64), vp, vp, P(C.c_uint64)]
    L.tr_roadmap_clear_validity.argtypes = [vp]
    L.tr_roadmap_prepare.argtypes = [vp, C.c_int32, C.c_int32]
    L.tr_roadmap_revalidate.argtypes = [vp, P(i64), P(i64)]
    L.tr_roadmap_get_validity.argtypes = [vp, P(C.c_uint8), P(C.c_uint8)]
    L.tr_roadmap_set_validity.argtypes = [vp, P(C.c_uint8), P(C.c_uint8)]
    L.tr_roadmap_solve.argtypes = [vp, P(C.c_int32), P(C.c_int32), i64, C.c_int32, P(C.c_int32), dp, P(i64), P(TrRoadmapStats)]
    L.tr_roadmap_fetch_paths.argtypes = [vp, P(C.c_int32), i64]
    L.tr_roadmap_search_stats.argtypes = [vp, P(i64)]
    L.tr_roadmap_profile.argtypes = [vp, dp]
    L.tr_roadmap_release_search_state.argtypes = [vp, P(i64)]
    L.tr_roadmap_search_state_bytes.argtypes = [vp, P(i64)]
    L.tr_roadmap_reserve_search_state.argtypes = [vp, i64]
    L.tr_roadmap_search_sweeps.argtypes = [vp, P(i64)]
    L.tr_voxelize_batch.argtypes = [vp, dp, i64, P(i64), P(C.c_uint64), dp]
    L.tr_voxelize_edges.argtypes = [vp, P(TrSpaceParams), dp, dp, i64, P(i64), P(C.c_uint64), P(C.c_int32)]
    L.tr_voxelize_edges_indexed.argtypes = [vp, P(TrSpaceParams), dp, i64, P(C.c_int32), i64, P(i64), P(C.c_uint64), P(C.c_int32)]
    L.tr_connect_edges_indexed.argtypes = L.tr_voxelize_edges_indexed.argtypes
    L.tr_voxelize_fetch.argtypes = [vp, P(C.c_uint32), P(C.c_uint64), i64]
    L.tr_voxelize_fetch_dev.argtypes = [vp, vp, vp, i64, vp]
    L.tr_voxelize_count.argtypes = [vp]
    L.tr_voxelize_count.restype = i64
    L.tr_knn.argtypes = [vp, dp, i64, C.c_int32, C.c_double, P(C.c_int32), dp]
    L.tr_knn_edges.argtypes = [vp, dp, i64, C.c_int32, C.c_double, P(C.c_int32), i64, P(i64)]
    L.tr_knn_edges_dev.argtypes = [vp, vp, i64, C.c_int32, C.c_double, vp, i64, P(i64)]
    L.tr_validate_edges_indexed_dev.argtypes = [vp, P(TrSpaceParams), vp, i64, vp, i64, vp, vp, P(i64)]
    L.tr_knn_range_dev.argtypes = [vp, vp, i64, i64, i64, C.c_int32, C.c_double, vp]
    L.tr_knn_table_edges_dev.argtypes = [vp, vp, i64, C.c_int32, vp, i64, P(i64)]
    L.tr_signature_words.argtypes = [vp]
    L.tr_signature_packed_words.argtypes = [vp]
    L.tr_pack_signatures_dev.argtypes = [vp, vp, i64, vp, P(i64), vp]
    L.tr_unpack_signatures_dev.argtypes = [vp, vp, i64, vp, vp]
    L.tr_validate_candidates_sig_dev.argtypes = [vp, C.c_uint64, C.c_uint64, i64, dp, dp, vp, vp, vp, vp]
    L.tr_validate_edges_indexed_sig_dev.argtypes = [vp, P(TrSpaceParams), vp, i64, vp, vp, i64, vp, vp, P(i64)]
    L.tr_knn_range.argtypes = [vp, dp, i64, i64, i64, C.c_int32, C.c_double, P(C.c_int32), dp]
    L.tr_knn_table_edges.argtypes = [vp, P(C.c_int32), i64, C.c_int32, P(C.c_int32), i64, P(i64)]
    u64 = C.c_uint64
    L.tr_candidate_states.argtypes = [vp, u64, u64, i64, dp, dp, dp]
    L.tr_candidate_states_dev.argtypes = [vp, u64, u64, i64, dp, dp, vp, vp]
    L.tr_validate_candidates_dev.argtypes = [vp, u64, u64, i64, dp, dp, vp, vp, vp, vp]
    L.tr_compact_rows_dev.argtypes = [vp, vp, i64, vp, C.c_int32, i64, vp, vp, P(i64), vp]
    L.tr_sample_valid_vertices.argtypes = [vp, u64, u64, dp, dp, i64, i64, dp, dp, P(i64), P(i64), P(i64)]
    L.tr_sample_valid_vertices_dev.argtypes = [vp, u64, u64, dp, dp, i64, i64, vp, vp, vp, P(i64), P(i64), vp]
    L.tr_sample_valid_vertices_sig_dev.argtypes = [vp, u64, u64, dp, dp, i64, i64, vp, vp, vp, vp, P(i64), P(i64), vp]
    L.tr_profile_begin.argtypes = [vp]
    L.tr_profile_read.argtypes = [vp, P(i64), dp]
    L.tr_profile_end.argtypes = [vp]
    L.tr_set_debug.argtypes = [vp, C.c_uint32]
    L.tr_edge_schedule_last.argtypes = [vp, P(C.c_uint32)]
    _lib = L
    return L


def check(ctx, status):
    """Raise the Python exception matching a non-zero tr_status."""
    if status == TR_OK:
        return
    msg = lib().tr_last_error(ctx)
    msg = msg.decode() if msg else "status %d" % status
    raise _EXC.get(status, TendonHipError)(msg)
