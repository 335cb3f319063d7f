"""Where the wall time of RoadmapBuilder.connect goes besides its kernels: the C call, the device fetch, the numpy tail."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
irt = importlib.import_module("interactive-rate-tendons_amd")
L = irt._lib
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
states, _ = rb.sample_valid_vertices(100000, batch=1 << 17)
edges = rb.knn_edges_gpu(states, 11)
eng = chk.engine
eng.reserve_edges(len(edges))
rb.connect(states, edges, device=True)
st = np.ascontiguousarray(states); e = np.ascontiguousarray(edges, dtype=np.int32); n = len(e)
for rep in range(3):
    t0 = time.perf_counter()
    sp = L.TrSpaceParams(0.02, 0.01, 0.0001)
    offsets = np.zeros(n + 1, dtype=np.int64); bits = np.zeros((n + 63) // 64, dtype=np.uint64); nfk = np.zeros(n, dtype=np.int32)
    t1 = time.perf_counter()
    L.check(eng._ctx, eng.lib.tr_connect_edges_indexed(eng._ctx, C.byref(sp), st.ctypes.data_as(C.POINTER(C.c_double)), st.shape[0],
                                                       e.ctypes.data_as(C.POINTER(C.c_int32)), n, offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                                       bits.ctypes.data_as(C.POINTER(C.c_uint64)), nfk.ctypes.data_as(C.POINTER(C.c_int32))))
    t2 = time.perf_counter()
    ids, masks = eng._fetch_lists(int(offsets[-1]), True)
    t3 = time.perf_counter()
    ok = irt.unpack_bits(bits, n)
    off2 = np.concatenate([offsets[:-1][ok], offsets[-1:]]); nf2 = nfk[ok]; ee = edges[ok]
    t4 = time.perf_counter()
    print("alloc %.1f  C call %.1f  fetch_dev %.1f  numpy tail %.1f  total %.1f ms" % tuple(1e3 * x for x in (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0)), flush=True)
