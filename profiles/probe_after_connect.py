"""What leaves the device unresponsive for 10 - 35 ms after a connect call?  Times a tiny torch kernel (+ synchronise) right after
(A) the C call alone, (B) the C call + the device fetch of the lists, (C) RoadmapBuilder.connect, (D) the C call followed by a 40 ms sleep."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import ctypes as C
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
L = irt._lib
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
states, _ = rb.sample_valid_vertices(100000)
edges = rb.knn_edges_gpu(states, 11)
eng = chk.engine
eng.reserve_edges(len(edges))
rb.connect(states, edges, device=True)
st = np.ascontiguousarray(states); e = np.ascontiguousarray(edges, dtype=np.int32); n = len(e)
x = torch.zeros(256, device="cuda")


def tiny():
    t0 = time.perf_counter()
    x.add_(1.0)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0)


def c_call():
    sp = L.TrSpaceParams(0.02, 0.01, 0.0001)
    offsets = np.zeros(n + 1, dtype=np.int64); bits = np.zeros((n + 63) // 64, dtype=np.uint64); nfk = np.zeros(n, dtype=np.int32)
    L.check(eng._ctx, eng.lib.tr_connect_edges_indexed(eng._ctx, C.byref(sp), st.ctypes.data_as(C.POINTER(C.c_double)), st.shape[0],
                                                       e.ctypes.data_as(C.POINTER(C.c_int32)), n, offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                                       bits.ctypes.data_as(C.POINTER(C.c_uint64)), nfk.ctypes.data_as(C.POINTER(C.c_int32))))
    return offsets


for rep in range(3):
    tiny(); tiny()
    off = c_call(); a = tiny(); a2 = tiny()
    off = c_call(); ids, masks = eng._fetch_lists(int(off[-1]), True); b = tiny()
    del ids, masks
    r = rb.connect(states, edges, device=True); c_ = tiny()
    del r
    off = c_call(); time.sleep(0.04); d = tiny()
    v = eng.validate_edges_indexed(st, e, 0.02, 0.01, 0.0001); f = tiny()
    print("tiny kernel + synchronise after: C call %.2f ms (again %.2f), C call + fetch %.2f, connect %.2f, C call + 40 ms sleep %.2f, validate_edges_indexed %.2f" % (a, a2, b, c_, d, f), flush=True)
