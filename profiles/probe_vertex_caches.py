"""tr_voxelize_batch at config 3's size: the C call, the fetch, the numpy tail -- against 3 ms of kernels."""
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
eng = chk.engine
eng.voxelize_batch(states, device=True)
st = np.ascontiguousarray(states); n = len(st)
for rep in range(4):
    t0 = time.perf_counter()
    offsets = np.zeros(n + 1, dtype=np.int64); bits = np.zeros((n + 63) // 64, dtype=np.uint64); tips = np.empty((n, 3))
    t1 = time.perf_counter()
    eng.profile_begin()
    L.check(eng._ctx, eng.lib.tr_voxelize_batch(eng._ctx, st.ctypes.data_as(C.POINTER(C.c_double)), n, offsets.ctypes.data_as(C.POINTER(C.c_int64)),
                                                bits.ctypes.data_as(C.POINTER(C.c_uint64)), tips.ctypes.data_as(C.POINTER(C.c_double))))
    t2 = time.perf_counter()
    p = eng.profile_read(); eng.profile_end()
    ids, masks = eng._fetch_lists(int(offsets[-1]), rep % 2 == 0)
    t3 = time.perf_counter()
    print("alloc %.2f  C call %.2f  fetch(%s) %.2f  total %.2f ms   kernels %s" % (1e3 * (t1 - t0), 1e3 * (t2 - t1), "dev" if rep % 2 == 0 else "host", 1e3 * (t3 - t2), 1e3 * (t3 - t0),
          {k: round(v["total_ms"], 2) for k, v in p.items() if v["launches"]}), flush=True)
