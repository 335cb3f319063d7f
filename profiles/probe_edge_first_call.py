"""The first full-size tr_validate_edges_indexed of a context (what bench_roadmap.py's config 3 line times) against the calls after it:
TENDON_HIP_EDGE_TIMING=1 laps of each."""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TENDON_HIP_EDGE_TIMING"] = "1"
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
chk.engine.reserve(1 << 20)
states, _ = rb.sample_valid_vertices(100000, batch=1 << 17)
edges = rb.knn_edges_gpu(states, 11)
chk.engine.reserve_edges(len(edges))
import ctypes as C
L = irt._lib
eng = chk.engine
_real = eng.lib.tr_validate_edges_indexed
def _timed(*a):
    t0 = time.perf_counter(); r = _real(*a); print("   C call %.3f ms" % (1e3 * (time.perf_counter() - t0)), file=sys.stderr, flush=True); return r
class _Lib:
    def __getattr__(self, k):
        return _timed if k == "tr_validate_edges_indexed" else getattr(_lib0, k)
_lib0 = eng.lib
eng.lib = _Lib()
for i in range(4):
    print("--- call %d" % i, file=sys.stderr, flush=True)
    t0 = time.perf_counter()
    if i == 0 and "cprofile" in sys.argv[1:]:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        v, nf = rb.validate_edges(states, edges)
        pr.disable(); dt = time.perf_counter() - t0
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(14)
    else:
        v, nf = rb.validate_edges(states, edges)
        dt = time.perf_counter() - t0
    print("python wall %.3f ms, %d edges, %.3g edges/s" % (1e3 * dt, len(edges), len(edges) / dt), file=sys.stderr, flush=True)
