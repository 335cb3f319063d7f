import importlib, os, sys, time
os.environ["TENDON_HIP_VOX_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)
for i in range(4):
    sys.stderr.write("--- call %d\n" % i); sys.stderr.flush()
    t0 = time.perf_counter(); rb.create_roadmap(100000, k=10, device=True); dt = time.perf_counter() - t0
    sys.stderr.write("create_roadmap %.2f ms: %s\n" % (1e3 * dt, {k: round(1e3 * v["seconds"], 2) for k, v in rb.timing.items() if "seconds" in v}))
