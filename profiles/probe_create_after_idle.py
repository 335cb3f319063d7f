"""create_roadmap's phases in a process that has done other work first / has idled: is the vertex phase's time stable?"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
irt = importlib.import_module("interactive-rate-tendons_amd")
W = irt.workloads
robot = W.robot_config3()
vox, _ = W.reach_environment(seed=7, n_spheres=64)
chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
rb = irt.RoadmapBuilder(chk, irt.VoxelBackboneMotionValidator(chk), seed=11)


def show(tag):
    t0 = time.perf_counter(); rb.create_roadmap(100000, k=10, device=True); dt = time.perf_counter() - t0
    print("%-28s create_roadmap %.2f ms:" % (tag, 1e3 * dt), {k: round(1e3 * v["seconds"], 2) for k, v in rb.timing.items() if "seconds" in v}, flush=True)


for i in range(3):
    show("call %d" % i)
time.sleep(6.0)
show("after 6 s of idling")
show("right after")
t0 = time.perf_counter()
x = 0.0
while time.perf_counter() - t0 < 6.0:                      # a busy host thread, idle GPU (what bench_roadmap's CPU comparison does)
    x += float(np.sum(np.random.default_rng(0).random(100000)))
show("after 6 s of host work")
show("right after")
chk.engine.set_grid(vox.Nx(), vox.limits(), vox.blocks)
show("after set_grid")
