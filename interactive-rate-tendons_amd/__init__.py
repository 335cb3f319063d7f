"""interactive-rate-tendons_amd -- MI355X-native batched tendon-robot forward kinematics +
voxel-collision engine behind the reference's TendonRobot / VoxelOctree / validity-checker API.

The directory name is not a Python identifier; import it with
    irt = importlib.import_module("interactive-rate-tendons_amd")
(tests/conftest.py and bench.py do exactly that).

All compute runs in libtendon_hip.so (HIP, gfx950) through the C ABI of include/tendon_hip.h.
There is no CPU fallback: without the built library or without a GPU the compute calls raise.
"""
from . import _lib
from ._lib import (TendonHipError, InvalidArgument, OutOfRange, DomainError, LengthError, HipError, Unsupported,
                   build, LIB_PATH)
from .engine import Engine, unpack_bits
from .tendon import BackboneSpecs, TendonSpecs, TendonResult, TendonRobot
from .collision import VoxelOctree
from .motion_planning import (VoxelEnvironment, VoxelBackboneValidityChecker, VoxelValidityChecker, VoxelBackboneMotionValidator,
                              VoxelBackboneDiscreteMotionValidator, FunctionTimer, Environment, Problem)
from . import workloads, distributed, roadmap, rmp, tip_control
from .roadmap import RoadmapBuilder, VoxelCachedLazyPRM

__all__ = [
    "TendonHipError", "InvalidArgument", "OutOfRange", "DomainError", "LengthError", "HipError", "Unsupported",
    "build", "LIB_PATH", "Engine", "unpack_bits", "BackboneSpecs", "TendonSpecs", "TendonResult", "TendonRobot",
    "VoxelOctree", "VoxelEnvironment", "VoxelBackboneValidityChecker", "VoxelValidityChecker", "VoxelBackboneMotionValidator", "VoxelBackboneDiscreteMotionValidator", "Environment", "Problem",
    "FunctionTimer", "workloads", "distributed", "roadmap", "RoadmapBuilder", "VoxelCachedLazyPRM", "tip_control",
]
