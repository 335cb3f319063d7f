import importlib
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def irt():
    return importlib.import_module("interactive-rate-tendons_amd")


@pytest.fixture(scope="session")
def orc():
    from oracle import oracle
    oracle.build()
    return oracle


def make_oracle_robot(orc, robot, **kw):
    """oracle Robot with the same constants as a package TendonRobot."""
    s = robot.specs
    return orc.Robot([t.C for t in robot.tendons], [t.D for t in robot.tendons], r=robot.r, L=s.L, dL=s.dL,
                     ro=s.ro, ri=s.ri, E=s.E, nu=s.nu,
                     max_tension=[t.max_tension for t in robot.tendons],
                     min_length=[t.min_length for t in robot.tendons],
                     max_length=[t.max_length for t in robot.tendons],
                     enable_rotation=robot.enable_rotation, enable_retraction=robot.enable_retraction,
                     residual_threshold=robot.residual_threshold, **kw)


def make_oracle_grid(orc, vox):
    g = orc.Grid(vox.Nx(), vox.limits())
    g.blocks()[...] = vox.blocks
    return g


@pytest.fixture(scope="session")
def helpers():
    class H:
        oracle_robot = staticmethod(make_oracle_robot)
        oracle_grid = staticmethod(make_oracle_grid)
    return H
