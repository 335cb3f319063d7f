"""Compile-time guard for the dominant kernel: fk_rk4_batch_uniform<3> must keep two waves per SIMD
without heavy scratch spilling (hipcc's register allocation for this kernel is sensitive to source
structure: an innocent refactor once cost 106 spilled VGPRs and 35 % of throughput; a few tens of spilled
registers, on the other hand, have measured as noise next to the instruction count)."""
import os
import re

import pytest
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "interactive-rate-tendons_amd", "csrc")

TU = r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "fk_kernel.hpp"
template __global__ void trk::fk_rk4_batch_uniform<3, false, false>(const double*, int64_t, int64_t, RobotK,
                                                                     const double*, const StepK*, int, trk::FkOut);
'''


def test_fk_kernel_register_budget(tmp_path):
    src = tmp_path / "k1.hip"
    src.write_text(TU)
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c",
                          "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-I", CSRC, str(src), "-o",
                          str(tmp_path / "k1.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    txt = out.stderr
    get = lambda key: int(re.search(key + r"[^:]*: (\d+)", txt).group(1))
    assert get("Occupancy") == 2
    assert get("ScratchSize") <= 160
    assert get("VGPRs Spill") <= 40
    assert get("AGPRs") == 0


FUSED_TU = r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "fused_kernel.hpp"
template __global__ void trk::fk_sweep_fused<3, false>(const double*, int64_t, int64_t, RobotK, const double*, const StepK*, int,
                                                       trk::FkOut, const trk::FusedSweepArgs*);
'''


def test_fused_kernel_register_budget(tmp_path):
    """The kernel the headline runs: K1's loop must stay as it is inside fk_sweep_fused (two waves per SIMD, a
    handful of spilled registers) -- the sweep behind it may not push the allocation over."""
    src = tmp_path / "kf.hip"
    src.write_text(FUSED_TU)
    out = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-c",
                          "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-I", CSRC, str(src), "-o",
                          str(tmp_path / "kf.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    txt = out.stderr
    get = lambda key: int(re.search(key + r"[^:]*: (\d+)", txt).group(1))
    assert get("Occupancy") == 2
    # (round 3: the signature tile of the edge samples -- SigStage, sweep_kernel.hpp -- cost this kernel a 152-byte frame and 39
    # spilled registers; with the tendon-length quadratures in LDS -- fk_kernel.hpp: li_in_lds -- it is back to 12 B and 2)
    assert get("ScratchSize") <= 32
    assert get("VGPRs Spill") <= 6
    assert get("AGPRs") == 0


VERDICT_TU = r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "verdict_kernel.hpp"
template __global__ void trk::fk_verdict<3, false, %s>(const double*, int64_t, RobotK, const double*, const StepK*, int, double*, const trk::VerdictArgs*);
'''


@pytest.mark.parametrize("spheres", ["false", "true"])
def test_verdict_kernel_register_budget_and_lds_address_space(tmp_path, spheres):
    """fk_verdict, the kernel the headline runs: two waves per SIMD with K1's loop essentially unspilled, and its
    per-point sweep state must be addressed as LDS (ds_*), not through flat pointers (an earlier version kept that state
    behind stored / volatile pointers and hipcc emitted 120 flat accesses with a wait after each)."""
    src = tmp_path / "kv.hip"
    src.write_text(VERDICT_TU % spheres)
    base = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-I", CSRC, str(src)]
    out = subprocess.run(base + ["-c", "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "kv.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    get = lambda key: int(re.search(key + r"[^:]*: (\d+)", out.stderr).group(1))
    assert get("Occupancy") == 2
    assert get("ScratchSize") <= 160
    assert get("VGPRs Spill") <= 40
    assert get("AGPRs") == 0
    asm = subprocess.run(base + ["-S", "-o", "-"], capture_output=True, text=True).stdout
    assert asm.count("ds_read") + asm.count("ds_write") >= 30
    assert len(re.findall(r"\bflat_(load|store)", asm)) <= (24 if spheres == "true" else 16)   # grid / field words and the epilogue's outputs only


VERDICT_RETRACT_TU = r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#define TRK_WITH_RETRACT_VERDICT
#include "fk_retract_kernel.hpp"
#include "verdict_kernel.hpp"
template __global__ void trk::fk_verdict_retract<%d, true, false>(const double*, int64_t, RobotK, const PolyK*, const double*, const StepK*, int, int,
                                                                  const double*, const double*, double*, const trk::VerdictArgs*, trk::RetractHandoff);
template __global__ void trk::fk_retract_prologue<%d, true>(const double*, int64_t, RobotK, const PolyK*, const double*, const StepK*, int, int,
                                                            const double*, const double*, const int32_t*, trk::RetractHandoff);
'''


@pytest.mark.parametrize("n_tendons", [3, 4])
def test_verdict_retract_kernel_occupancy_and_lds_address_space(tmp_path, n_tendons):
    """fk_verdict_retract (retraction robots): two waves per SIMD -- measured faster at every width, the sweep's loads hide
    behind the other wave -- and the point hook's state in LDS.  Its scratch is the lane-private first interval's
    (per-lane routing), outside the tip-aligned loop."""
    src = tmp_path / "kvr.hip"
    src.write_text(VERDICT_RETRACT_TU % (n_tendons, n_tendons))
    base = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-I", CSRC, str(src)]
    out = subprocess.run(base + ["-c", "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "kvr.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    get = lambda key: int(re.search(key + r"[^:]*: (\d+)", out.stderr).group(1))
    assert get("Occupancy") == 2
    assert get("ScratchSize") <= (700 if n_tendons == 3 else 1000)
    assert get("AGPRs") == 0
    asm = subprocess.run(base + ["-S", "-o", "-"], capture_output=True, text=True).stdout
    assert asm.count("ds_read") + asm.count("ds_write") >= 30


def test_isa_counts_are_current():
    """profiles/isa_counts.json -- the flops per RK4 step bench.py prices the fp64-VALU roofline with -- must be the
    count of THIS source tree's gfx950 assembly (profiles/count_isa.py rewrites it)."""
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("count_isa", os.path.join(ROOT, "profiles", "count_isa.py"))
    ci = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ci)
    tracked = json.load(open(ci.OUT))
    fresh = ci.count_all(["rk4_step<3>", "fk_sweep_fused<3,false>"])
    for k, v in fresh.items():
        assert tracked[k]["flops_per_step"] == v["flops_per_step"], "run `python profiles/count_isa.py`"
        assert tracked[k]["fp64_valu_instructions_per_step"] == v["fp64_valu_instructions_per_step"]
        # sanity of the count itself: an RK4 step is 4 evaluations of a ~770-flop right-hand side plus the stage updates
        assert 2500 < v["flops_per_step"] < 4500 and v["opcodes"].get("v_rcp_f64", 0) >= 8
