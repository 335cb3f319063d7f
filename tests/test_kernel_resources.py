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


SEARCH_TU = r'''
#include <hip/hip_runtime.h>
#include "search_kernel.hpp"
template __global__ void trk::roadmap_astar<%d>(trk::SearchArgs);
'''


@pytest.mark.parametrize("sx", [4, 12])
def test_search_kernel_occupancy_and_single_ticket_site(tmp_path, sx):
    """roadmap_astar: 12 waves per CU (<= 168 VGPRs), no register spilled to scratch beyond a few dwords, and ONE query loop: hipcc 7.2 once jump-threaded the back edge of that loop for the 63 lanes that did not draw the
    ticket (`if (lane == 0) ticket = atomicAdd(..)`) into a copy of the loop that lane 0 was not part of -- cross-lane operations
    without the lane that writes the list heads, a GPU memory fault.  The ticket is now an atomic every lane executes; a threaded
    or duplicated loop would show as a second ticket site.  Returning dword adds in the listing: the ticket and the path buffer's."""
    src = tmp_path / "ks.hip"
    src.write_text(SEARCH_TU % sx)
    base = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-I", CSRC, str(src)]
    out = subprocess.run(base + ["-c", "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "ks.o")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    get = lambda key: int(re.search(key + r"[^:]*: (\d+)", out.stderr).group(1))
    # (12 waves per CU: a step holds two passes' arcs, records and heuristics at once; 168 registers is where a wave per SIMD would be lost)
    # (the 12-coordinate form keeps a whole state row in registers: two waves per SIMD there)
    assert get("Occupancy") >= (3 if sx == 4 else 2) and get("VGPRs") <= (168 if sx == 4 else 192) and get("AGPRs") == 0 and get("VGPRs Spill") == 0
    assert get("ScratchSize") <= 64
    asm = subprocess.run(base + ["-S", "-o", "-"], capture_output=True, text=True).stdout
    returning_adds = re.findall(r"^\s*(?:global|flat)_atomic_add v\d+, .*\bsc0\b", asm, flags=re.M)
    assert len(returning_adds) == 2, returning_adds


EDGE_QUEUE_TU = r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "verdict_kernel.hpp"
#include "edge_kernel.hpp"
#include "edge_queue_kernel.hpp"
template __global__ void trk::fk_edge_queue<4, false>(RobotK, const double*, const StepK*, int, const trk::VerdictArgs*, const trk::EdgeQueueArgs*,
                                                      const trk::FusedSweepArgs*);
'''


def test_edge_queue_claim_sites_are_not_duplicated(tmp_path):
    """fk_edge_queue holds the same loop shape (`if (lane == 0) { atomics }` -> broadcast -> loop back, eq_claim and the main
    loop): its returning atomic adds appear once each in the listing -- the AVAIL and HEAD claims of eq_claim, the fold's
    `remaining` decrement, the tail allocation of an edge's next level, and the counter of the verdict body's list of
    configurations for the exact sweep.  A jump-threaded copy of one of those loops would duplicate its site."""
    src = tmp_path / "kq.hip"
    src.write_text(EDGE_QUEUE_TU)
    asm = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "--cuda-device-only", "-I", CSRC,
                          str(src), "-S", "-o", "-"], capture_output=True, text=True)
    assert asm.returncode == 0, asm.stderr[-2000:]
    returning_adds = re.findall(r"^\s*(?:global|flat)_atomic_add v\d+, .*\bsc0\b", asm.stdout, flags=re.M)
    assert len(returning_adds) == 5, returning_adds


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
