#!/usr/bin/env python3
"""Counts the fp64 VALU work of the RK4 loop of the hot kernels from the gfx950 ISA hipcc emits, and writes
profiles/isa_counts.json -- the figure bench.py turns into `roofline.achieved` for the fp64-VALU roofline
(flops per check = flops per RK4 step x steps per configuration; FMA = 2 flops).

    python profiles/count_isa.py            # rewrite profiles/isa_counts.json
    python profiles/count_isa.py --check    # exit 1 if the tracked file is stale (tests/test_kernel_resources.py does the same)

How: the kernel is compiled to assembly (`hipcc -S --cuda-device-only`, same flags as the build); LLVM annotates
every basic block with the loop it belongs to ("in Loop: Header=BBx_y"); the RK4 loop is the loop holding the
most v_fma_f64 instructions; every v_*_f64 instruction in its blocks is counted once.  Blocks of loops nested inside it
(the DDA walk of fk_verdict's deferred segments) are not counted, and neither are `rare_blocks`: blocks of the RK4 loop
that hold IEEE division / square-root expansions (v_div_scale_f64 ...), which the RK4 step itself never uses -- in
fk_verdict they are the set-up of the deferred walks, executed for ~3 % of the points by a few lanes.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "interactive-rate-tendons_amd", "csrc")
OUT = os.path.join(ROOT, "profiles", "isa_counts.json")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off"]

# flops per instruction; everything else matching v_*_f64 (moves, compares, conversions, min/max, ldexp ...) counts 0
FLOPS = {"v_fma_f64": 2, "v_fmac_f64": 2, "v_mul_f64": 1, "v_add_f64": 1, "v_rcp_f64": 1, "v_rsq_f64": 1, "v_sqrt_f64": 1,
         "v_div_fmas_f64": 2, "v_div_fixup_f64": 1, "v_div_scale_f64": 1, "v_pk_fma_f64": 4, "v_pk_mul_f64": 2,
         "v_pk_add_f64": 2}

_RK4_ONLY = r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "fk_kernel.hpp"
namespace trk {
// the RK4 step as fk_verdict and fk_sweep_fused hold it (no rotation, no R output, no backbone-length quadrature), without
// any per-point hook: the reference count of useful flops per step
template <int N> __global__ __launch_bounds__(64, 2) void rk4_step_only(const double *states, int64_t n, int64_t ld, RobotK K,
                                                                         const double *tab, const StepK *steps, int nsteps, FkOut out) {
  fk_uniform_body<N, false, false, false>(states, n, ld, K, tab, steps, nsteps, out);
}
}
template __global__ void trk::rk4_step_only<%d>(const double*, int64_t, int64_t, RobotK, const double*, const StepK*, int, trk::FkOut);
'''

KERNELS = {
    # name -> (translation unit, steps per configuration are supplied by the caller: P - 1)
    "rk4_step<3>": _RK4_ONLY % 3,
    "rk4_step<4>": _RK4_ONLY % 4,
    "fk_verdict<3,false>": r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "verdict_kernel.hpp"
template __global__ void trk::fk_verdict<3, false, false>(const double*, int64_t, RobotK, const double*, const StepK*, int, double*, const trk::VerdictArgs*);
''',
    "fk_verdict<4,false>": r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "verdict_kernel.hpp"
template __global__ void trk::fk_verdict<4, false, false>(const double*, int64_t, RobotK, const double*, const StepK*, int, double*, const trk::VerdictArgs*);
''',
    "fk_sweep_fused<3,false>": r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "fused_kernel.hpp"
template __global__ void trk::fk_sweep_fused<3, false>(const double*, int64_t, int64_t, RobotK, const double*, const StepK*, int,
                                                       trk::FkOut, const trk::FusedSweepArgs*);
''',
    "fk_sweep_fused<4,false>": r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "fused_kernel.hpp"
template __global__ void trk::fk_sweep_fused<4, false>(const double*, int64_t, int64_t, RobotK, const double*, const StepK*, int,
                                                       trk::FkOut, const trk::FusedSweepArgs*);
''',
    "fk_rk4_batch_uniform<3,false,false>": r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "fk_kernel.hpp"
template __global__ void trk::fk_rk4_batch_uniform<3, false, false>(const double*, int64_t, int64_t, RobotK,
                                                                     const double*, const StepK*, int, trk::FkOut);
''',
    "fk_rk4_batch_uniform<4,false,false>": r'''
#include <hip/hip_runtime.h>
#include "tr_types.hpp"
#include "fk_kernel.hpp"
template __global__ void trk::fk_rk4_batch_uniform<4, false, false>(const double*, int64_t, int64_t, RobotK,
                                                                     const double*, const StepK*, int, trk::FkOut);
''',
}

_BLOCK = re.compile(r"^(\.LBB\d+_\d+):\s*(?:;\s*(.*))?$")
_FALL = re.compile(r"^; (%bb\.\d+):\s*(?:;\s*(.*))?$")          # fall-through block without a label
_HDR = re.compile(r"Header=(BB\d+_\d+)")
_INSN = re.compile(r"^\s+([vs]_[a-z0-9_]+)")


def count_asm(asm):
    """-> dict for the loop with the most v_fma_f64: histogram of v_*_f64 opcodes, totals."""
    blocks = []                                               # (loop header or None, opcode histogram, #VALU) per basic block
    cur, hist, nv = None, collections.Counter(), 0
    for line in asm.splitlines():
        m = _BLOCK.match(line) or _FALL.match(line)
        if m:
            blocks.append((cur, hist, nv))
            hist, nv = collections.Counter(), 0
            note = m.group(2) or ""
            if "Loop Header" in note and "This" in note:
                cur = m.group(1)[2:] if m.group(1).startswith(".L") else None     # a loop header always carries a label
            else:
                h = _HDR.search(note)
                cur = h.group(1) if h else None
            continue
        m = _INSN.match(line)
        if not m:
            continue
        op = m.group(1)
        for suffix in ("_e32", "_e64", "_dpp", "_sdwa"):
            if op.endswith(suffix):
                op = op[: -len(suffix)]
        if op.startswith("v_"):
            nv += 1
            if op.endswith("_f64"):
                hist[op] += 1
    blocks.append((cur, hist, nv))
    loops = collections.defaultdict(collections.Counter)
    valu, rare = collections.Counter(), collections.Counter()
    for hdr, h, n in blocks:
        if hdr is None:
            continue
        if any(op.startswith("v_div_") for op in h):
            rare[hdr] += 1
            continue
        loops[hdr].update(h)
        valu[hdr] += n
    if not loops:
        raise RuntimeError("no fp64 loop found in the assembly")
    hdr = max(loops, key=lambda h: loops[h]["v_fma_f64"] + loops[h]["v_fmac_f64"])
    hist = dict(sorted(loops[hdr].items()))
    return {"loop": hdr, "fp64_valu_instructions_per_step": int(sum(hist.values())),
            "valu_instructions_per_step": int(valu[hdr]), "rare_blocks": int(rare[hdr]),
            "flops_per_step": int(sum(FLOPS.get(op, 0) * c for op, c in hist.items())),
            "opcodes": hist}


def compile_asm(tu, workdir):
    src = os.path.join(workdir, "k.hip")
    with open(src, "w") as f:
        f.write(tu)
    out = os.path.join(workdir, "k.s")
    subprocess.run(["hipcc"] + FLAGS + ["--cuda-device-only", "-S", "-I", CSRC, src, "-o", out], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def count_all(names=None):
    res = {}
    for name, tu in KERNELS.items():
        if names and name not in names:
            continue
        with tempfile.TemporaryDirectory() as d:
            res[name] = count_asm(compile_asm(tu, d))
    return res


def main():
    res = count_all()
    if "--check" in sys.argv:
        old = json.load(open(OUT))
        bad = [k for k in res if old.get(k, {}).get("flops_per_step") != res[k]["flops_per_step"]]
        if bad:
            print("stale:", bad)
            sys.exit(1)
        return
    json.dump(res, open(OUT, "w"), indent=1, sort_keys=True)
    for k, v in res.items():
        print(k, v["fp64_valu_instructions_per_step"], "fp64 instr/step,", v["flops_per_step"], "flop/step")


if __name__ == "__main__":
    main()
