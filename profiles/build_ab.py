#!/usr/bin/env python3
"""An A/B build of libtendon_hip.so: one translation unit (or several, comma-separated) recompiled with extra flags, the others as built.
    python profiles/build_ab.py roadmap -DTRK_SEARCH_CLOCKS -o profiles/_ab/libtendon_hip_clocks.so
then  TENDON_HIP_LIB=profiles/_ab/libtendon_hip_clocks.so python profiles/probe_search.py"""
import importlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    L = importlib.import_module("interactive-rate-tendons_amd._lib")
    L.build()
    args = sys.argv[1:]
    out = os.path.join(ROOT, "profiles", "_ab", "libtendon_hip_ab.so")
    if "-o" in args:
        i = args.index("-o")
        out = os.path.abspath(args[i + 1])
        del args[i:i + 2]
    unit, extra = args[0], args[1:]
    os.makedirs(os.path.dirname(out), exist_ok=True)
    objs = []
    for obj, src, flags, _ in L._units():
        o = os.path.join(L.OBJ_DIR, obj)
        if obj[:-2] in unit.split(","):
            o = os.path.join(os.path.dirname(out), "ab_" + obj)
            subprocess.check_call(["hipcc"] + L.HIPCC_FLAGS + flags + extra + ["-c", os.path.join(L.SRC_DIR, src), "-o", o])
        objs.append(o)
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out] + objs)
    print(out)


if __name__ == "__main__":
    main()
