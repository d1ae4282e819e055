"""Build librbpf_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m thesis_amd.build` or `thesis_amd.build.build_extension()`.  hipcc cross-compiles
for gfx950 without a GPU; the shared object lands next to this file so that it travels with the
tree and is the one the Python layer loads.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librbpf_hip.so")
SOURCES = ["rbpf_api.hip", "kernels_weight.hip", "kernels_mapupdate.hip", "kernels_mapfan.hip", "kernels_state.hip",
           "kernels_propose.hip", "kernels_resample.hip", "kernels_match.hip", "kernels_inputs.hip"]
HEADERS = ["rbpf_internal.h", "rbpf_math.h", "rbpf_device.h", "rbpf_mapupdate.h", os.path.join("..", "..", "include", "rbpf_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_extension(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    extra = []   # diagnostic builds: per-phase cycle stamps in ONE kernel (RBPF_STAMPS=mapupdate | match)
    stamp_target = os.environ.get("RBPF_STAMPS", "")
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        extra = ["-DRBPF_STAMPS"] if stamp_target and stamp_target in src else []
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_extension(force="--force" in sys.argv)
    print(LIB)
