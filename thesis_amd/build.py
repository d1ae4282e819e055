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
SOURCES = ["rbpf_api.hip", "kernels_weight.hip", "kernels_mapupdate.hip", "kernels_mapray.hip", "kernels_mapev.hip", "kernels_state.hip",
           "kernels_propose.hip", "kernels_resample.hip", "kernels_match.hip", "kernels_inputs.hip"]
HEADERS = ["rbpf_internal.h", "rbpf_math.h", "rbpf_device.h", "rbpf_mapupdate.h", os.path.join("..", "..", "include", "rbpf_hip.h")]
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-ffp-contract=off",
         "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


# LDS atomics whose address the compiler can prove wave-uniform (step 0 of every ray: the start cell; the matcher's score
# cells of a coarse rotation) are expanded by LLVM's atomic optimizer into a 64-iteration scan loop per wave - the LDS
# serialises those lanes faster by itself.  Measured (round 3): event walk -12 %, match_kernel 0.577 -> 0.560 ms at 4096,
# ray kernel 3.14 -> 3.08 ms at C5.  kernels_propose / kernels_resample showed no difference and keep the default.
_NO_ATOMIC_SCAN = ["-mllvm", "-amdgpu-atomic-optimizer-strategy=None"]
PER_FILE_FLAGS = {f: _NO_ATOMIC_SCAN for f in ("kernels_mapev.hip", "kernels_match.hip", "kernels_mapray.hip", "kernels_mapupdate.hip")}


def _newer(path: str, t: float) -> bool:
    return os.path.getmtime(path) > t


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    if not os.environ.get("RBPF_STAMPS") and any(os.path.exists(os.path.join(CSRC, s.replace(".hip", ".o.stamped"))) for s in SOURCES):
        return True                                     # a diagnostic (stamped) object must not survive into a normal build
    return any(_newer(d, t) for d in deps)


def build_extension(force: bool = False, verbose: bool = True) -> str:
    """Compiles what changed (a source is recompiled when it, a header or this script is newer than its object;
    sources compile in parallel) and links."""
    if not force and not _stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    stamp_target = os.environ.get("RBPF_STAMPS", "")   # diagnostic builds: per-phase cycle stamps in ONE kernel file (mapev | mapupdate | mapray | match)
    common = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, jobs = [], []
    for src in SOURCES:
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(obj)
        path = os.path.join(CSRC, src)
        stamped = bool(stamp_target) and stamp_target in src
        fresh = os.path.exists(obj) and not any(_newer(d, os.path.getmtime(obj)) for d in [path] + common)
        if fresh and not force and not stamped and not os.path.exists(obj + ".stamped"):
            continue
        extra = (["-DRBPF_STAMPS"] + os.environ.get("RBPF_STAMP_DEFS", "").split()) if stamped else []   # e.g. RBPF_STAMP_DEFS="-DABLATE=1"
        extra += PER_FILE_FLAGS.get(src, [])
        jobs.append(([hipcc, *FLAGS, *extra, "-c", path, "-o", obj], obj, stamped))

    def run(job):
        cmd, obj, stamped = job
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        marker = obj + ".stamped"                       # a stamped object must not survive into a normal build
        if stamped:
            open(marker, "w").close()
        elif os.path.exists(marker):
            os.remove(marker)

    with ThreadPoolExecutor(max(1, min(len(jobs), int(os.environ.get("RBPF_BUILD_JOBS", "6"))))) as ex:
        list(ex.map(run, jobs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build_extension(force="--force" in sys.argv)
    print(LIB)
