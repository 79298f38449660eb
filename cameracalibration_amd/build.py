"""Build recipe of libbevwarp.so (hipcc, gfx950 only).  Used by __graft_entry__.build() and by hand:

    python -m cameracalibration_amd.build [--force]

Three translation units -- bevwarp.hip (handles, table builders, tools, the camera-per-GPU exchange), bevwarp_plan.hip (the tile plan
and its per-frame kernels) and bevwarp_jpeg.hip (the JPEG codec) -- are compiled in parallel into objects under csrc/build/ and linked;
a unit is recompiled when its source, ANY header under csrc/ or the flag set changed.  The shared object is written next to this file (in-tree: it travels
to the GPU box with the repo snapshot and is git-ignored).  -ffp-contract=off is REQUIRED: the arithmetic being reproduced has no fused
multiply-add.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(HERE, "libbevwarp.so")
# translation units; every header under csrc/ (and the public header) is a dependency of every unit: a header edit can never leave a
# stale object behind (round 4's explicit lists had missed bevw_jpeg_walk.h and bevw_device.h for the JPEG unit)
UNITS = ["bevwarp.hip", "bevwarp_plan.hip", "bevwarp_jpeg.hip"]


def _headers():
    import glob
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(HERE, "..", "include", "bevwarp.h")]


CFLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-Wno-pass-failed", "-Wno-inline-asm"]
EXTRA = os.environ.get("BEVW_CFLAGS", "").split()   # experiment builds (e.g. -DBEVW_UNIT_DEPTH=4)
TAG = os.environ.get("BEVW_BUILD_TAG", "")            # ... land in build_var/libbevwarp_<tag>.so (objects in csrc/build_<tag>/); A/B through BEVW_LIB_PATH
if TAG:
    OBJ = os.path.join(CSRC, "build_" + TAG)
    LIB = os.path.join(os.path.dirname(HERE), "build_var", "libbevwarp_%s.so" % TAG)


def _newer(path: str, deps) -> bool:
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if EXTRA and not TAG:
        # experiment flags never land in the default objects / libbevwarp.so (a later plain build would keep them: fresh mtimes)
        raise SystemExit("BEVW_CFLAGS needs BEVW_BUILD_TAG=<tag>: experiment builds go to build_var/libbevwarp_<tag>.so")
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    me = os.path.abspath(__file__)
    # the flag set the objects were compiled with: a different one (another BEVW_CFLAGS under the same tag) rebuilds everything
    stamp, flags = os.path.join(OBJ, "flags.stamp"), " ".join(CFLAGS + EXTRA)
    if not os.path.exists(stamp) or open(stamp).read() != flags:
        force = True
    jobs = []
    hdrs = _headers()
    for src in UNITS:
        obj = os.path.join(OBJ, src.replace(".hip", ".o"))
        deps = [os.path.join(CSRC, src)] + hdrs + [me]
        if force or _newer(obj, deps):
            jobs.append([hipcc] + CFLAGS + EXTRA + ["-c", os.path.join(CSRC, src), "-o", obj])
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in UNITS]
    if not jobs and not _newer(LIB, objs):
        return LIB

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=len(UNITS)) as pool:
        list(pool.map(run, jobs))
    with open(stamp, "w") as f:
        f.write(flags)
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared"] + objs + ["-o", LIB + ".tmp"])
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
