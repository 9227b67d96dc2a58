"""Build the gfx950 HIP library in-tree: csrc/*.hip -> libvistaf_ftp.so (hipcc, no torch dependency)."""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "_obj")
LIB = os.path.join(HERE, "libvistaf_ftp.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]
FLAGS += os.environ.get("VISTAF_EXTRA_HIPCC_FLAGS", "").split()      # e.g. -DVISTAF_DEBUG: shader-clock stamps of the march kernels (diagnostics only)


def _newer(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + [os.path.join(HERE, "..", "include", "vistaf_ftp.h"), os.path.join(HERE, "..", "include", "vistaf_align.h"),
                                                     os.path.join(HERE, "..", "include", "vistaf_temp.h")]
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        if force or _newer(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or force or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-lhipfft"])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
