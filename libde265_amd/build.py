"""Builds libde265_amd/libde265_hip.so (gfx950 only) with hipcc.

    python -m libde265_amd.build [--force]

The .so is built in-tree so that it travels with the repo snapshot to the GPU
box; hipcc cross-compiles without a GPU present.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SO = os.path.join(_HERE, "libde265_hip.so")
SOURCES = ["k_tu.hip", "k_mc.hip", "k_lf.hip", "k_rext.hip", "k_scan.hip", "host.hip", "vtable.hip", "pipeline.hip"]
HEADERS = ["dev_common.h", "kernels.h", "scan.h", "scan_core.h", "env.h", "dct_table.inc", os.path.join("..", "..", "include", "de265_hip.h"),
           os.path.join("..", "..", "include", "de265_hip_vtable.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + \
    os.environ.get("DE265HIP_EXTRA_CXXFLAGS", "").split()


def _newer(src, dst):
    return not os.path.exists(dst) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=False):
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    hdr_paths = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(h, obj) for h in hdr_paths):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)

    if jobs:
        with ThreadPoolExecutor(max_workers=4) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or not os.path.exists(SO):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-o", SO] + objs + ["-L/opt/rocm/lib", "-lhsa-runtime64"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return SO


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
