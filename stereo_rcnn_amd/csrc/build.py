"""Build libsrcnn_hip.so for gfx950 (in-tree, explicit hipcc; no JIT cache).

    python -m stereo_rcnn_amd.csrc.build [--force]

-ffp-contract=off: the box/IoU/ROIAlign kernels must round every float32 operation
separately to match the reference's arithmetic order (MFMA code is unaffected).
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(os.path.dirname(HERE), "libsrcnn_hip.so")
OBJ_DIR = os.path.join(HERE, "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


def _newer(dst, srcs):
    if not os.path.exists(dst):
        return False
    t = os.path.getmtime(dst)
    return all(os.path.getmtime(s) <= t for s in srcs)


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(HERE, "*.hip")))
    hdrs = sorted(glob.glob(os.path.join(HERE, "*.h"))) + [os.path.join(ROOT, "include", "srcnn_hip.h")]
    if not force and _newer(OUT, srcs + hdrs):
        return OUT
    os.makedirs(OBJ_DIR, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        if force or not _newer(obj, [src] + hdrs):
            cmd = ["hipcc"] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(srcs))) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
