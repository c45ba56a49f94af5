"""Build recipe for the CPU oracle's C restatement (TEST INFRASTRUCTURE ONLY).

`python -m oracle.build` compiles oracle/csrc/oracle_ops.c with gcc into
oracle/_build/liboracle_ops.so.  -ffp-contract=off keeps every float32
operation separate (no FMA fusing) so the arithmetic order matches the
reference's device code (see oracle/csrc/oracle_ops.c header).

oracle/_ref/: the reference's own native sources for this path
(lib/model/nms/src/nms_cuda_kernel.cu, lib/model/roi_align/src/roi_align_kernel.cu)
are CUDA translation units that need nvcc and the PyTorch-0.3 THC headers
(nms_cuda.c:1, roi_align_cuda.c:1); neither exists in this image and they have
no CPU branch, so the reference is UNBUILDABLE here and oracle/_ref/ stays
empty (documented in DESIGN.md).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "oracle_ops.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "liboracle_ops.so")


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if (not force and os.path.exists(OUT)
            and os.path.getmtime(OUT) >= os.path.getmtime(SRC)):
        return OUT
    cmd = ["gcc", "-O2", "-std=c99", "-ffp-contract=off", "-fno-fast-math",
           "-shared", "-fPIC", SRC, "-o", OUT, "-lm"]
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
