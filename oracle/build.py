"""Build recipe for the CPU oracle's C restatement (TEST INFRASTRUCTURE ONLY).

`python -m oracle.build` compiles oracle/csrc/oracle_ops.c with gcc into
oracle/_build/liboracle_ops.so.  -ffp-contract=off keeps every float32
operation separate (no FMA fusing) so the arithmetic order matches the
reference's device code (see oracle/csrc/oracle_ops.c header).

oracle/_ref/ (build_ref): the reference's own native sources for this path are two self-contained CUDA translation
units (lib/model/nms/src/nms_cuda_kernel.cu, lib/model/roi_align/src/roi_align_kernel.cu; the .c files next to them
are THC glue for PyTorch 0.3 and are not needed).  They compile unchanged with hipcc for gfx950 when the dozen CUDA
runtime names they use are spelled the HIP way by a force-included header (oracle/ref_cuda_on_hip.h):
    hipcc --offload-arch=gfx950 -x hip -include oracle/ref_cuda_on_hip.h <the two .cu files where they lie> -shared
-> oracle/_ref/libref_ops.so (default floating-point contraction, as nvcc's -fmad=true) and
   oracle/_ref/libref_ops_nofma.so (-ffp-contract=off).
Built only where /root/reference exists (this container); the .so files are git-ignored but travel to the GPU box,
where tests/test_ref_kernels_gpu.py runs the REFERENCE'S kernels on the MI355X against the C restatement and the
product kernels.  Nothing is copied from the reference into the repository.
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


REF_ROOT = "/root/reference/lib/model"
REF_SRCS = [os.path.join(REF_ROOT, "nms", "src", "nms_cuda_kernel.cu"),
            os.path.join(REF_ROOT, "roi_align", "src", "roi_align_kernel.cu")]
REF_DIR = os.path.join(HERE, "_ref")
REF_OUT = {"fma": os.path.join(REF_DIR, "libref_ops.so"), "nofma": os.path.join(REF_DIR, "libref_ops_nofma.so")}


def build_ref(force=False):
    """Compile the reference's own CUDA kernels for gfx950 (see the module docstring).  Returns the dict of outputs,
    or None where the reference tree is absent (the GPU box: it uses the prebuilt files)."""
    if not all(os.path.exists(p) for p in REF_SRCS):
        return None
    os.makedirs(REF_DIR, exist_ok=True)
    shim = os.path.join(HERE, "ref_cuda_on_hip.h")
    for kind, out in REF_OUT.items():
        if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(shim):
            continue
        cmd = ["hipcc", "--offload-arch=gfx950", "-O2", "-fPIC", "-shared", "-w", "-x", "hip", "-include", shim]
        cmd += ["-I" + os.path.dirname(p) for p in REF_SRCS]
        if kind == "nofma":
            cmd.append("-ffp-contract=off")
        subprocess.check_call(cmd + REF_SRCS + ["-o", out])
    return dict(REF_OUT)


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_ref(force="--force" in sys.argv))
