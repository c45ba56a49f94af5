#!/bin/bash
# usage (on the GPU box): bash tools/probes/bisect/run_bisect.sh classes | ranges LO HI [PARTS]
set -e
R=/root/repo; W=/tmp/bisect; rm -rf $W; mkdir -p $W; cd $W
hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -S --cuda-device-only $R/tools/probes/geometry_lane_probe.hip -o probe.s 2>/dev/null
hipcc -O2 $R/tools/probes/bisect/co_runner.cpp -o co_runner 2>/dev/null
python $R/tools/probes/bisect/make_variants.py probe.s $W "$@"
CL=/opt/rocm/lib/llvm/bin
for f in $W/[A-Z]_*.s; do
  t=$(basename $f .s)
  $CL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $f -o $t.o 2>$t.err || { echo "$t: assemble failed: $(head -2 $t.err)"; continue; }
  $CL/ld.lld -shared $t.o -o $t.co
  echo "$t: $(timeout 120 ./co_runner $t.co 40)"
done
