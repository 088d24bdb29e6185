#!/bin/bash
# Compile a DUMPED plan translation unit against edited kernel headers without rebuilding libcloudy_hip.so (no GPU needed; seconds):
#   1. python tools/jit_resources.py --keep /tmp/jd cfg4q_converged      (or conv:long:1,1,1 ...) leaves /tmp/jd/cloudy_plan_*.hip
#   2. cp cloudy.jl_amd/csrc/*.hpp cloudy.jl_amd/csrc/*.inc /tmp/hx/ ; edit /tmp/hx/quad_conv.hpp ...
#   3. tools/jit_offline.sh /tmp/hx /tmp/jd/cloudy_plan_<id>.hip out.co  -> spilled registers, scratch and LDS bytes of every kernel
#   4. python tools/isa_loops.py out.co <kernel>                         -> instructions / scratch / LDS accesses per loop
# The options are hiprtc's of jit.hpp (jit_compile); machine LICM stays on as for the RHS kernels.
HDR=$1; TU=$2; OUT=$3; shift 3
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=fast "-DINFINITY=__builtin_huge_val()" -DCLOUDY_JIT=1 \
    -mllvm -pragma-unroll-threshold=100000 -I"$HDR" --cuda-device-only --no-gpu-bundle-output -c "$TU" -o "$OUT" "$@" || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes "$OUT" | grep -E "\.name:|vgpr_count|vgpr_spill_count|private_segment_fixed_size|group_segment_fixed_size" | paste - - - - - | sed 's/  */ /g'
