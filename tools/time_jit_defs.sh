#!/bin/bash
# A/B timing of plan-time compiled kernel variants on the GPU box: every argument is one CLOUDY_HIP_JIT_DEFS string
# (macros read by kernels.hpp, e.g. "-DCLOUDY_EARLY_SERIES=24 -DCLOUDY_EARLY_TMAX=2.0 -DCLOUDY_EARLY_UA=1.0",
# "-DCLOUDY_ABLATE_LATE=1"); the default build is timed first and between rounds.  WORKLOADS selects the workloads.
#   gpurun -- 'bash tools/time_jit_defs.sh "-DCLOUDY_EARLY_SERIES=32 -DCLOUDY_EARLY_TMAX=3.5 -DCLOUDY_EARLY_UA=1.5"'
WORKLOADS=${WORKLOADS:-"cfg3b cfg4 moving4"}
for round in 1 2; do
  for d in "" "$@"; do
    echo "DEFS=$d"
    CLOUDY_HIP_JIT_DEFS="$d" python tools/time_kernels.py --reps 8 $WORKLOADS 2>&1 | tail -1
  done
done
