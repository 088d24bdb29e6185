#!/bin/bash
# timing of pass variants through CLOUDY_HIP_JIT_DEFS; GPU box only
for r in 1 2; do
for d in "" "-DCLOUDY_RANK_TRIPS=2.0f" "-DCLOUDY_RANK_TRIPS=1.0f" "-DCLOUDY_RANK_TRIPS=4.0f"; do
  echo "DEFS=$d"; CLOUDY_HIP_JIT_DEFS="$d" python tools/time_kernels.py --reps 8 cfg3b cfg4 2>&1 | tail -1
done; done
