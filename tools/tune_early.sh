#!/bin/bash
# timing (and worst parity error) of pass variants through CLOUDY_HIP_JIT_DEFS; GPU box only
for r in 1 2; do
for d in "" "-DCLOUDY_LATE_ASCENDING=1"; do
  echo "DEFS=$d"; CLOUDY_HIP_JIT_DEFS="$d" python tools/time_kernels.py --reps 8 cfg3b cfg4 moving4 2>&1 | tail -1
done; done
python tools/fuzz_parity.py --configs 40 --seed 5 2>&1 | grep -v "^ok" | tail -3
