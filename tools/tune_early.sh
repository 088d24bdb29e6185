#!/bin/bash
# timing and worst parity error of early-series variants (terms / radius), through CLOUDY_HIP_JIT_DEFS; GPU box only
for d in "" "-DCLOUDY_EARLY_SERIES=30 -DCLOUDY_EARLY_TMAX=3.0 -DCLOUDY_EARLY_UA=1.5" "-DCLOUDY_EARLY_SERIES=32 -DCLOUDY_EARLY_TMAX=3.5 -DCLOUDY_EARLY_UA=1.5" "-DCLOUDY_EARLY_SERIES=30 -DCLOUDY_EARLY_TMAX=3.0 -DCLOUDY_EARLY_UA=2.0"; do
  echo "DEFS=$d"; CLOUDY_HIP_JIT_DEFS="$d" python tools/time_kernels.py --reps 8 cfg3b cfg4 moving4 2>&1 | tail -1
  CLOUDY_HIP_JIT_DEFS="$d" python tools/fuzz_parity.py --configs 40 --seed 5 2>&1 | tail -3
done
