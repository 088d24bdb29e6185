for d in "" "-DCLOUDY_ABLATE_NODES=1" "-DCLOUDY_ABLATE_LATE=1" "-DCLOUDY_ABLATE_PTOP=1" "-DCLOUDY_ABLATE_INV=1" "-DCLOUDY_ABLATE_PAIR=1"; do
  echo "DEFS=$d"; CLOUDY_HIP_JIT_DEFS="$d" python tools/time_kernels.py --reps 3 cfg3b cfg4 moving4 2>&1 | tail -1
done
