for w in "" 3 5; do for bs in "" 256 1024; do
  echo "WAVES=$w BS=$bs"; CLOUDY_HIP_JIT_WAVES=$w CLOUDY_HIP_JIT_SORTED_BS=$bs python tools/time_kernels.py --reps 8 cfg3b cfg4 moving4 2>&1 | tail -1
done; done
echo "tol 1e-14"; CLOUDY_HIP_JIT_DEFS="-DCLOUDY_SERIES_TOL=1e-14 -DCLOUDY_CF_TOL=1e-14" python tools/time_kernels.py --reps 8 --error cfg3b cfg4 moving4 2>&1 | tail -1
echo "default"; python tools/time_kernels.py --reps 8 --error cfg3b cfg4 moving4 2>&1 | tail -1
