mkdir -p gpurun_out/r6d
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lds_fits or plans_beyond or closure_stats or error_returns" 2>&1 | tail -12 > gpurun_out/r6d/tests.txt
for bs in 256 512; do
  CLOUDY_HIP_CONV_BLOCK=$bs python tools/time_conv_one.py hydrodynamic 1,1,1 12500000 3 > gpurun_out/r6d/time_hydro_$bs.txt 2>&1
  CLOUDY_HIP_CONV_BLOCK=$bs python tools/time_conv_one.py long 1,1,1 12500000 3 > gpurun_out/r6d/time_long_$bs.txt 2>&1
done
CLOUDY_HIP_CONV_BLOCK=1024 python tools/time_conv_one.py hydrodynamic 1,1,1 12500000 3 > gpurun_out/r6d/time_hydro_1024.txt 2>&1
CLOUDY_HIP_CONV_BLOCK=512 timeout 900 python -m pytest tests/test_gpu_numerical.py -x -q -m gpu -k "cost_hints or converged_mode_vs_same_rule" 2>&1 | tail -3 >> gpurun_out/r6d/tests.txt
CLOUDY_HIP_CONV_BLOCK=512 bash tools/pmc_one.sh r6d_hydro512 tools/time_conv_one.py hydrodynamic 1,1,1 4000000 3 > /dev/null 2>&1
CLOUDY_HIP_CONV_BLOCK=512 bash tools/pmc_one.sh r6d_long512 tools/time_conv_one.py long 1,1,1 4000000 3 > /dev/null 2>&1
cat gpurun_out/r6d/*.txt; grep -A2 "cloudy_jit" gpurun_out/r6d_hydro512_pmc.txt gpurun_out/r6d_long512_pmc.txt | grep "per lane"
