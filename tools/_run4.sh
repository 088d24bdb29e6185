mkdir -p gpurun_out/r6c
timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "closure_stats or lds_fits or error_returns or column or rainshaft" 2>&1 | tail -15 > gpurun_out/r6c/tests.txt
timeout 1500 python -m pytest tests/test_gpu_numerical.py -x -q -m gpu -k "cost_hints or converged_mode_vs_same_rule" 2>&1 | tail -5 >> gpurun_out/r6c/tests.txt
python tools/time_conv_one.py long 1,1,1 12500000 3 > gpurun_out/r6c/time_long_split.txt 2>&1
python tools/time_conv_one.py hydrodynamic 1,1,1 12500000 3 > gpurun_out/r6c/time_hydro.txt 2>&1
bash tools/pmc_one.sh r6c_long_split tools/time_conv_one.py long 1,1,1 4000000 3 > /dev/null 2>&1
bash tools/pmc_one.sh r6c_hydro tools/time_conv_one.py hydrodynamic 1,1,1 4000000 3 > /dev/null 2>&1
cat gpurun_out/r6c/*.txt; grep -A2 "cloudy_jit" gpurun_out/r6c_long_split_pmc.txt gpurun_out/r6c_hydro_pmc.txt | grep "per lane"
