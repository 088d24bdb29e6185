mkdir -p gpurun_out/r6b
timeout 1500 python -m pytest tests/test_gpu_numerical.py -x -q -m gpu -k "long or cost_hints or converged_mode_vs_same_rule or fused_ssprk33 or golden" 2>&1 | tail -5 > gpurun_out/r6b/tests.txt
python tools/time_conv_one.py long 1,1,1 12500000 3 > gpurun_out/r6b/time_long_split.txt 2>&1
CLOUDY_HIP_LONG_SPLIT=0 python tools/time_conv_one.py long 1,1,1 12500000 3 > gpurun_out/r6b/time_long_nosplit.txt 2>&1
python tools/time_conv_one.py hydrodynamic 1,1,1 12500000 3 > gpurun_out/r6b/time_hydro.txt 2>&1
python tools/time_conv_one.py linear 3,3 2000000 3 > gpurun_out/r6b/time_lin33.txt 2>&1
bash tools/pmc_one.sh r6b_long_split tools/time_conv_one.py long 1,1,1 4000000 3 > /dev/null 2>&1
bash tools/pmc_one.sh r6b_hydro tools/time_conv_one.py hydrodynamic 1,1,1 4000000 3 > /dev/null 2>&1
cat gpurun_out/r6b/*.txt; grep -A2 "cloudy_jit" gpurun_out/r6b_long_split_pmc.txt gpurun_out/r6b_hydro_pmc.txt | grep "per lane"
