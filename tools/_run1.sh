set -x
mkdir -p gpurun_out/r6a
timeout 1500 python -m pytest tests/test_gpu_numerical.py -x -q -m gpu -k "long or cost_hints or converged_mode_vs_same_rule or fused_ssprk33" 2>&1 | tail -15 > gpurun_out/r6a/tests.txt
python tools/time_conv_one.py long 1,1,1 12500000 3 > gpurun_out/r6a/time_long_split.txt 2>&1
CLOUDY_HIP_LONG_SPLIT=0 python tools/time_conv_one.py long 1,1,1 12500000 3 > gpurun_out/r6a/time_long_nosplit.txt 2>&1
python tools/time_conv_one.py hydrodynamic 1,1,1 12500000 3 > gpurun_out/r6a/time_hydro.txt 2>&1
cat gpurun_out/r6a/*.txt
