mkdir -p gpurun_out/r6e
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "closure_stats" 2>&1 | tail -30 > gpurun_out/r6e/tests.txt
CLOUDY_HIP_CONV_ROUNDS=2 timeout 1500 python -m pytest tests/test_gpu_numerical.py -x -q -m gpu -k "cost_hints or converged_mode_vs_same_rule or golden" 2>&1 | tail -12 >> gpurun_out/r6e/tests.txt
for r in 1 2 4; do
  CLOUDY_HIP_CONV_ROUNDS=$r python tools/time_conv_one.py hydrodynamic 1,1,1 12500000 3 > gpurun_out/r6e/time_hydro_rounds$r.txt 2>&1
done
CLOUDY_HIP_CONV_BLOCK=128 python tools/time_conv_one.py hydrodynamic 1,1,1 12500000 3 > gpurun_out/r6e/time_hydro_wg128.txt 2>&1
CLOUDY_HIP_CONV_BLOCK=128 python tools/time_conv_one.py long 1,1,1 12500000 3 > gpurun_out/r6e/time_long_wg128.txt 2>&1
CLOUDY_HIP_CONV_ROUNDS=2 python tools/time_conv_one.py linear 3,3 2000000 3 > gpurun_out/r6e/time_lin33_rounds2.txt 2>&1
CLOUDY_HIP_CONV_ROUNDS=1 python tools/time_conv_one.py linear 3,3 2000000 3 > gpurun_out/r6e/time_lin33_rounds1.txt 2>&1
CLOUDY_HIP_CONV_ROUNDS=2 bash tools/pmc_one.sh r6e_hydro_r2 tools/time_conv_one.py hydrodynamic 1,1,1 4000000 3 > /dev/null 2>&1
WORKLOADS="cfg3b cfg4" bash tools/time_jit_defs.sh "-DCLOUDY_F32_TAIL=1" > gpurun_out/r6e/f32_tail.txt 2>&1
CLOUDY_HIP_JIT_DEFS="-DCLOUDY_F32_TAIL=1" python tools/time_kernels.py --reps 3 --error cfg3b cfg4 >> gpurun_out/r6e/f32_tail.txt 2>&1
python tools/time_kernels.py --reps 3 --error cfg3b cfg4 >> gpurun_out/r6e/f32_tail.txt 2>&1
for f in gpurun_out/r6e/*.txt; do echo "== $f"; cat $f; done; grep -A2 "cloudy_jit" gpurun_out/r6e_hydro_r2_pmc.txt | grep "per lane"
