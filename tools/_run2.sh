bash tools/pmc_one.sh r6_long_split tools/time_conv_one.py long 1,1,1 4000000 3 > /dev/null 2>&1
CLOUDY_HIP_LONG_SPLIT=0 bash tools/pmc_one.sh r6_long_nosplit tools/time_conv_one.py long 1,1,1 4000000 3 > /dev/null 2>&1
cat gpurun_out/r6_long_split_pmc.txt gpurun_out/r6_long_nosplit_pmc.txt
