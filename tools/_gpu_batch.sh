O=gpurun_out
python tools/fuzz_parity.py --configs 100 --seed 61 > $O/r06_fuzz_tensor.txt 2>&1
python tools/fuzz_parity.py --wild --configs 50 --seed 62 > $O/r06_fuzz_tensor_wild.txt 2>&1
python tools/fuzz_parity.py --converged --configs 60 --seed 63 > $O/r06_fuzz_converged.txt 2>&1
python tools/fuzz_parity.py --converged --wild --configs 40 --seed 64 > $O/r06_fuzz_converged_wild.txt 2>&1
python tools/fuzz_parity.py --numerical --configs 40 --seed 65 > $O/r06_fuzz_numerical.txt 2>&1
python tools/fuzz_parity.py --big --configs 10 --seed 66 > $O/r06_fuzz_tensor_big.txt 2>&1
python tools/fuzz_parity.py --converged --big --configs 6 --seed 67 > $O/r06_fuzz_converged_big.txt 2>&1
python tools/conv_wild_device.py 1500 > $O/r06_converged_wild_device.txt 2>&1
tail -n 3 $O/r06_fuzz_*.txt $O/r06_converged_wild_device.txt
