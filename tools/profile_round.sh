#!/bin/bash
# Collects the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r01   ->  gpurun_out/<tag>_*  (copy the summaries into profiles/ afterwards)
# Kernel trace/stats and PMC passes are separate runs; FETCH_SIZE and WRITE_SIZE do not fit one TCC pass.
TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
mkdir -p $OUT
python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cp bench_variants.json $OUT/${TAG}_bench_variants.json
# (the profiled passes leave out the 1e8-parcel launch of configs[3]: every kernel name keeps ONE launch size in the stats)
export CLOUDY_BENCH_SKIP_FULL=1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_stats -- python3 bench.py --steps 200 --warmup 50 --no-cpu-baseline > $OUT/${TAG}_stats_bench.json 2> $OUT/${TAG}_stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmc_write.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmc_sq.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc_sq2 -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2> $OUT/${TAG}_pmc_sq2.err
./tools/hbm_ceiling 10000000 80 > $OUT/${TAG}_hbm_ceiling.txt 2>&1
find $OUT -name "*kernel_stats.csv" -newer $OUT/${TAG}_bench.json | head
cat $OUT/${TAG}_bench.json
