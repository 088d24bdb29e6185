#!/bin/bash
# Per-kernel VALU counters of one command on the GPU box: tools/pmc_quick.sh <tag> <python args...>
# (counters in their own passes, kernel trace only; summaries land in gpurun_out/<tag>_pmc*.txt)
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_pmc1 -- python3 "$@" > /dev/null 2> $OUT/${TAG}_pmc1.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc2 -- python3 "$@" > /dev/null 2> $OUT/${TAG}_pmc2.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/${TAG}_pmc3 -- python3 "$@" > /dev/null 2> $OUT/${TAG}_pmc3.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc4 -- python3 "$@" > /dev/null 2> $OUT/${TAG}_pmc4.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_pmc5 -- python3 "$@" > /dev/null 2> $OUT/${TAG}_pmc5.err
python3 - <<PY
import csv, glob, statistics, collections
for d in ("pmc1","pmc2","pmc3","pmc4","pmc5"):
    t=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/${TAG}_%s/*/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][-60:]
            t[(k,r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open("$OUT/${TAG}_%s.txt"%d,"w") as o:
        for k,cs in sorted(t.items()):
            o.write("%s grid=%s: "%k + ", ".join("%s=%.4g"%(c,statistics.median(v)) for c,v in sorted(cs.items()))+"\n")
PY
cat $OUT/${TAG}_pmc1.txt $OUT/${TAG}_pmc2.txt $OUT/${TAG}_pmc3.txt $OUT/${TAG}_pmc4.txt $OUT/${TAG}_pmc5.txt
