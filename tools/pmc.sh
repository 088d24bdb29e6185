#!/bin/bash
# VALU / wait / LDS / scratch counters of one python command on the GPU box (separate passes, kernel trace only):
#   tools/pmc.sh <tag> <script.py args...>  ->  gpurun_out/<tag>_pmc.txt
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out
mkdir -p $OUT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/${TAG}_p1 -- python3 "$@" > /dev/null 2> $OUT/${TAG}_p1.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/${TAG}_p2 -- python3 "$@" > /dev/null 2> $OUT/${TAG}_p2.err
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_FLAT --kernel-trace --output-format csv -d $OUT/${TAG}_p3 -- python3 "$@" > /dev/null 2> $OUT/${TAG}_p3.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_p4 -- python3 "$@" > /dev/null 2> $OUT/${TAG}_p4.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/${TAG}_p5 -- python3 "$@" > /dev/null 2> $OUT/${TAG}_p5.err
python3 - <<PY
import csv, glob, statistics, collections
t=collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("p1","p2","p3","p4","p5"):
    for f in glob.glob("$OUT/${TAG}_%s/*/*counter_collection.csv"%d):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][-60:]
            t[(k,r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$OUT/${TAG}_pmc.txt","w") as o:
    for k,cs in sorted(t.items()):
        c={n:statistics.median(v) for n,v in cs.items()}
        o.write("%s grid=%s\n  "%k + ", ".join("%s=%.4g"%(n,v) for n,v in sorted(c.items()))+"\n")
        try:
            g=float(k[1]); cyc=c["GRBM_GUI_ACTIVE"]/8.0
            o.write("  per lane: VALU insts %.0f, lanes %.3f, VALU busy (ACTIVE_INST_VALU x4 / SIMD cycles) %.3f, cycles %.4g (%.3f ms at 2.4 GHz), LDS insts/lane %.1f, VMEM insts/lane %.1f, HBM MB %.1f\n" % (
                c["SQ_INSTS_VALU"]*64/g, c["SQ_THREAD_CYCLES_VALU"]/(c["SQ_ACTIVE_INST_VALU"]*64), c["SQ_ACTIVE_INST_VALU"]*4/(1024*cyc), cyc, cyc/2.4e6,
                c.get("SQ_INSTS_LDS",0)*64/g, c.get("SQ_INSTS_VMEM",0)*64/g, (2*c.get("FETCH_SIZE",0)+c.get("WRITE_SIZE",0))*1024/1e6))
        except Exception as e:
            o.write("  (summary failed: %s)\n"%e)
PY
cat $OUT/${TAG}_pmc.txt
