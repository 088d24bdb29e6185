#!/bin/bash
# VALU instruction count per lane of the rhs kernel for one JIT_DEFS setting
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/$TAG -- python3 tools/time_rainshaft_unfused.py > /dev/null 2> gpurun_out/$TAG.err
python3 - <<PY
import csv,glob,statistics,collections
t=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/$TAG/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "rainshaft_rhs" in r["Kernel_Name"]:
            t[r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for g,c in t.items():
    m={k:statistics.median(v) for k,v in c.items()}
    print("$TAG", "DEFS=", "$CLOUDY_HIP_JIT_DEFS", "VALU insts/lane", m["SQ_INSTS_VALU"]*64/float(g), "lanes", m["SQ_THREAD_CYCLES_VALU"]/(m["SQ_ACTIVE_INST_VALU"]*64))
PY
