cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04i
python tools/time_conv.py 12500000 2 3 > gpurun_out/r04i/time_conv.txt 2>&1
cat gpurun_out/r04i/time_conv.txt
for w in 2 3; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/r04i/pmc_${w}_$c -- python3 tools/time_conv.py 4000000 $w > /dev/null 2> gpurun_out/r04i/pmc_${w}_$c.err
  done
done
python3 - <<PY
import csv, glob, statistics
for w in (2, 3):
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob(f"gpurun_out/r04i/pmc_{w}_{c}/*/*counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if "quad_n3c8" in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
        tot[c] = statistics.median(vals) if vals else None
    if tot["FETCH_SIZE"] is not None:
        b = 2 * tot["FETCH_SIZE"] * 1024 + tot["WRITE_SIZE"] * 1024
        print(f"waves {w}: FETCH {tot['FETCH_SIZE']:.0f} KiB WRITE {tot['WRITE_SIZE']:.0f} KiB -> {b/1e6:.1f} MB per 4e6 parcels = {b/(4e6*144):.2f} x algorithmic")
PY
