#!/usr/bin/env python3
"""Condenses gpurun_out/<tag>_* (tools/profile_round.sh) into profiles/<tag>_*: kernel stats CSV, per-kernel PMC
table (median per dispatch), the HBM traffic JSON that bench.py reports as roofline.traffic, and the bench line."""
import csv, glob, json, os, shutil, statistics, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
os.makedirs(dst, exist_ok=True)
def newest(pattern):
    """gpurun merges every call's files into gpurun_out/: keep only the most recent run of each pass"""
    files = glob.glob(pattern)
    return [max(files, key=os.path.getmtime)] if files else []


for f in newest(os.path.join(src, f"{tag}_stats", "*", "*kernel_stats.csv")):
    shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats.csv"))
for name in ("bench.json", "bench_variants.json", "hbm_ceiling.txt"):
    f = os.path.join(src, f"{tag}_{name}")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_{name}"))
table = {}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    for f in newest(os.path.join(src, f"{tag}_{d}", "*", "*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "cloudy" not in k:
                continue
            short = k.split("cloudy::")[1].split("(")[0] if "cloudy::" in k else k.split("(")[0]  # cloudy_jit_* are extern "C"
            table.setdefault((short, int(r["Grid_Size"])), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
rows = []
for (k, grid), cs in sorted(table.items()):
    row = {"kernel": k, "grid_size": grid}
    row.update({c: statistics.median(v) for c, v in cs.items()})
    rows.append(row)
json.dump(rows, open(os.path.join(dst, f"{tag}_pmc_per_dispatch_median.json"), "w"), indent=1)
traffic = {}
for row in rows:
    if "FETCH_SIZE" in row and "WRITE_SIZE" in row:
        # gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> x2 for coalesced streaming reads (MI355X_MICROARCH.md, HBM)
        traffic[f'{row["kernel"]}@{row["grid_size"]}'] = {
            "fetch_size_kib": row["FETCH_SIZE"], "write_size_kib": row["WRITE_SIZE"],
            "hbm_bytes_per_launch": 2 * row["FETCH_SIZE"] * 1024 + row["WRITE_SIZE"] * 1024}
json.dump(traffic, open(os.path.join(dst, f"{tag}_hbm_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))

# what bench.py reads back (profiles/measured_latest.json): HBM bytes per launch of the headline kernel and the fp64
# flop count per parcel of the threshold kernel, both at the bench's 1e7-parcel workloads
latest = {"tag": tag, "n_parcels": 10_000_000}
try:   # the kernel sources these figures were collected on (bench.py quotes them only while the library still holds them)
    sys.path.insert(0, root)
    from __graft_entry__ import load_package

    _L = load_package().lib()
    latest["source_hash_all"] = f"{_L.cloudy_source_hash(0):016x}"
    latest["source_hash_allinf"] = f"{_L.cloudy_source_hash(1):016x}"
except Exception as e:   # noqa: BLE001
    print("summarize_profiles: no source hash:", e)
for row in rows:
    if (row["kernel"].startswith(("cloudy_jit_allinf2_n2p3_f64", "coal_rhs_allinf2_kernel<2, 3, double>"))
            and row["grid_size"] >= 5_000_000 and "FETCH_SIZE" in row and "WRITE_SIZE" in row
            and not ("cfg3a_kernel" in latest and latest["cfg3a_kernel"].startswith("cloudy_jit"))):
        latest["cfg3a_hbm_bytes_per_launch"] = 2 * row["FETCH_SIZE"] * 1024 + row["WRITE_SIZE"] * 1024
        latest["cfg3a_kernel"] = row["kernel"]
    if (row["kernel"].startswith(("cloudy_jit_sorted_n2p3_f64", "coal_rhs_sorted_kernel<2, 3, 1, double"))
            and row["grid_size"] >= 10_000_000 and "SQ_INSTS_VALU_FMA_F64" in row
            and not ("cfg3b_kernel" in latest and latest["cfg3b_kernel"].startswith("cloudy_jit"))):
        util = row["SQ_THREAD_CYCLES_VALU"] / (row["SQ_ACTIVE_INST_VALU"] * 64.0)
        flops = (2 * row["SQ_INSTS_VALU_FMA_F64"] + row["SQ_INSTS_VALU_MUL_F64"] + row["SQ_INSTS_VALU_ADD_F64"]) * 64.0 * util
        latest["cfg3b_fp64_flops_per_launch"] = flops
        latest["cfg3b_valu_lane_utilisation"] = util
        latest["cfg3b_valu_insts_per_parcel"] = row["SQ_INSTS_VALU"] * 64.0 / latest["n_parcels"]
        latest["cfg3b_kernel"] = row["kernel"]
# fp64-VALU figures of the compute-bound variants of bench.py: useful flops / VALU instructions per item (parcel, or cell
# for the column integrator; per CALL for the fused integrators) and the active-lane fraction
VARIANTS = {   # bench variant -> (kernel name prefix, items per launch in the bench)
    "cfg3b": ("cloudy_jit_sorted_n2p3_f64", 10_000_000),
    "cfg4": ("cloudy_jit_sorted_n3p5_f64", 12_500_000),
    "moving4": ("cloudy_jit_sorted_n4p2_f64", 2_500_000),
    "cfg4q": ("cloudy_jit_quad_n3q10_hydro_f64", 12_500_000),
    "cfg3a_fused_ssprk33": ("cloudy_jit_ssprk33_n2p3_f64", 10_000_000),
    "rainshaft_ssprk33_columns": (("cloudy_jit_rainshaft_ssprk33_n2p3_f64_b512", "cloudy_jit_rainshaft_ssprk33_n2p3_f64",
                                   "rainshaft_ssprk33_kernel<2, 3, 1, double>"), 10_000_000),
    "cfg4q_converged": ("cloudy_jit_quad_n3c8_hydro_f64", 12_500_000),
    "cfg4q_converged_long": ("cloudy_jit_quad_n3c8_long_f64", 12_500_000),
    "numerical_lognorm_example": ("cloudy_jit_quad_n2c8_linear_f64", 2_000_000 // 2),   # (round 6: two parcels per lane, jit_conv_rounds)
    "cfg0_tsit5": ("cloudy_jit_tsit5_n1p2_f64", 10_000_000),
    "cfg3b_f32_fast": ("cloudy_jit_sorted_n2p3_f32fast", 10_000_000),
    "cfg5_f32_planes": ("cloudy_jit_sorted_rs_n2p3_f32", 12_500_000),
    "cfg5_f32_fast": ("cloudy_jit_sorted_rs_n2p3_f32fast", 12_500_000),
    "cfg2": ("cloudy_jit_allinf2_n1p2_f64", 1_000_000 // 2),
    "cfg3a_f32_planes": ("cloudy_jit_allinf2_n2p3_f32", 10_000_000 // 2),
    "cfg3a_f32_fast_packed": ("cloudy_jit_allinf4_n2p3_f32fast", 10_000_000 // 4),
    "cfg3a_aot_kernels": ("coal_rhs_allinf2_kernel<2, 3, double>", 10_000_000 // 2),
    "rainshaft_rhs": (("cloudy_jit_rainshaft_rhs_n2p3_f64_b512", "cloudy_jit_rainshaft_rhs_n2p3_f64"), 10_000_000),
    "cfg3b_f64_relaxed": ("cloudy_jit_sorted_n2p3_f64r", 10_000_000),
    "cfg4_f64_relaxed": ("cloudy_jit_sorted_n3p5_f64r", 12_500_000),
}
kern = {}
for name, (prefix, items) in VARIANTS.items():
    prefixes = (prefix,) if isinstance(prefix, str) else prefix
    cands = [r for r in rows if r["kernel"] in prefixes and "SQ_INSTS_VALU_FMA_F64" in r and r["grid_size"] >= 0.9 * items]
    if not cands:
        continue
    r = max(cands, key=lambda x: x["grid_size"])
    util = r["SQ_THREAD_CYCLES_VALU"] / (r["SQ_ACTIVE_INST_VALU"] * 64.0)
    flops = (2 * r["SQ_INSTS_VALU_FMA_F64"] + r["SQ_INSTS_VALU_MUL_F64"] + r["SQ_INSTS_VALU_ADD_F64"]) * 64.0 * util
    n_items = items * 2 if name in ("cfg2", "cfg3a_f32_planes", "cfg3a_aot_kernels", "numerical_lognorm_example") else items   # two parcels per lane
    n_items = items * 4 if name == "cfg3a_f32_fast_packed" else n_items                            # four parcels per lane
    kern[name] = {"kernel": r["kernel"], "grid_size": r["grid_size"], "fp64_flops_per_item": flops / n_items,
                  "valu_insts_per_item": r["SQ_INSTS_VALU"] * 64.0 / n_items, "lane_utilisation": util}
    t = traffic.get(f'{r["kernel"]}@{r["grid_size"]}')
    if t:
        kern[name]["hbm_bytes_per_launch"] = t["hbm_bytes_per_launch"]
latest["kernels"] = kern
json.dump(latest, open(os.path.join(dst, "measured_latest.json"), "w"), indent=1)
print(json.dumps(latest, indent=1))

# ---- the per-kernel table of profiles/README.md, written from the files above (no hand-typed numbers) --------------------
def kernel_table(tag):
    """profiles/<tag>_kernel_table.md: one row per cloudy kernel of the stats pass -- calls, average / min / max duration
    (rocprofv3 --kernel-trace --stats), and from the PMC medians of the launch with the largest grid: VALU instructions,
    active-lane fraction, VALU issue fraction (instructions x 4 cycles / (1024 SIMDs x 2.4 GHz x average duration)), useful
    fp64 flops as a fraction of the 78.6 TF vector peak, HBM bytes (2 x FETCH_SIZE + WRITE_SIZE) and achieved TB/s."""
    stats_path = os.path.join(dst, f"{tag}_kernel_stats.csv")
    if not os.path.exists(stats_path):
        return
    pmc = {}
    for row in rows:
        if row["kernel"] not in pmc or row["grid_size"] > pmc[row["kernel"]]["grid_size"]:
            pmc[row["kernel"]] = row
    lines = [f"<!-- generated by tools/summarize_profiles.py {tag} from {tag}_kernel_stats.csv and {tag}_pmc_per_dispatch_median.json -->",
             "| kernel | calls | avg µs | min µs | max µs | grid (PMC launch) | VALU insts / lane | active lanes | VALU issue | fp64 peak frac | HBM MB | TB/s |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for r in csv.DictReader(open(stats_path)):
        name = r["Name"]
        if "cloudy" not in name:
            continue
        short = name.split("cloudy::")[1].split("(")[0] if "cloudy::" in name else name.split("(")[0]
        avg = float(r["AverageNs"]) * 1e-3
        cells = [f"`{short}`", r["Calls"], f"{avg:.2f}", f"{float(r['MinNs']) * 1e-3:.2f}", f"{float(r['MaxNs']) * 1e-3:.2f}"]
        p = pmc.get(short)
        if p and "SQ_INSTS_VALU" in p and p.get("SQ_ACTIVE_INST_VALU"):
            util = p["SQ_THREAD_CYCLES_VALU"] / (p["SQ_ACTIVE_INST_VALU"] * 64.0)
            flops = (2 * p["SQ_INSTS_VALU_FMA_F64"] + p["SQ_INSTS_VALU_MUL_F64"] + p["SQ_INSTS_VALU_ADD_F64"]) * 64.0 * util
            issue = p["SQ_INSTS_VALU"] * 4.0 / (1024 * 2.4e9 * avg * 1e-6)
            # (the 4-cycle issue model holds for the fp64 kernels only: single-precision kernels measure > 1 with it)
            issue_s = "n/a (fp32)" if short.endswith(("_f32fast", "_f32")) or issue > 1.0 else f"{issue:.2f}"
            cells += [str(p["grid_size"]), f"{p['SQ_INSTS_VALU'] * 64.0 / p['grid_size']:.0f}", f"{util:.2f}", issue_s,
                      f"{flops / (avg * 1e-6) / 78.6e12:.2f}"]
        else:
            cells += [str(p["grid_size"]) if p else "", "", "", "", ""]
        if p and "FETCH_SIZE" in p and "WRITE_SIZE" in p:
            b = 2 * p["FETCH_SIZE"] * 1024 + p["WRITE_SIZE"] * 1024
            cells += [f"{b / 1e6:.1f}", f"{b / (avg * 1e-6) / 1e12:.2f}"]
        else:
            cells += ["", ""]
        lines.append("| " + " | ".join(cells) + " |")
    with open(os.path.join(dst, f"{tag}_kernel_table.md"), "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


kernel_table(tag)
