#!/usr/bin/env python3
"""Turn gpurun_out/prof_* and gpurun_out/pmc into the committed summaries under profiles/ (round tag argv[1])."""
import collections, csv, glob, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs("profiles", exist_ok=True)
for name in ("bench", "cifar"):
    for p in glob.glob(f"gpurun_out/prof_{name}/*/*kernel_stats.csv"):
        rows = list(csv.reader(open(p)))
        keep = [rows[0]] + [r for r in rows[1:] if "card" in r[0] or "wide_" in r[0] or "pack" in r[0]]
        csv.writer(open(f"profiles/{tag}_kernel_stats_{name}.csv", "w")).writerows(keep)
    log = f"gpurun_out/prof_{name}.log"
    if os.path.exists(log):
        js = [l for l in open(log) if l.startswith("{")]
        open(f"profiles/{tag}_{name}_under_rocprof.json", "w").writelines(js)
out = {}
for p in sorted(glob.glob("gpurun_out/pmc/*/vm/*counter_collection.csv")):
    agg, dur = collections.defaultdict(float), None
    for r in csv.DictReader(open(p)):
        if "card_sample" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"])
            dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    out[p.split("/")[2]] = dict(agg, kernel_seconds=dur)
s1, s2 = out["sq1"], out["sq2"]
simd_cycles = s1["GRBM_GUI_ACTIVE"] / 8
chains, T = 1048576, 1000
d = {
    "workload": "one launch: 1024 rows x 1024 z = 1 048 576 chains, T=1000, x_dim=1, in-kernel Philox, last-only output",
    "command": "bash tools/profile_pmc.sh gpurun_out/pmc (4 separate rocprofv3 --pmc passes of: python3 bench.py --rows 1024 --nz 1024 --steps 1 --warmup 0 --no-cpu-baseline)",
    "kernel_seconds_under_pmc": s1["kernel_seconds"],
    "effective_clock_GHz": simd_cycles / s1["kernel_seconds"] / 1e9,
    "mfma_pipe_utilisation": s1["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / simd_cycles,
    "mfma_per_wave_step": s1["SQ_INSTS_MFMA"] / (chains / 32 * T),
    "valu_incl_mfma_per_wave_step": s1["SQ_INSTS_VALU"] / (chains / 32 * T),
    "trans_per_wave_step": s2["SQ_INSTS_VALU_TRANS_F32"] / (chains / 32 * T),
    "lds_insts_per_wave_step": s2["SQ_INSTS_LDS"] / (chains / 32 * T),
    "lds_bank_conflict_cycles": s2["SQ_LDS_BANK_CONFLICT"],
    "FETCH_SIZE_KiB_raw": out["tcc_fetch"]["FETCH_SIZE"], "WRITE_SIZE_KiB": out["tcc_write"]["WRITE_SIZE"],
    "hbm_bytes_per_launch": (2 * out["tcc_fetch"]["FETCH_SIZE"] + out["tcc_write"]["WRITE_SIZE"]) * 1024,
    "hbm_note": "FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; calibrated for 16-B/lane streams, so an upper bound here). n_z >= 32 launches keep the hoisted layer-1 columns in LDS: no per-step global traffic; reads are the packed weights (once per CU) + gamma rows (L2 hits) + inputs, writes are the y_0 output (4 B/chain).",
    "l2_hit_rate": out["tcc_write"]["TCC_HIT_sum"] / (out["tcc_write"]["TCC_HIT_sum"] + out["tcc_write"]["TCC_MISS_sum"]),
}
json.dump({"raw": out, "derived": d}, open(f"profiles/{tag}_pmc_c2_launch.json", "w"), indent=1)
json.dump({"x_dim": 1, "T": 1000, "rows_per_launch": chains, "hbm_bytes_per_launch": d["hbm_bytes_per_launch"],
           "source": f"profiles/{tag}_pmc_c2_launch.json"}, open("profiles/hbm_traffic.json", "w"), indent=1)
print(json.dumps(d, indent=1))
