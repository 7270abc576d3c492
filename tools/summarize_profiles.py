#!/usr/bin/env python3
"""Turn gpurun_out/prof_* and gpurun_out/pmc* (tools/profile_all.sh) into the committed summaries under
profiles/ (round tag argv[1])."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
os.makedirs("profiles", exist_ok=True)
if os.path.exists("gpurun_out/ubench.txt"):
    # keep the hand-written notes (everything from the first dashed rule on) when the measurements are refreshed
    dst, note = f"profiles/{tag}_ubench.txt", ""
    if os.path.exists(dst):
        old = open(dst).read()
        cut = old.find("\n" + "-" * 70)
        note = old[cut:] if cut >= 0 else ""
    open(dst, "w").write(open("gpurun_out/ubench.txt").read() + note)

for name in ("bench", "cifar", "stream", "mnist"):
    for p in sorted(glob.glob(f"gpurun_out/prof_{name}/*/*kernel_stats.csv"), key=os.path.getmtime)[-1:]:      # the newest run
        rows = list(csv.reader(open(p)))
        keep = [rows[0]] + [r for r in rows[1:] if "card" in r[0] or "wide_" in r[0] or "pack" in r[0]]
        csv.writer(open(f"profiles/{tag}_kernel_stats_{name}.csv", "w")).writerows(keep)
    log = f"gpurun_out/prof_{name}.log"
    if os.path.exists(log):
        js = [l for l in open(log) if l.startswith("{")]
        open(f"profiles/{tag}_{name}_under_rocprof.json", "w").writelines(js)


def passes(dirname, match):
    """counter sums over the dispatches whose kernel name contains `match`, per PMC pass"""
    out = {}
    # one file per pass: the newest (gpurun_out/ keeps the files of earlier runs of the same script)
    for p in sorted(glob.glob(f"gpurun_out/{dirname}/*/vm/*counter_collection.csv"), key=os.path.getmtime):
        agg, dur, n = collections.defaultdict(float), 0.0, collections.Counter()
        for r in csv.DictReader(open(p)):
            if match in r["Kernel_Name"]:
                agg[r["Counter_Name"]] += float(r["Counter_Value"])
                n[r["Counter_Name"]] += 1
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        out[p.split("/")[2]] = dict(agg, kernel_seconds=dur, dispatches=max(n.values()) if n else 0)
    return out


HBM_NOTE = ("FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B; calibrated for 16-B/lane "
            "streams, so an upper bound for narrower loads).")
traffic = []


def persistent(dirname, match, chains, tile, T, workload, command, outname, x_dim, hidden):
    out = passes(dirname, match)
    if "sq1" not in out or not out["sq1"].get("GRBM_GUI_ACTIVE"):
        return
    s1, s2 = out["sq1"], out["sq2"]
    simd_cycles = s1["GRBM_GUI_ACTIVE"] / 8          # summed over the 8 XCDs
    mfma_cycles = 64 if tile == 32 else 32           # v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 issue cycles
    ws = chains / tile * T                           # wave-steps
    d = {
        "workload": workload, "command": command,
        "kernel_seconds_under_pmc": s1["kernel_seconds"],
        "effective_clock_GHz": simd_cycles / s1["kernel_seconds"] / 1e9,
        "mfma_pipe_utilisation": s1["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / simd_cycles,
        "mfma_per_wave_step": s1["SQ_INSTS_MFMA"] / ws,
        "mfma_issue_cycles_each": mfma_cycles,
        "valu_incl_mfma_per_wave_step": s1["SQ_INSTS_VALU"] / ws,
        "trans_per_wave_step": s2["SQ_INSTS_VALU_TRANS_F32"] / ws,
        "lds_insts_per_wave_step": s2["SQ_INSTS_LDS"] / ws,
        "lds_bank_conflict_cycles": s2["SQ_LDS_BANK_CONFLICT"],
        "FETCH_SIZE_KiB_raw": out["tcc_fetch"]["FETCH_SIZE"], "WRITE_SIZE_KiB": out["tcc_write"]["WRITE_SIZE"],
        "hbm_bytes_per_launch": (2 * out["tcc_fetch"]["FETCH_SIZE"] + out["tcc_write"]["WRITE_SIZE"]) * 1024,
        "hbm_note": HBM_NOTE,
        "l2_hit_rate": out["tcc_write"]["TCC_HIT_sum"] / (out["tcc_write"]["TCC_HIT_sum"] + out["tcc_write"]["TCC_MISS_sum"]),
    }
    json.dump({"raw": out, "derived": d}, open(f"profiles/{tag}_{outname}.json", "w"), indent=1)
    traffic.append({"x_dim": x_dim, "hidden": hidden, "T": T, "rows_per_launch": chains,
                    "hbm_bytes_per_launch": d["hbm_bytes_per_launch"], "source": f"profiles/{tag}_{outname}.json"})
    print(outname, json.dumps(d, indent=1))


persistent("pmc", "card_sample_guided", 1048576, 32, 1000,
           "one launch: 1024 rows x 1024 z = 1 048 576 chains, T=1000, x_dim=1, hidden 128, in-kernel Philox, last-only output",
           "bash tools/profile_pmc.sh gpurun_out/pmc (4 separate rocprofv3 --pmc passes of: python3 bench.py --rows 1024 "
           "--nz 1024 --steps 1 --warmup 0 --no-cpu-baseline)", "pmc_c2_launch", 1, 128)
persistent("pmc_stream", "card_sample_stream", 262144, 16, 1000,
           "one launch: 2048 rows x 128 z = 262 144 chains, T=1000, x_dim=13, hidden 256, in-kernel Philox, last-only output",
           "bash tools/profile_pmc.sh gpurun_out/pmc_stream --workload sweep256 --rows 2048 --nz 128 --steps 1 --warmup 0 "
           "--no-cpu-baseline", "pmc_stream_launch", 13, 256)
if traffic:
    json.dump({"entries": traffic}, open("profiles/hbm_traffic.json", "w"), indent=1)

# GEMM back end: the F x F wide_gemm_kernel<EPI_SOFTPLUS> launches of tools/bench_wide.py (CIFAR-10 head shape)
w = passes("pmc_wide", "wide_gemm_kernel<0>")
if "sq1" in w and w["sq1"].get("GRBM_GUI_ACTIVE"):
    s1, n = w["sq1"], w["sq1"]["dispatches"]
    simd_cycles = s1["GRBM_GUI_ACTIVE"] / 8
    d = {
        "workload": "tools/bench_wide.py --T 6 --reps 1: N = 25 000 chains, F = 4096, C = 10; the F x F GEMM launches only",
        "dispatches": n,
        "mfma_pipe_utilisation": s1["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / simd_cycles,
        "valu_non_mfma_per_mfma": (s1["SQ_INSTS_VALU"] - s1["SQ_INSTS_MFMA"]) / s1["SQ_INSTS_MFMA"],
        "lds_bank_conflict_cycles": w["sq2"]["SQ_LDS_BANK_CONFLICT"],
        "l2_hit_rate": w["tcc_write"]["TCC_HIT_sum"] / (w["tcc_write"]["TCC_HIT_sum"] + w["tcc_write"]["TCC_MISS_sum"]),
        "hbm_bytes_per_launch": (2 * w["tcc_fetch"]["FETCH_SIZE"] / w["tcc_fetch"]["dispatches"]
                                 + w["tcc_write"]["WRITE_SIZE"] / w["tcc_write"]["dispatches"]) * 1024,
        "algorithmic_bytes_per_launch": 25000 * 4096 * 4 * 2 + 4096 * 4096 * 4,
        "hbm_note": HBM_NOTE,
    }
    json.dump({"raw": w, "derived": d}, open(f"profiles/{tag}_pmc_wide_gemm.json", "w"), indent=1)
    print("pmc_wide_gemm", json.dumps(d, indent=1))
