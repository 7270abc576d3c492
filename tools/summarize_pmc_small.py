#!/usr/bin/env python3
"""Sum the SQ counters of tools/profile_pmc_small.sh per kernel: usage  summarize_pmc_small.py <outdir> <label>"""
import collections, csv, glob, json, re, sys
out, label = sys.argv[1], sys.argv[2]
res = collections.defaultdict(lambda: collections.defaultdict(float))
disp = collections.Counter()
import os
newest = {}
for p in glob.glob(f"{out}/*/*/*counter_collection.csv"):      # one file per pass: the newest
    k = p.split("/")[-3]
    if k not in newest or os.path.getmtime(p) > os.path.getmtime(newest[k]):
        newest[k] = p
for p in sorted(newest.values()):
    seen = set()
    for r in csv.DictReader(open(p)):
        mm = re.search(r"card_sample\w+", r["Kernel_Name"])
        if not mm:
            continue
        k = mm.group(0)
        res[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("GRBM_GUI_ACTIVE", "SQ_INSTS_LDS") and r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                disp[k] += 1
for k, c in res.items():
    simd_cycles = c["GRBM_GUI_ACTIVE"] / 8
    d = {"label": label, "kernel": k, "dispatches": disp[k],
         "mfma_pipe_busy_of_simd_cycles": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / simd_cycles, 4) if simd_cycles else None,
         "insts_mfma": c["SQ_INSTS_MFMA"], "insts_valu_incl_mfma": c["SQ_INSTS_VALU"], "insts_trans": c["SQ_INSTS_VALU_TRANS_F32"],
         "insts_lds": c["SQ_INSTS_LDS"], "insts_salu": c["SQ_INSTS_SALU"], "lds_bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"],
         "lds_idx_active_cycles": c["SQ_LDS_IDX_ACTIVE"], "wait_inst_lds": c["SQ_WAIT_INST_LDS"], "wave_cycles": c["SQ_WAVE_CYCLES"],
         "wait_inst_any": c["SQ_WAIT_INST_ANY"], "active_inst_any": c["SQ_ACTIVE_INST_ANY"]}
    print(json.dumps(d))
