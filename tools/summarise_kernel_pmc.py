"""rocprofv3 --pmc passes over tools/panel_pmc.py -> one JSON with, per kernel (cs_panel_kernel, cs_gemm256_kernel), the per-launch SQ counters
and the fractions they give: vector-port issue, LDS issue, MFMA pipe busy, waiting, LDS bank conflicts.  SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles summed over the SIMDs (MI355X_MICROARCH.md constants table); both
kernels keep 2 waves per SIMD (8 waves per CU, one workgroup).  usage: summarise_kernel_pmc.py OUT.json DIR [DIR...]"""
import collections, csv, glob, json, sys
out, dirs = sys.argv[1], sys.argv[2:]
res = {}
for key, flops in (("cs_panel_kernel", 48 * 1370 * (2.0 * 384 * 384 + 4.0 * 384 * 1536)), ("cs_gemm256_kernel", 2.0 * 131520 * 2304 * 768)):
    agg = collections.defaultdict(list)
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if key in r["Kernel_Name"]:
                    agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not agg:
        continue
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    e = {"launches_averaged": min(len(v) for v in agg.values()), "counters_per_launch": c, "waves_per_simd": 2}
    if "SQ_WAVE_CYCLES" in c:
        wc = 4.0 * c["SQ_WAVE_CYCLES"]
        for k, name in (("SQ_ACTIVE_INST_VALU", "valu_issue_frac_of_wave_cycles"), ("SQ_ACTIVE_INST_LDS", "lds_issue_frac_of_wave_cycles"),
                        ("SQ_ACTIVE_INST_ANY", "any_issue_frac_of_wave_cycles"), ("SQ_WAIT_INST_ANY", "waiting_on_counter_frac_of_wave_cycles"),
                        ("SQ_WAIT_ANY", "wait_any_frac_of_wave_cycles"), ("SQ_WAIT_INST_LDS", "waiting_on_lds_frac_of_wave_cycles")):
            if k in c: e[name] = 4.0 * c[k] / wc
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_pipe_busy_frac_at_2_waves_per_simd"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (wc / 2.0)
    if "SQ_INSTS_MFMA" in c:
        e["flop_per_mfma_instruction"] = flops / c["SQ_INSTS_MFMA"]
        for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"):
            if k in c: e[k.lower() + "_per_mfma"] = c[k] / c["SQ_INSTS_MFMA"]
    if "SQ_LDS_BANK_CONFLICT" in c and "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        e["lds_bank_conflict_frac_of_lds_cycles"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
    res[key] = e
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:600])
