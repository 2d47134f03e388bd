"""rocprofv3 --pmc passes over tools/attn_pmc.py (encoder-shaped attention: 48 images x 6 heads, 1370 tokens, dh 64) -> one JSON:
per-launch SQ counters of cs_attn_kernel<64> and what they mean per wave and key tile (SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES cycles: MI355X_MICROARCH.md constants table).
usage: summarise_attn_pmc.py OUT.json DIR [DIR...]"""
import collections, csv, glob, json, sys
out, dirs = sys.argv[1], sys.argv[2:]
agg = collections.defaultdict(list)
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "cs_attn_kernel" in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in agg.items()}
B, H, L, dh = 48, 6, 1370, 64
waves = B * H * ((L + 127) // 128) * 4
tiles = (L + 63) // 64
res = {"kernel": "cs_attn_kernel<64>", "shape": {"images": B, "heads": H, "tokens": L, "dh": dh}, "launches_averaged": min(len(v) for v in agg.values()),
       "counters_per_launch": c, "waves": waves, "key_tiles_per_wave": tiles}
if "SQ_WAVE_CYCLES" in c:
    wc = 4.0 * c["SQ_WAVE_CYCLES"]
    res["cycles_per_wave_tile"] = wc / waves / tiles
    for k, name in (("SQ_ACTIVE_INST_VALU", "valu_issue_frac"), ("SQ_ACTIVE_INST_LDS", "lds_issue_frac"), ("SQ_ACTIVE_INST_ANY", "any_issue_frac"),
                    ("SQ_WAIT_INST_ANY", "waiting_on_counter_frac"), ("SQ_WAIT_ANY", "wait_any_frac")):
        if k in c: res[name] = 4.0 * c[k] / wc
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c:
    # MFMA-busy cycles are summed over the 4 SIMDs of each CU, SQ_BUSY_CYCLES per SQ (one per CU... reported per XCD-SE): ratio per SIMD
    res["mfma_busy_cycles_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"] if "SQ_INSTS_MFMA" in c else None
if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c:
    res["valu_insts_per_mfma"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"]
    res["mfma_per_wave_tile"] = c["SQ_INSTS_MFMA"] / waves / tiles
    res["valu_per_wave_tile"] = c["SQ_INSTS_VALU"] / waves / tiles
if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_WAVE_CYCLES" in c:
    # 3 waves share a SIMD: SIMD time = wave-cycles / 3 when fully occupied
    res["mfma_pipe_busy_frac_at_3_waves_per_simd"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_WAVE_CYCLES"] / 3.0)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
