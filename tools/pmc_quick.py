import collections, csv, glob, sys
agg = collections.defaultdict(list)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "panel" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in agg.items()}
for k in sorted(c): print(f"{k:32s} {c[k]:.4g}")
if "SQ_WAVE_CYCLES" in c:
    wc = c["SQ_WAVE_CYCLES"]
    for k in ("SQ_ACTIVE_INST_ANY","SQ_ACTIVE_INST_VALU","SQ_ACTIVE_INST_LDS","SQ_WAIT_INST_ANY","SQ_WAIT_ANY","SQ_WAIT_INST_LDS","SQ_BUSY_CYCLES"):
        if k in c: print(f"  {k}/WAVE_CYCLES = {c[k]/wc:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c: print(f"  MFMA busy / (4*wave_cycles) = {c['SQ_VALU_MFMA_BUSY_CYCLES']/(4*wc):.3f}")
