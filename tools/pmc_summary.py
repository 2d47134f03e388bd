import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            k = r["Kernel_Name"].replace("void (anonymous namespace)::", "").split("(")[0]
            if "cs_" not in k: continue
            acc[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sorted(acc.items()):
    print(k)
    for n, v in sorted(c.items()):
        print(f"   {n:32s} {sum(v)/len(v):16.1f}  (n={len(v)})")
