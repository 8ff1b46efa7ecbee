import collections, csv, glob, sys
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for fn in files:
    for r in csv.DictReader(open(fn)):
        agg[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    if "acfm" not in k:
        continue
    print(k)
    for c, vals in sorted(v.items()):
        print("   %-26s %16.0f  (n=%d)" % (c, sum(vals) / len(vals), len(vals)))
