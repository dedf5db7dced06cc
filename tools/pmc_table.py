"""Per-kernel sums of every counter found in rocprofv3 --pmc csv outputs under the given dirs."""
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        seen = set()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (f, r["Dispatch_Id"])
            if key not in seen and d == sys.argv[1]:
                seen.add(key); calls[k] += 1
names = sorted({c for v in agg.values() for c in v})
for k, v in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_INSTS_VALU", 0))[:14]:
    print(k[:48], "calls", calls[k])
    for c in names:
        if c in v:
            print("    %-28s %.4g" % (c, v[c]))
