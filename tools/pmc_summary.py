"""per-kernel means of the counters collected by tools/pmc_bench.sh (csv passes under <dir>/p*/): kernel name + grid -> counter means"""
import csv, glob, os, sys, collections
root = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        key = (r["Kernel_Name"].split("(")[0][:48], r.get("Grid_Size", ""), )
        a = acc[key][r["Counter_Name"]]
        a[0] += 1; a[1] += float(r["Counter_Value"])
rows = []
for key, ctrs in acc.items():
    n = max(v[0] for v in ctrs.values())
    rows.append((n, key, {c: v[1] / v[0] for c, v in ctrs.items()}))
names = sorted({c for _, _, m in rows for c in m})
print("%-50s %-12s %6s " % ("kernel", "grid", "calls") + " ".join("%16s" % c[:16] for c in names))
for n, key, m in sorted(rows, key=lambda t: -t[2].get("FETCH_SIZE", 0) * t[0])[:45]:
    print("%-50s %-12s %6d " % (key[0], key[1], n) + " ".join("%16.1f" % m.get(c, float("nan")) for c in names))
