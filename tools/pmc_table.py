"""tools/pmc_table.py <dir> — per-build table of the counters tools/pmc_compare.sh collected (render kernel only, mean per launch)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
tab = collections.defaultdict(dict)
for f in sorted(glob.glob(os.path.join(d, "*", "**", "*counter_collection.csv"), recursive=True)):
    v = os.path.relpath(f, d).split(os.sep)[0].rsplit("_", 1)[0]
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "render_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, vals in acc.items():
        tab[k][v] = sum(vals) / len(vals)
vs = sorted({v for k in tab for v in tab[k]})
print("%-28s" % "counter" + "".join("%18s" % v for v in vs) + ("%10s" % "ratio" if len(vs) == 2 else ""))
for k in sorted(tab):
    row = [tab[k].get(v, float("nan")) for v in vs]
    print("%-28s" % k + "".join("%18.4g" % x for x in row) + ("%10.3f" % (row[1] / row[0]) if len(vs) == 2 and row[0] else ""))
