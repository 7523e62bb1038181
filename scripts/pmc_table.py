"""Per-kernel averages of every counter found in a rocprofv3 --pmc csv (scripts/pmc_pass.sh)."""
import csv, glob, re, sys
from collections import defaultdict
out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/pass/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("ssmi::", "")
        k = k.split("(")[0] if "<" not in k.split("(")[0] or k.split("(")[0].count("<") == k.split("(")[0].count(">") else k[:60]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
names = sorted({c for k in acc for c in acc[k]})
print("%-44s %6s " % ("kernel", "n") + " ".join("%14s" % c[-14:] for c in names))
order = sorted(acc, key=lambda k: -acc[k].get("SQ_BUSY_CYCLES", acc[k][names[0]])[0])
for k in order[:32]:
    n = max(v[1] for v in acc[k].values())
    print("%-44s %6d " % (k[:44], n) + " ".join("%14.0f" % (acc[k][c][0] / max(acc[k][c][1], 1)) for c in names))
