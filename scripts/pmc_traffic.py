"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; see scripts/pmc_traffic.sh).

Both counters are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes,
i.e. reports half of the bytes such a stream fetches, so the read side is doubled; WRITE_SIZE is exact for 16-byte
per-lane streaming stores.  Output: {kernel: {launches, fetch_kib_avg, write_kib_avg, hbm_bytes_per_launch}}."""
import csv
import glob
import json
import re
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short  # noqa: E402

out = sys.argv[1]




def collect(sub, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in glob.glob("{}/{}/**/*counter_collection.csv".format(out, sub), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            a = acc[short(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


fetch, write = collect("fetch", "FETCH_SIZE"), collect("write", "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, [0.0, 0])
    w, nw = write.get(k, [0.0, 0])
    fa, wa = (f / nf if nf else 0.0), (w / nw if nw else 0.0)
    res[k] = {"launches": max(nf, nw), "fetch_kib_avg": round(fa, 1), "write_kib_avg": round(wa, 1),
              "hbm_bytes_per_launch": round((2.0 * fa + wa) * 1024)}
json.dump(res, open(out + "/traffic.json", "w"), indent=1, sort_keys=True)
top = sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:25]
for k, v in top:
    print("%-60s n %5d  fetch %10.1f KiB  write %10.1f KiB  hbm %8.2f MB/launch" % (
        k[:60], v["launches"], v["fetch_kib_avg"], v["write_kib_avg"], v["hbm_bytes_per_launch"] / 1e6))
