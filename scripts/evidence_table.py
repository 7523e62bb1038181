"""Joins the passes of scripts/clean_evidence.sh into one per-kernel table:  python scripts/evidence_table.py <dir> [steps]

  avg us          rocprofv3 --kernel-trace --stats, weight gradients on the main stream (one kernel at a time)
  HBM MB/launch   FETCH_SIZE / WRITE_SIZE counters (scripts/pmc_traffic.py), GB/s = that over the duration
  alg GFLOP       algorithmic flops per launch of the convolution variant (library's own records, scripts/conv_records.py)
  achieved TF     alg GFLOP / avg us
  MFMA busy %     SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs), same process, eager launches
  issued TF       busy cycles * 64 flop / duration of the counter pass: what the matrix unit executed including the padding
                  rows / columns of partial tiles, so issued >= achieved, the gap is padding."""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

out = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def short(name):
    name = re.sub(r"^void\s+", "", name).replace("ssmi::", "")
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    name = name[:cut].strip()
    return re.sub(r"<(float|__bf16), ", "<", name)      # the records name variants without the storage type


dur = {}
for r in csv.DictReader(open(out + "/kernel_stats.csv")):
    dur[short(r["Name"])] = (int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6 / steps)
traffic = {short(k): v for k, v in json.load(open(out + "/traffic.json")).items()}
recs = json.load(open(out + "/conv_records.json"))
pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/pass/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = pmc[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
print("| kernel | launches/step | avg us | ms/step | HBM MB/launch | GB/s | alg GFLOP/launch | achieved TF | MFMA busy % | issued TF |")
print("|---|---|---|---|---|---|---|---|---|---|")
tot = 0.0
for k, (n, us, ms) in sorted(dur.items(), key=lambda kv: -kv[1][2]):
    tot += ms
    if ms < 0.02:
        continue
    t = traffic.get(k)
    mb = t["hbm_bytes_per_launch"] / 1e6 if t else None
    rec = recs.get(k)
    if rec is None:      # trace names of the forward kernel carry the normalise-on-load flag, the library's records a suffix
        m = re.match(r"^(conv_fwd_kernel<\d+, (?:true|false)), (true|false)>$", k)
        if m:
            rec = recs.get(m.group(1) + ">" + (" prebn" if m.group(2) == "true" else ""))
    gf = rec["flops_per_step"] / max(rec["launches_per_step"], 1) / 1e9 if rec else None
    c = pmc.get(k)
    busy = issued = None
    if c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["SQ_VALU_MFMA_BUSY_CYCLES"][0] > 0:
        cyc = c["GRBM_GUI_ACTIVE"][0] / c["GRBM_GUI_ACTIVE"][1] / 8.0
        b = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / c["SQ_VALU_MFMA_BUSY_CYCLES"][1]
        busy = 100.0 * b / (cyc * 1024.0)
        issued = b * 64.0 / (us * 1e-6) / 1e12
    f = lambda v, p: "" if v is None else p % v
    print("| `%s` | %.1f | %.1f | %.3f | %s | %s | %s | %s | %s | %s |" % (
        k, n, us, ms, f(mb, "%.1f"), f(None if mb is None else mb / us * 1e3, "%.0f"), f(gf, "%.3f"),
        f(None if gf is None else gf / us * 1e3, "%.1f"), f(busy, "%.1f"), f(issued, "%.1f")))
print()
print("summed kernel time %.2f ms/step" % tot)
