"""Joins the passes of scripts/clean_evidence.sh into one per-kernel table:  python scripts/evidence_table.py <dir> [steps]

  avg us          rocprofv3 --kernel-trace --stats, weight gradients on the main stream (one kernel at a time)
  HBM MB/launch   FETCH_SIZE / WRITE_SIZE counters (scripts/pmc_traffic.py), GB/s = that over the duration
  alg GFLOP       algorithmic flops per launch of the convolution variant (library's own records, scripts/conv_records.py)
  achieved TF     alg GFLOP / avg us
  issued TF       SQ_VALU_MFMA_BUSY_CYCLES (summed over the SIMDs) * flop per busy cycle / the TRACE pass's duration: what the
                  matrix unit executed including the padding rows / columns of partial tiles, so issued >= achieved, the gap
                  is padding.  Flop per busy SIMD cycle: 64 for the float32 matrix instructions, 1024 for the bfloat16 ones.
  MFMA % of peak  issued TF / dense peak of the instructions the kernel runs on (157.3 TFLOP/s float32, 2500 bfloat16):
                  numerator and denominator on the same clock (the trace's), unlike a ratio against the counter pass's own
                  cycle count, which runs at a different clock under the profiler."""
F32_PEAK, BF16_PEAK = 157.3, 2500.0
import csv
import glob
import json
import re
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short, storage_type  # noqa: E402

out = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12




dur = {}
for r in csv.DictReader(open(out + "/kernel_stats.csv")):
    a = dur.setdefault(short(r["Name"]), [0, 0.0])
    a[0] += int(r["Calls"])
    a[1] += float(r["TotalDurationNs"])
dur = {k: (c / steps, ns / c / 1e3, ns / 1e6 / steps) for k, (c, ns) in dur.items()}
traffic = {short(k): v for k, v in json.load(open(out + "/traffic.json")).items()}


def uses_bf16_mfma(k, recs):
    """bfloat16 matrix instructions: the bfloat16 weight-gradient kernels by name, the direct kernels when the library's
    records of this run list the variant with its ' bf16 mfma' suffix."""
    if "_bf16<" in k or k.endswith("_bf16"):      # bfloat16-only kernels (tiled / patch convolutions, weight gradients)
        return "pack" not in k
    if storage_type(k) != "bf16":
        return False
    base = short(k, strip_type=True)
    m = re.match(r"^(conv_fwd_kernel<\d+, (?:true|false))(?:, (?:true|false|\d+))?(?:, \d+)?>$", base)
    return bool(m) and any(r.startswith(m.group(1) + ">") and r.endswith("bf16 mfma") for r in recs)

recs = json.load(open(out + "/conv_records.json"))
pmc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/pass/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        a = pmc[short(r["Kernel_Name"])][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
print("| kernel | launches/step | avg us | ms/step | HBM MB/launch | GB/s | alg GFLOP/launch | achieved TF | issued TF | MFMA % of peak |")
print("|---|---|---|---|---|---|---|---|---|---|")
tot = 0.0
for k, (n, us, ms) in sorted(dur.items(), key=lambda kv: -kv[1][2]):
    tot += ms
    if ms < 0.02:
        continue
    t = traffic.get(k)
    mb = t["hbm_bytes_per_launch"] / 1e6 if t else None
    kb = short(k, strip_type=True)      # the library's records name variants without the storage type
    bf = uses_bf16_mfma(k, recs)
    rec = recs.get(kb) or recs.get(kb + " bf16 mfma")
    if rec is None:      # trace names of the forward kernel carry the normalise-on-load flag, the library's records a suffix
        m = re.match(r"^(conv_fwd_kernel<\d+, (?:true|false))(?:, (true|false|\d+))?(?:, \d+)?>$", kb)
        if m:
            key = m.group(1) + ">" + {"true": " prebn", "1": " prebn", "2": " premerge"}.get(m.group(2) or "", "")
            rec = recs.get(key + " bf16 mfma" if bf else key) or recs.get(key)
    gf = rec["flops_per_step"] / max(rec["launches_per_step"], 1) / 1e9 if rec else None
    c = pmc.get(k)
    busy = issued = None
    if c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["SQ_VALU_MFMA_BUSY_CYCLES"][0] > 0:
        cyc = c["GRBM_GUI_ACTIVE"][0] / c["GRBM_GUI_ACTIVE"][1] / 8.0
        b = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / c["SQ_VALU_MFMA_BUSY_CYCLES"][1]
        issued = b * (1024.0 if bf else 64.0) / (us * 1e-6) / 1e12
        busy = 100.0 * issued / (BF16_PEAK if bf else F32_PEAK)
    f = lambda v, p: "" if v is None else p % v
    print("| `%s` | %.1f | %.1f | %.3f | %s | %s | %s | %s | %s | %s |" % (
        k, n, us, ms, f(mb, "%.1f"), f(None if mb is None else mb / us * 1e3, "%.0f"), f(gf, "%.3f"),
        f(None if gf is None else gf / us * 1e3, "%.1f"), f(issued, "%.1f"), f(busy, "%.1f")))
print()
print("summed kernel time %.2f ms/step" % tot)
