"""MFMA utilisation per kernel from one rocprofv3 --pmc pass (scripts/pmc_pass.sh <tag> "SQ_VALU_MFMA_BUSY_CYCLES
SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"):  python scripts/mfma_table.py gpurun_out/pmc_<tag>

SQ_VALU_MFMA_BUSY_CYCLES is summed over the 1024 SIMDs of the chip (64 cycles per v_mfma_f32_32x32x2_f32, 32 per
16x16x4: 64 FLOP per busy cycle), GRBM_GUI_ACTIVE over the 8 XCDs, so
    utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)    and    MFMA TFLOP/s issued = MFMA_BUSY * 64 / duration.
Padding rows / columns of partial tiles count as busy: the figure is an upper bound of the useful share."""
import csv
import glob
import re
import sys
from collections import defaultdict

out = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/pass/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void\s+", "", r["Kernel_Name"]).replace("ssmi::", "")
        depth, name = 0, []
        for ch in k:
            if ch == "<":
                depth += 1
            elif ch == ">":
                depth -= 1
            elif ch == "(" and depth == 0:
                break
            name.append(ch)
        a = acc["".join(name).strip()][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
print("| kernel | launches | avg cycles (GUI_ACTIVE / 8) | MFMA busy cycles / launch | MFMA utilisation |")
print("|---|---|---|---|---|")
rows = []
for k, c in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    n = c["GRBM_GUI_ACTIVE"][1]
    cyc = c["GRBM_GUI_ACTIVE"][0] / n / 8.0
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"][0] / c["SQ_VALU_MFMA_BUSY_CYCLES"][1]
    if busy <= 0:
        continue
    rows.append((c["GRBM_GUI_ACTIVE"][0], k, n, cyc, busy, busy / (cyc * 1024.0)))
for _, k, n, cyc, busy, u in sorted(rows, reverse=True):
    print("| `{}` | {} | {:.0f} | {:.3e} | {:.1f} % |".format(k, n, cyc, busy, 100 * u))
