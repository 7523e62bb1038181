"""Per-kernel HBM throughput table:  python scripts/hbm_table.py <kernel_stats.csv> <traffic.json> [steps]

Joins the rocprofv3 --kernel-trace --stats averages with the PMC traffic per launch (scripts/pmc_traffic.sh) and prints,
per kernel, launches per step, average duration, HBM bytes per launch and the resulting GB/s against the 8 TB/s peak.
Durations come from the full training step, where the weight-gradient streams run beside the main chain: a kernel's
GB/s is what it got while sharing the memory system, not what it reaches alone."""
import csv
import json
import re
import sys

PEAK_GBS = 8000.0


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("ssmi::", "")
    depth, out = 0, []
    for ch in name:            # cut the argument list: the first '(' outside template brackets
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            break
        out.append(ch)
    return "".join(out).strip()


def main():
    stats, traffic = sys.argv[1], json.load(open(sys.argv[2]))
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    rows = list(csv.DictReader(open(stats)))
    print("| kernel | launches/step | avg µs | HBM MB/launch (PMC) | GB/s | % of 8 TB/s | ms/step |")
    print("|---|---|---|---|---|---|---|")
    tot_bytes = tot_ms = 0.0
    for r in rows:
        k = short(r["Name"])
        t = traffic.get(k)
        if t is None or not k or k.startswith("__amd"):
            continue
        avg_us = float(r["AverageNs"]) / 1e3
        mb = t["hbm_bytes_per_launch"] / 1e6
        gbs = mb / avg_us * 1e3
        calls = int(r["Calls"]) / steps
        ms = float(r["TotalDurationNs"]) / 1e6 / steps
        tot_bytes += mb * calls
        tot_ms += ms
        if ms < 0.02:
            continue
        print("| `{}` | {:.1f} | {:.1f} | {:.1f} | {:.0f} | {:.0f} % | {:.3f} |".format(k, calls, avg_us, mb, gbs, 100 * gbs / PEAK_GBS, ms))
    print("\nall kernels with traffic data: {:.1f} GB per step over {:.2f} ms of summed kernel time".format(tot_bytes / 1e3, tot_ms))


if __name__ == "__main__":
    main()
