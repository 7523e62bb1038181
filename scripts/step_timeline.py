"""Timeline of one replayed training step from a rocprofv3 kernel trace: scripts/step_timeline.py <kernel_trace.csv> [tail_n]
Steps are cut at the optimizer's update kernel; prints, for the last complete step, each hardware queue's busy time, the
idle time of the busiest (main-chain) queue, and the last `tail_n` kernels with their queue and start / end offsets (us)."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
tail_n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
cuts = [i for i, r in enumerate(rows) if "adam_apply" in r[3] or "sgd_apply" in r[3]]
if len(cuts) < 3:
    sys.exit("fewer than three optimizer launches in the trace")
a, b = cuts[-2] + 1, cuts[-1] + 1
step = rows[a:b]
t0 = rows[cuts[-2]][1]
print("step: %d kernels, %.1f us from the previous update's end to this update's end" % (len(step), (step[-1][1] - t0) / 1e3))
busy = defaultdict(int)
count = defaultdict(int)
for s, e, q, _ in step:
    busy[q] += e - s
    count[q] += 1
for q in sorted(busy, key=lambda k: -count[k]):
    print("queue %-6s %4d kernels  busy %8.1f us" % (q, count[q], busy[q] / 1e3))
mainq = max(count, key=lambda k: count[k])
prev = t0
gaps = []
for s, e, q, n in step:
    if q == mainq:
        gaps.append(((s - prev) / 1e3, n))
        prev = e
print("main queue %s: idle %.1f us in %d gaps; gaps > 8 us:" % (mainq, sum(g for g, _ in gaps), len(gaps)))
for g, n in sorted(gaps, reverse=True)[:12]:
    print("   %7.1f us in front of %s" % (g, n[:90]))
print("last %d kernels:" % tail_n)
for s, e, q, n in step[-tail_n:]:
    print("  q%-5s %9.1f -> %9.1f  %s" % (q, (s - t0) / 1e3, (e - t0) / 1e3, n[:100]))
