"""Occupancy of the queues over the last replayed steps of a rocprofv3 kernel trace.

usage: timeline.py KERNEL_TRACE.csv [marker-kernel-substring]
Steps are delimited by the marker kernel (default: the loss kernel, launched once per step).  Prints per step:
wall = first start to last end, busy = union of kernel intervals, main = union over the queue that ran the marker,
side = union over the other queues, both = time with kernels of the main queue and of a side queue running together,
idle = wall - busy; then the kernels of the main queue with the largest summed wait in front of them (gap between the
previous main-queue kernel's end and this one's start).
"""
import csv
import sys
from collections import defaultdict


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s <= ce:
            ce = max(ce, e)
        else:
            tot += ce - cs
            cs, ce = s, e
    if cs is not None:
        tot += ce - cs
    return tot


def inter(a, b):
    a, b = sorted(a), sorted(b)
    # merge each
    def merge(iv):
        out = []
        for s, e in iv:
            if out and s <= out[-1][1]:
                out[-1][1] = max(out[-1][1], e)
            else:
                out.append([s, e])
        return out
    a, b = merge(a), merge(b)
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "xent_kernel"
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]) for r in rows]
ev.sort()
marks = [i for i, e in enumerate(ev) if marker in e[3]]
mainq = ev[marks[-1]][2]
print("steps seen:", len(marks), "main queue:", mainq)
gaps = defaultdict(lambda: [0, 0])
for k in range(max(1, len(marks) - 4), len(marks) - 1):
    # a step runs from the first kernel after the previous step's last optimizer kernel; approximate by marker to marker
    seg = ev[marks[k]:marks[k + 1]]
    t0, t1 = seg[0][0], max(e[1] for e in seg)
    # only count up to the next marker's start
    t1 = ev[marks[k + 1]][0]
    allv = [(s, min(e, t1)) for s, e, q, n in seg]
    mainv = [(s, min(e, t1)) for s, e, q, n in seg if q == mainq]
    sidev = [(s, min(e, t1)) for s, e, q, n in seg if q != mainq]
    wall = t1 - t0
    busy = union(allv)
    print("step %d: wall %.3f ms  busy %.3f  main %.3f  side %.3f  both %.3f  idle %.3f  kernels %d (main %d)" % (
        k, wall / 1e6, busy / 1e6, union(mainv) / 1e6, union(sidev) / 1e6, inter(mainv, sidev) / 1e6, (wall - busy) / 1e6,
        len(seg), len(mainv)))
    prev = None
    for s, e, q, n in seg:
        if q != mainq:
            continue
        if prev is not None and s > prev:
            g = gaps[n[:60]]
            g[0] += s - prev
            g[1] += 1
        prev = e if prev is None else max(prev, e)
print("largest main-queue waits (summed over the steps above):")
for n, (t, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:14]:
    print("  %-60s %8.1f us over %4d gaps (%.2f us each)" % (n, t / 1e3, c, t / 1e3 / c))
