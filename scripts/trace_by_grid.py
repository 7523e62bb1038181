"""Per (kernel, grid size) durations from a rocprofv3 kernel_trace.csv: trace_by_grid.py <csv> <name substring> [steps]"""
import csv, os, sys
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short
f, sub = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
agg = OrderedDict()
for r in csv.DictReader(open(f)):
    n = short(r['Kernel_Name'])
    if sub not in n:
        continue
    key = (n, int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), int(r['Grid_Size_Y']), r.get('LDS_Block_Size', ''))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for (n, gx, gy, lds), (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-50s wgs %6d x %3d lds %6s calls/step %4.1f avg %8.1f us  ms/step %6.3f" % (n[:50], gx, gy, lds, c / steps, ns / c / 1e3, ns / 1e6 / steps))
