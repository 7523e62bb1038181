"""Per-kernel summary of a rocprofv3 kernel_stats.csv: prof_summary.py <csv> [steps] [rows].  Names are demangled by
scripts/kernel_names.py (profiles are taken with rocprofv3 -M, see there); rows with the same short name are merged."""
import csv, os, sys
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from kernel_names import short
f = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
agg = OrderedDict()
for r in csv.DictReader(open(f)):
    a = agg.setdefault(short(r['Name']), [0, 0.0])
    a[0] += int(r['Calls'])
    a[1] += float(r['TotalDurationNs'])
tot = sum(v[1] for v in agg.values())
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
for name, (calls, ns) in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print("%-70s calls/step %5.1f  ms/step %7.3f  avg %8.1f us  %5.1f%%" % (name[:70], calls / steps, ns / 1e6 / steps, ns / calls / 1e3, 100.0 * ns / tot))
print("total ms/step %.2f" % (tot / 1e6 / steps))
