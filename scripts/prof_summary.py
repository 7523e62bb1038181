import csv, sys
f = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 22]:
    print("%-70s calls/step %5.1f  ms/step %7.3f  avg %8.1f us  %5.1f%%" % (r['Name'][:70], int(r['Calls']) / steps, float(r['TotalDurationNs']) / 1e6 / steps, float(r['AverageNs']) / 1e3, float(r['Percentage'])))
print("total ms/step %.2f" % (tot / 1e6 / steps))
