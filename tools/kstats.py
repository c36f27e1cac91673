"""Per-step kernel table from a rocprofv3 --stats CSV (test infrastructure):  python tools/kstats.py <kernel_stats.csv> <steps> [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
for r in rows[:top]:
    t = float(r['TotalDurationNs']) / 1e6 / steps
    print('%-100s %6.1f /step %9.1f us/step  avg %8.1f us' % (r['Name'][:100], int(r['Calls']) / steps, t * 1000, float(r['AverageNs']) / 1e3))
print('sum of kernel time per step: %.3f ms' % (sum(float(r['TotalDurationNs']) for r in rows) / 1e6 / steps))
