"""Per-step view of a `rocprofv3 --kernel-trace --stats` CSV: python profiles/summarize_kernel_stats.py FILE.csv STEPS [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('kernel time per step: %.3f ms, launches per step: %.1f' % (tot / steps / 1e6, sum(int(r['Calls']) for r in rows) / steps))
for r in rows[:top]:
    print('%-84s %5.1f /step  avg %7.1f us  %7.1f us/step  %4.1f%%' % (r['Name'][:84], int(r['Calls']) / steps, float(r['AverageNs']) / 1e3,
                                                                    float(r['TotalDurationNs']) / steps / 1e3, float(r['Percentage'])))
