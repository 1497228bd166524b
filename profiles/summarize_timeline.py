"""One window of a rocprofv3 --kernel-trace database as text: python summarize_timeline.py <db> <kernel-name-substring[,..]> <kernels>
(start / duration in us relative to the window's first kernel, queue, kernel name), followed by the busy time of the kernels that
match the substring over the WHOLE trace: length of the union of their [start, end] intervals, mean duration, launches in flight --
the cross-check of the HIP-event union bench.py reports (roofline.busy_time_s)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
keys = tuple(sys.argv[2].split(','))
marks = [i for i, r in enumerate(rows) if any(k in r[0] for k in keys)]
i0 = marks[len(marks) // 2]
t0 = rows[i0][1]
print('# start_us  duration_us  queue  kernel')
for r in rows[i0:i0 + int(sys.argv[3])]:
    print(f'{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f}  q{r[3]}  {r[0][:90]}')
iv = sorted((rows[i][1], rows[i][2]) for i in marks)
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # leading launches to leave out (warm-up)
iv = iv[skip:]
busy, cur0, cur1 = 0, iv[0][0], iv[0][1]
for a, b in iv[1:]:
    if a <= cur1:
        cur1 = max(cur1, b)
    else:
        busy += cur1 - cur0
        cur0, cur1 = a, b
busy += cur1 - cur0
tot = sum(b - a for a, b in iv)
print(f'# {len(iv)} launches matching {keys}: mean duration {tot / len(iv) / 1e3:.1f} us, union of intervals {busy / 1e3:.1f} us '
      f'= {busy / len(iv) / 1e3:.1f} us per launch, launches in flight {tot / busy:.2f}, first start -> last end {(iv[-1][1] - iv[0][0]) / 1e3:.1f} us')
