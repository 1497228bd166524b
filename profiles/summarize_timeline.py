"""One window of a rocprofv3 --kernel-trace database as text: python summarize_timeline.py <db> <first-kernel-prefix> <kernels>
(start / duration in us relative to the window's first kernel, queue, kernel name)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if r[0].startswith(tuple(sys.argv[2].split(',')))]
i0 = marks[len(marks) // 2]
t0 = rows[i0][1]
print('# start_us  duration_us  queue  kernel')
for r in rows[i0:i0 + int(sys.argv[3])]:
    print(f'{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f}  q{r[3]}  {r[0][:90]}')
