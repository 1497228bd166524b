"""Summary of a rocprofv3 --kernel-trace database of profiles/exp_train_timeline.py: per training step (delimited by the front-end's
first kernel) the number of kernels, the sum of their durations, the busy time (union of intervals) and the step period."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
marks = [i for i, r in enumerate(rows) if r[0].startswith(('scene_orig', 'frontend_small'))]
marks = marks[len(marks) // 2: len(marks) // 2 + 11]
out = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b]
    tot = sum(r[2] - r[1] for r in seg)
    iv = sorted((r[1], r[2]) for r in seg)
    busy, cur0, cur1 = 0, iv[0][0], iv[0][1]
    for s, e in iv[1:]:
        if s <= cur1: cur1 = max(cur1, e)
        else: busy += cur1 - cur0; cur0, cur1 = s, e
    busy += cur1 - cur0
    out.append((len(seg), tot / 1e3, busy / 1e3, (rows[b][1] - rows[a][1]) / 1e3, len({r[3] for r in seg})))
print('kernels  sum_us  busy_us  period_us  queues')
for o in out: print('%7d %7.0f %8.0f %10.0f %7d' % o)
if len(sys.argv) > 2:
    a, b = marks[0], marks[1]
    t0 = rows[a][1]
    for r in rows[a:b]: print(f'{(r[1] - t0) / 1e3:9.1f} {(r[2] - r[1]) / 1e3:7.1f} q{r[3]} {r[0][:80]}')
