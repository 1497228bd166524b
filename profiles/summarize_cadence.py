"""Launch cadence of a rocprofv3 --kernel-trace database: python summarize_cadence.py <db> [kernel-substring] [window]
One line per launch of the matching kernel (default traj_chain): index, queue, start (ms from the first), duration, distance to the previous
launch's start, launches in flight at its start; then the mean start-to-start distance per window of `window` launches -- where a run
speeds up or slows down -- and, per queue, every OTHER kernel with its share of the trace (what sits between the launches)."""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
key = sys.argv[2] if len(sys.argv) > 2 else 'traj_chain'
win = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rows = db.execute('select name,start,end,queue_id from kernels order by start').fetchall()
ch = [r for r in rows if key in r[0]]
t0 = ch[0][1]
print(f'# {len(ch)} launches of *{key}*; columns: idx queue start_ms dur_us d_start_us in_flight')
ends = []
prev = None
for i, (nm, a, b, q) in enumerate(ch):
    ends = [e for e in ends if e > a]
    print(f'{i:4d} q{q} {(a - t0) / 1e6:9.3f} {(b - a) / 1e3:8.1f} {((a - prev) / 1e3 if prev else 0):8.1f} {len(ends) + 1}')
    ends.append(b)
    prev = a
print('# window means (start-to-start, us):')
st = [r[1] for r in ch]
for w in range(0, len(st) - 1, win):
    seg = st[w:w + win + 1]
    if len(seg) > 1:
        print(f'#   launches {w:4d}..{w + len(seg) - 1:4d}: {(seg[-1] - seg[0]) / (len(seg) - 1) / 1e3:8.1f}')
print(f'# whole trace: first start -> last end {(ch[-1][2] - t0) / 1e6:.3f} ms over {len(ch)} launches = {(ch[-1][2] - t0) / len(ch) / 1e3:.1f} us per launch')
other = collections.defaultdict(lambda: [0, 0])
for nm, a, b, q in rows:
    if key in nm or a < t0:
        continue
    o = other[(q, nm[:70])]
    o[0] += 1; o[1] += b - a
print('# other kernels after the first launch: queue, name, count, total ms')
for (q, nm), (c, tot) in sorted(other.items(), key=lambda kv: -kv[1][1])[:24]:
    print(f'#   q{q} {nm:70s} {c:6d} {tot / 1e6:9.3f}')
