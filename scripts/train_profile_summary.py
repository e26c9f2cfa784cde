"""Summarise the rocprofv3 database of `scripts/train_profile.py` (kernel time and launches per training iteration)."""
import sqlite3, sys, collections
db, n_it = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 23
c = sqlite3.connect(db)
rows = c.execute("select name, start, end from kernels order by start").fetchall()
tot = sum(e - s for _, s, e in rows)
print(f'{len(rows) / n_it:.0f} launches and {tot / n_it / 1e6:.2f} ms of kernel time per iteration ({n_it} iterations incl. warm-up)')
agg = collections.defaultdict(lambda: [0, 0])
for name, s, e in rows:
    k = name[:90]
    agg[k][0] += 1; agg[k][1] += e - s
for k, (cnt, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[3]) if len(sys.argv) > 3 else 25]:
    print(f'{t / n_it / 1e3:9.1f} us/iter {cnt / n_it:7.1f} calls/iter {t / cnt / 1e3:8.1f} us/call  {k}')
