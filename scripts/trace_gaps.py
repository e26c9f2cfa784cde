"""Kernel timeline of a few rollout steps from a rocprofv3 --kernel-trace CSV: per-kernel durations, the gaps between consecutive
kernels, and per step the sum of kernel time vs wall time.   python scripts/trace_gaps.py <kernel_trace.csv> [kernels per step]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the steady state: the last 40 % of the trace
rows = rows[int(len(rows) * 0.6):]
t0 = int(rows[0]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows)
wall = int(rows[-1]['End_Timestamp']) - t0
gaps = [max(0, int(b['Start_Timestamp']) - int(a['End_Timestamp'])) for a, b in zip(rows, rows[1:])]
print(f'{len(rows)} kernels, wall {wall / 1e3:.1f} us, kernel time {busy / 1e3:.1f} us ({100 * busy / wall:.1f} %), summed gaps {sum(gaps) / 1e3:.1f} us, mean gap {sum(gaps) / len(gaps) / 1e3:.2f} us')
per = collections.defaultdict(list)
for r in rows:
    per[r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f'  {sum(v) / 1e3:8.3f} ms  x{len(v):4d}  avg {sum(v) / len(v):7.1f} us   {k}')
