#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (scripts/profile_gpu.sh) into the committed summaries under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (top rows)
  profiles/<tag>_summary.md         per-kernel time per step + HBM bytes per launch of our kernels
  profiles/traffic.json             HBM bytes per launch of the dominant kernels (read by bench.py)
gfx950 corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE is in KiB and reports exactly half of
the bytes of a wide coalesced streaming read -> doubled; WRITE_SIZE (KiB) is exact for 16-B/lane stores."""
import csv
import json
import os
import sys
from collections import defaultdict

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
KEEP_TRAFFIC_JSON = '--no-traffic-json' in sys.argv      # profiles of the non-default configs must not replace the default workload's stamp
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')
os.makedirs(dst, exist_ok=True)

rows = list(csv.DictReader(open(os.path.join(src, 'kernel_stats.csv'))))
with open(os.path.join(dst, f'{tag}_kernel_stats.csv'), 'w') as f:
    w = csv.writer(f)
    w.writerow(['Name', 'Calls', 'TotalDurationNs', 'AverageNs', 'Percentage', 'MinNs', 'MaxNs'])
    for r in rows[:40]:
        w.writerow([r['Name'][:160], r['Calls'], r['TotalDurationNs'], r['AverageNs'], r['Percentage'], r['MinNs'], r['MaxNs']])
tot = sum(float(r['TotalDurationNs']) for r in rows)
bench = {}
p = os.path.join(src, 'trace_bench.json')
if os.path.exists(p) and os.path.getsize(p):
    bench = json.loads(open(p).read().strip().splitlines()[-1])


def counter(path, name):
    per = defaultdict(list)
    if not os.path.exists(path):
        return per
    for r in csv.DictReader(open(path)):
        if r.get('Counter_Name') == name:
            per[r['Kernel_Name'].split('(')[0]].append(float(r['Counter_Value']))
    return per


fetch = counter(os.path.join(src, 'pmc_fetch.csv'), 'FETCH_SIZE')
write = counter(os.path.join(src, 'pmc_write.csv'), 'WRITE_SIZE')
traffic = {}
_stamp_file = os.path.join(src, 'stamp.json')
_wl = json.load(open(_stamp_file)).get('workload', 'E2/MSMP-PDE/2048/n3') if os.path.exists(_stamp_file) else 'E2/MSMP-PDE/2048/n3'
lines = [f'# rocprofv3 summary `{tag}` (MI355X, bench.py workload {_wl}: experiment / model / graphs / neighbours)', '']
if bench:
    lines += [f"bench line of the traced run: {bench['value']:.2f} rollout-steps/s, {bench['ms_per_step']:.2f} ms/step; "
              f"edge_mlp avg launch {bench['roofline']['avg_launch_ms']:.3f} ms (HIP events) -> "
              f"{bench['roofline']['achieved']:.1f} TFLOP/s fp32-equivalent = {100 * bench['roofline']['frac']:.1f} % of the roofline peak "
              f"({bench['roofline']['peak']:.0f} TFLOP/s: {bench['roofline'].get('peak_note', 'dense fp32 MFMA peak')})", '']
lines += [f'total kernel time in trace: {tot / 1e6:.1f} ms', '',
          '| kernel | calls | avg us | total ms | % |', '|---|---|---|---|---|']
for r in rows[:25]:
    lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs']) / 1e3:.1f} | {float(r['TotalDurationNs']) / 1e6:.2f} | {r['Percentage']} |")
lines += ['', '## HBM traffic per launch (separate --pmc passes; FETCH_SIZE doubled per the gfx950 correction)', '',
          '| kernel | launches | FETCH_SIZE KiB (raw, mean) | read bytes (x2 x1024) | WRITE_SIZE KiB (mean) | write bytes | total HBM bytes/launch |',
          '|---|---|---|---|---|---|---|']
for k in sorted(set(fetch) | set(write)):
    fm = sum(fetch[k]) / len(fetch[k]) if fetch.get(k) else float('nan')
    wm = sum(write[k]) / len(write[k]) if write.get(k) else float('nan')
    rb, wb = fm * 2 * 1024, wm * 1024
    short = k.replace('void ', '').replace('msmp::', '').split('<')[0]
    traffic[short] = {'fetch_kib_raw': fm, 'write_kib': wm, 'read_bytes': rb, 'write_bytes': wb, 'hbm_bytes_per_launch': rb + wb}
    lines.append(f'| `{short}` | {len(fetch.get(k, []))} | {fm:.0f} | {rb:.3e} | {wm:.0f} | {wb:.3e} | {rb + wb:.3e} |')
open(os.path.join(dst, f'{tag}_summary.md'), 'w').write('\n'.join(lines) + '\n')
if traffic and not KEEP_TRAFFIC_JSON:
    # stamp: bench.py reports `roofline.traffic` only when the kernel sources, the workload and the matrix path are the ones
    # these counters were taken on (no silent staleness)
    sys.path.insert(0, ROOT)
    import bench as _bench
    cfg = bench.get('config', {}) if bench else {}
    dominant = max((r for r in rows if 'msmp::' in r['Name'] and 'edge' in r['Name']), key=lambda r: float(r['TotalDurationNs']), default=None)
    dom_short = dominant['Name'].replace('void ', '').replace('msmp::', '').split('<')[0].split('(')[0] if dominant else None
    stamp_file = os.path.join(src, 'stamp.json')
    stamp = json.load(open(stamp_file)) if os.path.exists(stamp_file) else {}
    stamp.update({'tag': tag, 'dominant_kernel': dom_short})
    stamp.setdefault('source_hash', _bench.source_hash())
    traffic['_stamp'] = stamp
    json.dump(traffic, open(os.path.join(dst, 'traffic.json'), 'w'), indent=1)
print('\n'.join(lines))
