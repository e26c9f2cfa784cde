"""A/B of msmp_tune settings on the default bench workload in ONE gpurun call (separate processes, interleaved):
    python scripts/ab_tune.py [rounds] [bench args ...] -- tile_persist=0 tile_persist=3 "lem=3,tile_persist=2" """
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
argv = sys.argv[1:]
cut = argv.index('--')
pre, sets = argv[:cut], argv[cut + 1:]
rounds = int(pre[0]) if pre else 3
extra = pre[1:]
res = {s: [] for s in sets}
for r in range(rounds):
    for s in sets:
        tune = [x for kv in s.split(',') if kv for x in ('--tune', kv)]
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '40', '--warmup', '3', '--no-cpu-baseline', '--no-extras'] + extra + tune,
                             capture_output=True, text=True)
        lines = [x for x in out.stdout.splitlines() if x.startswith('{')]
        if not lines:
            print(out.stderr[-2000:])
            sys.exit(1)
        d = json.loads(lines[-1])
        res[s].append((round(d['ms_per_step'], 3), round(d['roofline']['avg_launch_ms'] * 1e3, 1)))
for s in sets:
    print(s, 'ms/step, dominant kernel us/launch:', res[s], flush=True)
