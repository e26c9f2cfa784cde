"""A/B of two builds of the library on the default bench workload in ONE gpurun call (separate processes, interleaved):
    python scripts/ab_libs.py libA.so libB.so [libC.so ...] [rounds]"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = [a for a in sys.argv[1:] if not a.isdigit()]
rounds = int(sys.argv[-1]) if sys.argv[-1].isdigit() else 3
res = {l: [] for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, MSMP_LIB_PATH=os.path.join(ROOT, 'msmp-pde_amd', l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '40', '--warmup', '3', '--no-cpu-baseline', '--no-extras'],
                             env=env, capture_output=True, text=True)
        d = json.loads([x for x in out.stdout.splitlines() if x.startswith('{')][-1])
        res[l].append((round(d['ms_per_step'], 3), round(d['roofline']['avg_launch_ms'] * 1e3, 1)))
for l in libs:
    print(l, 'ms/step, edge kernel us/launch:', res[l])
