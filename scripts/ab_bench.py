"""A/B helper: run bench.py's workload in ONE process for several msmp_tune settings, interleaved rounds.
usage: python scripts/ab_bench.py key v1 v2 ... [--rounds R] [--steps K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import msmp_pde_amd as mp
from msmp_pde_amd import _lib
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS

args = sys.argv[1:]
rounds = int(args[args.index('--rounds') + 1]) if '--rounds' in args else 3
steps = int(args[args.index('--steps') + 1]) if '--steps' in args else 5
model_name = args[args.index('--model') + 1] if '--model' in args else 'MSMP-PDE'
key = args[0]
vals = [int(v) for v in args[1:] if v.lstrip('-').isdigit() and args[args.index(v) - 1] not in ('--rounds', '--steps')]
L = mp.lib()
torch.manual_seed(0)
case = make_case('E2', 2048, seed=1000, device='cuda', dtype=torch.float32)
model = mp.MODEL_NAMES[model_name](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=6).cuda().eval()
data, labels = case.creator.create_data(case.u_super, [50] * 2048)
graph = case.creator.create_graph(data, labels, case.x, case.variables, [50] * 2048)
names = {0: 'edge', 2: 'node_update', 3: 'norm', 4: 'lem', 5: 'node_proj'}
with torch.no_grad():
    pred = model(graph)
    res = {v: [] for v in vals}
    for r in range(rounds):
        for v in vals:
            L.msmp_tune(key.encode(), v)
            pred = model(graph)
            L.msmp_timing_reset(); L.msmp_timing_enable(0b111111)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(steps):
                pred = model(graph)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps * 1e3
            L.msmp_timing_enable(0)
            ks = {names[k]: round(_lib.timing_read(k)[1] / steps, 3) for k in names}
            res[v].append((round(dt, 3), ks))
for v in vals:
    print(key, v, 'forward ms:', [x[0] for x in res[v]], 'kernels ms/step (last round):', res[v][-1][1])
