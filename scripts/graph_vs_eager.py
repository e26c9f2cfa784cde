"""Eager forward vs hipGraph replay (Solver.capture) of the rollout forward at several batch sizes (E2 MSMP-PDE)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
for bsz in (32, 128, 512, 2048):
    case = make_case('E2', bsz, seed=1000, device='cuda', dtype=torch.float32)
    model = mp.MODEL_NAMES['MSMP-PDE'](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=6).cuda().eval()
    data, labels = case.creator.create_data(case.u_super, [50] * bsz)
    graph = case.creator.create_graph(data, labels, case.x, case.variables, [50] * bsz)
    with torch.no_grad():
        model(graph)
        step = model.capture(graph)
        res = {}
        for f, nm in ((lambda: model(graph), 'eager'), (lambda: step(graph), 'graph')):
            for _ in range(3): f()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 50 if bsz <= 512 else 20
            for _ in range(n): f()
            torch.cuda.synchronize(); res[nm] = (time.perf_counter() - t0) / n * 1e3
    print(f'{bsz:5d} graphs: eager {res["eager"]:.3f} ms, hipGraph replay {res["graph"]:.3f} ms  ({res["eager"] / res["graph"]:.2f}x)', flush=True)
