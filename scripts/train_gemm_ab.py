import sys, os, time
sys.path.insert(0, '/root/repo')
import torch
import msmp_pde_amd as mp
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
from msmp_pde_amd.train import training_step
L = mp.lib()
for name in ('MSMP-PDE', 'MP-PDE'):
  for bsz in (16, 64):
    for mode in (0, 1, 0, 1):
        L.msmp_tune(b'bwd_gemm', mode)
        torch.manual_seed(0)
        case = make_case('E2', bsz, seed=1, device='cuda', dtype=torch.float32)
        model = mp.MODEL_NAMES[name](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=6).cuda().train()
        opt = mp.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-8)
        steps = [60] * bsz
        for _ in range(4): training_step(model, case.creator, case.u_super, case.x, case.variables, steps, 1, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n): loss = training_step(model, case.creator, case.u_super, case.x, case.variables, steps, 1, opt)
        torch.cuda.synchronize()
        print(f'{name} batch {bsz} bwd_gemm={mode} ({"rocblas" if mode == 0 else "own"}): {(time.perf_counter() - t0) / n * 1e3:.2f} ms per iteration, loss {float(loss):.5f}', flush=True)
