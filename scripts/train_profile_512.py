"""One training iteration of MSMP-PDE at batch 512 (E2) for a rocprofv3 kernel trace:
   rocprofv3 --kernel-trace --stats -- python3 scripts/train_profile_512.py [batch]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
from msmp_pde_amd.train import training_step
bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 512
torch.manual_seed(0)
case = make_case('E2', bsz, seed=1, device='cuda', dtype=torch.float32)
model = mp.MODEL_NAMES['MSMP-PDE'](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=6).cuda().train()
opt = mp.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-8)
steps = [60] * bsz
for _ in range(2): training_step(model, case.creator, case.u_super, case.x, case.variables, steps, 1, opt)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 5
for _ in range(n): loss = training_step(model, case.creator, case.u_super, case.x, case.variables, steps, 1, opt)
torch.cuda.synchronize()
print(f'batch {bsz}: {(time.perf_counter() - t0) / n * 1e3:.2f} ms per training iteration, loss {float(loss):.4f}')
