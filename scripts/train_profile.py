"""Workload for `rocprofv3 --kernel-trace --stats -- python3 scripts/train_profile.py [name] [batch] [iters]`: training iterations
(train.training_step) of one model at one batch size, so the kernel statistics are those of the training path."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
from msmp_pde_amd.train import training_step
name = sys.argv[1] if len(sys.argv) > 1 else 'MSMP-PDE'
bsz = int(sys.argv[2]) if len(sys.argv) > 2 else 16
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
torch.manual_seed(0)
case = make_case('E2', bsz, seed=1, device='cuda', dtype=torch.float32)
model = mp.MODEL_NAMES[name](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=6).cuda().train()
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-8, fused=True)
steps = [60] * bsz
for _ in range(3): training_step(model, case.creator, case.u_super, case.x, case.variables, steps, 1, opt)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(iters): loss = training_step(model, case.creator, case.u_super, case.x, case.variables, steps, 1, opt)
torch.cuda.synchronize()
print(f'{name} batch {bsz}: {(time.perf_counter() - t0) / iters * 1e3:.2f} ms per training iteration over {iters} (+3 warm-up) iterations', flush=True)
