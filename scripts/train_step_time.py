"""Wall time of one training iteration (train.training_step: pushforward unroll under no_grad + forward/backward + AdamW)
on the HIP forward / native backward path, E2, reference batch size 16 and larger.  Optimizer: msmp_pde_amd.optim.AdamW
(msmp_adamw_f32; `--torch-adamw` times torch.optim.AdamW(fused=True) instead)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
from msmp_pde_amd.train import training_step
from msmp_pde_amd.lem import LEM
variants = [('MSMP-PDE', True), ('Gated', True), ('MP-PDE', True)]
TORCH_ADAMW = '--torch-adamw' in sys.argv
for name, lem_kernels in variants:
    LEM.TRAIN_KERNELS = lem_kernels
    for bsz in (16, 128, 512):
        torch.manual_seed(0)
        case = make_case('E2', bsz, seed=1, device='cuda', dtype=torch.float32)
        model = mp.MODEL_NAMES[name](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=6).cuda().train()
        opt = (torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-8, fused=True) if TORCH_ADAMW
               else mp.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-8))
        steps = [60] * bsz
        for _ in range(4): training_step(model, case.creator, case.u_super, case.x, case.variables, steps, 1, opt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 5
        for _ in range(n): loss = training_step(model, case.creator, case.u_super, case.x, case.variables, steps, 1, opt)
        torch.cuda.synchronize()
        print(f'{name:9s}{"" if lem_kernels else " (LEM in PyTorch)"} batch {bsz:4d}: {(time.perf_counter() - t0) / n * 1e3:8.2f} ms per training iteration (1 unrolled step), loss {float(loss):.4f}', flush=True)
