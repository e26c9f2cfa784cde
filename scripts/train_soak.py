"""Short training soak on synthetic E2 data: `iters` iterations of train.training_step (pushforward unroll + forward + backward +
AdamW) with the native backward and, for comparison, with torch.autograd over the restatement, same seed.  Prints the loss
curves; both must fall together (they differ only by fp32 rounding, which chaotic training amplifies slowly)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd import autograd as A
from msmp_pde_amd.lem import LEM
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
from msmp_pde_amd.train import training_step
name = sys.argv[1] if len(sys.argv) > 1 else 'MSMP-PDE'
if '--own-gemm' in sys.argv:        # force the library's own row GEMMs + the factorised message_net_1 backward (default only from 32 768 edges on)
    sys.argv.remove('--own-gemm')
    mp.lib().msmp_tune(b'bwd_gemm', 2)
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 120
curves = {}
for path in (2, 0, 20, 10):        # 20 / 10: the same two paths once more (run-to-run spread of each: atomics in both)
    A.EXPLICIT_BACKWARD = 2 if path in (2, 20) else 0
    LEM.TRAIN_KERNELS = path in (2, 20)
    torch.manual_seed(0)
    case = make_case('E2', 16, seed=1, device='cuda', dtype=torch.float32)
    model = mp.MODEL_NAMES[name](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=6).cuda().train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-8, fused=True)
    gen = torch.Generator().manual_seed(5)
    losses = []
    for it in range(iters):
        steps = [int(s) for s in torch.randint(50, 200, (16,), generator=gen)]
        losses.append(float(training_step(model, case.creator, case.u_super, case.x, case.variables, steps, int(it % 3 > 0), opt)))
    assert all(l == l for l in losses), 'NaN loss'
    curves[path] = losses
    print(f'{name} backward path {path}: loss ' + ' '.join(f'{l:.3f}' for l in losses[::10]) + f' ... final {losses[-1]:.4f}', flush=True)
A.EXPLICIT_BACKWARD, LEM.TRAIN_KERNELS = 2, True
for a, b, what in ((2, 0, 'native vs autograd'), (2, 20, 'native vs native'), (0, 10, 'autograd vs autograd')):
    d = [abs(x - y) / y for x, y in zip(curves[a], curves[b])]
    print(f'{what}: relative difference of the loss curves: first 10 iterations max {max(d[:10]):.2e}, all {max(d):.2e}')
