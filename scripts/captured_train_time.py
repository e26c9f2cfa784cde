"""Optimisation step (forward + loss + backward + AdamW on a prepared batch; experiments/train_helper.py:125-141) eager vs as one
hipGraph launch (train.CapturedTrainStep), E2, the reference's batch size 16 and larger ones.
    python scripts/captured_train_time.py [batch sizes ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd import train as T
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [16, 64]
names = [a for a in sys.argv[1:] if a in mp.MODEL_NAMES] or ['MSMP-PDE', 'Gated', 'MP-PDE']
modes = [a for a in sys.argv[1:] if a in ('eager', 'captured')] or ['eager', 'captured']
for name in names:
    for bsz in sizes:
        torch.manual_seed(0)
        case = make_case('E2', bsz, seed=1, device='cuda', dtype=torch.float32)
        steps = [60] * bsz
        data, labels = case.creator.create_data(case.u_super, steps)
        graph = case.creator.create_graph(data, labels, case.x, case.variables, steps)
        res = {}
        for mode in modes:
            torch.manual_seed(1)
            model = mp.MODEL_NAMES[name](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=6).cuda().train()
            opt = mp.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-8, capturable=True)
            if mode == 'captured':
                step = T.CapturedTrainStep(model, opt, graph)
                run = lambda: step(graph)
            else:
                def run():
                    opt.zero_grad(set_to_none=True)
                    loss = T.dp_loss_backward(model, graph)
                    opt.step()
                    return loss
                for _ in range(3): run()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 20
            for _ in range(n): loss = run()
            torch.cuda.synchronize()
            res[mode] = ((time.perf_counter() - t0) / n * 1e3, float(loss))
        print(f'{name:9s} batch {bsz:4d}: ' + ', '.join(f'{m} {res[m][0]:7.2f} ms' for m in modes) + ' per optimisation step (loss after the timed steps '
              + ' / '.join(f'{res[m][1]:.5f}' for m in modes) + ')', flush=True)
