"""Fused AdamW on the HIP kernel msmp_adamw_f32 (the reference's optimizer is
`optim.AdamW(model.parameters(), lr=args.lr)`, experiments/train.py:410).  A torch.optim.Optimizer subclass with the same
constructor arguments, state layout (`step`, `exp_avg`, `exp_avg_sq`) and update rule as torch.optim.AdamW, so schedulers
(train.py:411 MultiStepLR), state_dict() and the package's optimizer-step hook (packed-weight invalidation) work unchanged;
every parameter tensor of the model is updated by one or two kernel launches."""
import ctypes

import torch

from ._lib import lib, check, current_stream


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('invalid AdamW hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = lib()
        for group in self.param_groups:
            plan = group.get('_msmp_plan')
            live = [p for p in group['params'] if p.grad is not None]
            if plan is None or plan['ids'] != [id(p) for p in live]:
                plan = group['_msmp_plan'] = self._plan(live)
            if not live:
                continue
            for p in live:
                if p.grad.is_sparse or not p.grad.is_contiguous():
                    raise RuntimeError('msmp_pde_amd.optim.AdamW: dense contiguous gradients only')
            plan['t'] += 1
            plan['step'] += 1                                # ONE tensor object shared by the `step` entries of all these parameters
            n = len(live)
            grads = (ctypes.c_void_p * n)(*[p.grad.data_ptr() for p in live])     # zero_grad(set_to_none=True) re-allocates them
            b1, b2 = group['betas']
            check(L.msmp_adamw_f32(n, plan['p'], grads, plan['m'], plan['v'], plan['numel'], float(group['lr']), float(b1), float(b2),
                                   float(group['eps']), float(group['weight_decay']), plan['t'], current_stream()), 'msmp_adamw_f32')
        return loss

    def _plan(self, live):
        """Pointer tables of one parameter group (rebuilt only when the set of parameters with gradients changes)."""
        t = None
        for p in live:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise RuntimeError('msmp_pde_amd.optim.AdamW: float32 contiguous CUDA parameters only')
            st = self.state[p]
            if not st:
                st['step'] = torch.zeros((), dtype=torch.float32)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            ti = int(st['step'].item())
            if t is not None and ti != t:
                raise RuntimeError('msmp_pde_amd.optim.AdamW: the parameters of a group must share their step count')
            t = ti
        n = len(live)
        shared = torch.full((), float(t or 0), dtype=torch.float32)
        for p in live:
            self.state[p]['step'] = shared                   # torch's state layout keeps `step` per parameter: same value, one object
        arr = lambda ts: (ctypes.c_void_p * n)(*[x.data_ptr() for x in ts])
        return {'ids': [id(p) for p in live], 't': t or 0, 'step': shared, 'p': arr(live),
                'm': arr([self.state[p]['exp_avg'] for p in live]), 'v': arr([self.state[p]['exp_avg_sq'] for p in live]),
                'numel': (ctypes.c_int64 * n)(*[p.numel() for p in live])}

    def state_dict(self):
        sd = super().state_dict()
        for g in sd['param_groups']:
            g.pop('_msmp_plan', None)
        return sd
