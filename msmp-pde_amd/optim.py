"""Fused AdamW on the HIP kernel msmp_adamw_f32 (the reference's optimizer is
`optim.AdamW(model.parameters(), lr=args.lr)`, experiments/train.py:410).  A torch.optim.Optimizer subclass with the same
constructor arguments, state layout (one `step`, `exp_avg`, `exp_avg_sq` per parameter) and update rule as torch.optim.AdamW, so schedulers
(train.py:411 MultiStepLR), state_dict() and the package's optimizer-step hook (packed-weight invalidation) work unchanged;
every parameter tensor of the model is updated by one or two kernel launches.

`capturable=True` (the meaning of torch.optim.AdamW's flag): the step count and the learning rate live in device memory
(msmp_adamw_capturable_f32), so a hipGraph that contains `step()` advances correctly on every replay (train.CapturedTrainStep);
`param_groups[i]['lr']` is still what schedulers write, its value is copied to the device at the start of each `step()` /
`sync_lr()` call."""
import ctypes

import torch

from ._lib import lib, check, current_stream


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, capturable=False):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError('invalid AdamW hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=bool(capturable)))

    def sync_lr(self):
        """capturable groups: copy param_groups[i]['lr'] to the device word the kernels read (a scheduler changed it between two
        replays of a captured step; call it outside the capture)."""
        for group in self.param_groups:
            dev = group.get('_msmp_dev')
            if dev is not None and dev['lr_host'] != float(group['lr']):
                dev['lr'].fill_(float(group['lr']))
                dev['lr_host'] = float(group['lr'])

    def _device_state(self, group, live):
        """Device-side step count (int64) and learning rate of a capturable group; all parameters of the group share the count."""
        dev = group.get('_msmp_dev')
        if dev is None:
            d = live[0].device
            t0 = {int(self.state[p]['step'].item()) for p in live if self.state[p]}
            if len(t0) > 1:
                raise RuntimeError('msmp_pde_amd.optim.AdamW(capturable=True): the parameters of a group must share one step count')
            dev = group['_msmp_dev'] = {'step': torch.full((1,), t0.pop() if t0 else 0, dtype=torch.int64, device=d),
                                        'lr': torch.full((1,), float(group['lr']), dtype=torch.float32, device=d), 'lr_host': float(group['lr'])}
        return dev

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        L = lib()
        for group in self.param_groups:
            live = [p for p in group['params'] if p.grad is not None]
            if not live:
                continue
            for p in live:
                if p.grad.is_sparse or not p.grad.is_contiguous():
                    raise RuntimeError('msmp_pde_amd.optim.AdamW: dense contiguous gradients only')
            b1, b2 = group['betas']
            if group.get('capturable'):
                dev = self._device_state(group, live)
                if not torch.cuda.is_current_stream_capturing():
                    self.sync_lr()
                plans = self._plans(group, live)
                if len(plans) != 1:
                    # the device-side count is ONE word per group: parameters with another host-side step (state loaded at step N plus
                    # a parameter that gets its first gradient now) would be skipped or mis-corrected
                    raise RuntimeError('msmp_pde_amd.optim.AdamW(capturable=True): the live parameters of a group carry '
                                       f'{len(plans)} different step counts; a capturable group shares one')
                plan = plans[0][1]
                n = len(plan['params'])
                grads = (ctypes.c_void_p * n)(*[p.grad.data_ptr() for p in plan['params']])
                check(L.msmp_adamw_capturable_f32(n, plan['p'], grads, plan['m'], plan['v'], plan['numel'], dev['lr'].data_ptr(), float(b1), float(b2),
                                                  float(group['eps']), float(group['weight_decay']), dev['step'].data_ptr(), current_stream()),
                      'msmp_adamw_capturable_f32')
                continue
            # One launch per step count: parameters that received their first gradient later than their group peers (torch handles
            # that) carry their own count, like everything else in torch's per-parameter state layout.
            for t, plan in self._plans(group, live):
                torch._foreach_add_(plan['steps'], 1)         # one `step` tensor PER parameter (torch.optim.AdamW's state layout)
                n = len(plan['params'])
                grads = (ctypes.c_void_p * n)(*[p.grad.data_ptr() for p in plan['params']])     # zero_grad(set_to_none=True) re-allocates them
                check(L.msmp_adamw_f32(n, plan['p'], grads, plan['m'], plan['v'], plan['numel'], float(group['lr']), float(b1), float(b2),
                                       float(group['eps']), float(group['weight_decay']), t + 1, current_stream()), 'msmp_adamw_f32')
                plan['t'] = t + 1
        return loss

    def _plans(self, group, live):
        """[(step count, pointer tables)] of a parameter group, one entry per distinct step count (normally one).  The tables are
        cached and keyed on the STORAGES they point at (parameter, exp_avg, exp_avg_sq data pointers): model.to() / .float(),
        `p.data = ...`, load_state_dict or a user-assigned state tensor rebuild them instead of leaving stale pointers."""
        for p in live:
            if p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise RuntimeError('msmp_pde_amd.optim.AdamW: float32 contiguous CUDA parameters only')
            st = self.state[p]
            if not st:
                st['step'] = torch.zeros((), dtype=torch.float32)
                st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
        key = tuple((p.data_ptr(), self.state[p]['exp_avg'].data_ptr(), self.state[p]['exp_avg_sq'].data_ptr(), id(self.state[p]['step']))
                    for p in live)
        cache = group.get('_msmp_plan')
        if cache is None or cache['key'] != key:
            by_t = {}
            for p in live:
                by_t.setdefault(int(self.state[p]['step'].item()), []).append(p)       # host read-back: only when the tables are rebuilt
            plans = []
            for t, ps in sorted(by_t.items()):
                n = len(ps)
                arr = lambda ts: (ctypes.c_void_p * n)(*[x.data_ptr() for x in ts])
                plans.append({'t': t, 'params': ps, 'steps': [self.state[p]['step'] for p in ps], 'p': arr(ps),
                              'm': arr([self.state[p]['exp_avg'] for p in ps]), 'v': arr([self.state[p]['exp_avg_sq'] for p in ps]),
                              'numel': (ctypes.c_int64 * n)(*[p.numel() for p in ps])})
            cache = group['_msmp_plan'] = {'key': key, 'plans': plans}
        return [(pl['t'], pl) for pl in cache['plans']]

    def state_dict(self):
        sd = super().state_dict()
        for g, live in zip(sd['param_groups'], self.param_groups):
            g.pop('_msmp_plan', None)
            dev = g.pop('_msmp_dev', None)
            if dev is not None:         # the device-side count is the truth of a capturable group: fold it back into the per-parameter entries
                t = float(dev['step'].item())
                for pid in g['params']:
                    if pid in sd['state']:
                        sd['state'][pid] = dict(sd['state'][pid], step=torch.tensor(t, dtype=torch.float32))
        return sd
