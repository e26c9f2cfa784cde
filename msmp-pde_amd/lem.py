"""LEM ("Long Expressive Memory") node encoder in PyTorch-ROCm.

The reference reaches an absent CUDA extension here (`import lem_cuda`, experiments/models_gnn.py:8,
285-342); BASELINE.json's north_star keeps the encoder in PyTorch, so this is a pure-torch
restatement of the recurrence with the reference's parameter names (`rnn.weights` [3H, ninp+H],
`rnn.weights_lin_z` [H, ninp+H], `rnn.bias` [3H], `rnn.bias_lin_z` [H]) and init
U(-1/sqrt(H), 1/sqrt(H)) (:318-321), dt = 1 (:334).  PARITY UNPINNED: the column order inside
lem_cuda cannot be confirmed offline; it follows the published LEM cell:
    X = [y, x_t]; g = X W^T + b -> (g1, g2, g3); dt_bar = dt s(g1); dt_ = dt s(g2)
    z <- (1-dt_) z + dt_ tanh(g3);  y <- (1-dt_bar) y + dt_bar tanh([z, x_t] Wz^T + bz)
"""
import math

import torch
from torch import nn

from ._lib import lib, check, ptr, current_stream, MSMP_ERR_UNSUPPORTED, PARAM_EPOCH


class LEMcuda(nn.Module):
    """Parameter holder named like the reference's (experiments/models_gnn.py:305-330)."""

    def __init__(self, ninp, nhid, dt):
        super().__init__()
        self.ninp, self.nhid, self.dt = ninp, nhid, float(dt)
        self.weights = nn.Parameter(torch.empty(3 * nhid, ninp + nhid, dtype=torch.float32))
        self.weights_lin_z = nn.Parameter(torch.empty(nhid, ninp + nhid, dtype=torch.float32))
        self.bias = nn.Parameter(torch.empty(3 * nhid, dtype=torch.float32))
        self.bias_lin_z = nn.Parameter(torch.empty(nhid, dtype=torch.float32))
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1.0 / math.sqrt(self.nhid)
        for w in self.parameters():
            w.data.uniform_(-stdv, +stdv)

    def forward(self, inputs, states=None, return_state=False):
        """inputs [T, N, ninp] -> final y [N, nhid] (and z with return_state); `states` = (y0, z0) as in the reference
        (models_gnn.py:325-332), default zeros."""
        t_len, n, _ = inputs.shape
        nh = self.nhid
        y, z = (inputs.new_zeros(n, nh), inputs.new_zeros(n, nh)) if states is None else states
        wy, wx = self.weights[:, :nh].t().contiguous(), self.weights[:, nh:].t().contiguous()
        zy, zx = self.weights_lin_z[:, :nh].t().contiguous(), self.weights_lin_z[:, nh:].t().contiguous()
        # input projections for all steps at once (the recurrent part stays sequential)
        gx = torch.matmul(inputs, wx) + self.bias          # [T, N, 3H]
        lx = torch.matmul(inputs, zx) + self.bias_lin_z    # [T, N, H]
        if inputs.is_cuda and inputs.dtype == torch.float32 and not torch.is_grad_enabled():
            # inference at any width (the GLU classes): the two recurrent GEMMs stay library calls, the pointwise work between them is one
            # HIP launch each (msmp_wide_lem_z_f32 / _y_f32) instead of ~20 elementwise kernels per time step
            L = lib()
            y, z = y.contiguous().clone(), z.contiguous().clone()
            dt_bar = torch.empty_like(y)
            for t in range(t_len):
                g = torch.addmm(gx[t], y, wy)
                check(L.msmp_wide_lem_z_f32(ptr(g), n, nh, float(self.dt), ptr(z), ptr(dt_bar), current_stream()), 'msmp_wide_lem_z_f32')
                lin = torch.addmm(lx[t], z, zy)
                check(L.msmp_wide_lem_y_f32(ptr(lin), ptr(dt_bar), n * nh, ptr(y), current_stream()), 'msmp_wide_lem_y_f32')
            return (y, z) if return_state else y
        for t in range(t_len):
            g = torch.addmm(gx[t], y, wy)
            dt_bar = self.dt * torch.sigmoid(g[:, :nh])
            dt_ = self.dt * torch.sigmoid(g[:, nh:2 * nh])
            z = (1.0 - dt_) * z + dt_ * torch.tanh(g[:, 2 * nh:])
            y = (1.0 - dt_bar) * y + dt_bar * torch.tanh(torch.addmm(lx[t], z, zy))
        return (y, z) if return_state else y


def _padded_inputs(xin):
    """xin [N, T, ninp] -> float32, rows zero-padded to msmp_lem_input_stride(ninp), contiguous."""
    ninp = xin.shape[2]
    stride = lib().msmp_lem_input_stride(ninp)
    x = xin.detach().to(torch.float32)
    return (torch.nn.functional.pad(x, (0, stride - ninp)) if stride != ninp else x).contiguous()


def _state(t):
    return None if t is None else t.detach().to(torch.float32).contiguous()


def _train_forward(owner, x, ninp, y0, z0, saved):
    """msmp_lem_train_fwd_f32: the exact-fp32 recurrence from the states (y0, z0) (None = zeros) -> (y_T, z_T)."""
    n, t_len = x.shape[0], x.shape[1]
    out = torch.empty(n, owner.nhid, dtype=torch.float32, device=x.device)
    z_out = torch.empty_like(out)
    check(lib().msmp_lem_train_fwd_f32(ptr(x), n, t_len, ninp, owner.rnn.dt, ptr(owner._pack(None)), ptr(y0), ptr(z0), ptr(saved),
                                       ptr(out), ptr(z_out), current_stream()), 'msmp_lem_train_fwd_f32')
    return out, z_out


class _LEMTrainFunction(torch.autograd.Function):
    """LEMFunction of the reference (experiments/models_gnn.py:285-302) on the HIP training kernels: forward =
    msmp_lem_train_fwd_f32 (saves the per-step activations), backward = msmp_lem_train_bwd_f32 (BPTT, one launch) followed
    by the weight-gradient GEMMs over the N*T rows.  Like the reference it returns no gradient for the step inputs."""

    @staticmethod
    def forward(ctx, owner, xin, weights, weights_lin_z, bias, bias_lin_z, y0=None, z0=None):
        x = _padded_inputs(xin)
        n, t_len, ninp = xin.shape
        saved = torch.empty(lib().msmp_lem_saved_floats(n, t_len), dtype=torch.float32, device=x.device)
        y0, z0 = _state(y0), _state(z0)
        out, z_out = _train_forward(owner, x, ninp, y0, z0, saved)
        ctx.save_for_backward(x, saved, weights, weights_lin_z, *([y0, z0] if y0 is not None else []))
        ctx.dt, ctx.ninp = owner.rnn.dt, ninp
        ctx.mark_non_differentiable(z_out)          # the carried state: a constant for the next call (models_gnn.py:350-353)
        return out, z_out

    @staticmethod
    def backward(ctx, grad_y, _grad_z=None):
        L = lib()
        x, saved, weights, weights_lin_z, *states = ctx.saved_tensors
        y0, z0 = states if states else (None, None)
        n, t_len, stride = x.shape
        nh = weights_lin_z.shape[0]
        blob = torch.empty(L.msmp_packed_lem_bwd_floats(), dtype=torch.float32, device=x.device)
        w, wz = (p.detach().to(torch.float32).contiguous() for p in (weights, weights_lin_z))
        check(L.msmp_pack_lem_bwd_f32(ptr(w), ptr(wz), ctx.ninp, ptr(blob), current_stream()), 'msmp_pack_lem_bwd_f32')
        dg = torch.empty(n * t_len, 4 * nh, dtype=torch.float32, device=x.device)
        g = grad_y.to(torch.float32).contiguous()
        check(L.msmp_lem_train_bwd_f32(ptr(g), ptr(saved), ptr(y0), ptr(z0), n, t_len, ctx.dt, ptr(blob), ptr(dg), current_stream()),
              'msmp_lem_train_bwd_f32')
        planes = saved.view(6, n, t_len, nh)
        xs = x.view(n * t_len, stride)[:, :ctx.ninp]
        y_prev = planes[4].reshape(n * t_len, nh)            # the forward saves y_{t-1} (y0 at t = 0): the rows of [y_{t-1} ; x_t]
        z_new = planes[5].reshape(n * t_len, nh)
        from .autograd import grad_weights          # weight / bias gradients: sums over the N*T rows; [state | x_t] is never materialised
        g = grad_weights([(dg[:, :nh], y_prev, xs), (dg[:, nh:2 * nh], y_prev, xs), (dg[:, 2 * nh:3 * nh], y_prev, xs), (dg[:, 3 * nh:], z_new, xs)])
        d_w, d_b = torch.cat(g[0:6:2], 0), torch.cat(g[1:6:2], 0)
        return None, None, d_w.to(weights.dtype), g[6].to(weights_lin_z.dtype), d_b, g[7], None, None


class LEM(nn.Module):
    """experiments/models_gnn.py:333-342: returns all_y[-1].

    `forward` is the differentiable path: the HIP training kernels (_LEMTrainFunction).  There is NO CPU path: host tensors
    raise.  TRAIN_KERNELS = False is a validation aid only (scripts/train_soak.py, scripts/train_step_time.py): it routes GPU
    tensors through the PyTorch-ROCm restatement `LEMcuda.forward`, which the tests also call directly in float64 as the
    reference of the kernels.  `encode` /
    `encode_nodes` are the product path for inference: the fused HIP kernel (recurrence + lemoutput_mlp in one launch,
    states in registers)."""
    TRAIN_KERNELS = True

    def __init__(self, ninp, nhid, dt=1.):
        super().__init__()
        self.ninp, self.nhid = ninp, nhid
        self.rnn = LEMcuda(ninp, nhid, dt)
        self._packed = None
        self._packed_key = None

    def forward(self, inputs):
        """inputs [T, N, ninp] (the reference's layout) -> all_y[-1] [N, nhid]."""
        return self.forward_nodes(inputs.permute(1, 0, 2))

    def forward_nodes(self, xin):
        """Same with node-major step inputs xin [N, T, ninp] (the layout the kernels read)."""
        if not xin.is_cuda:
            raise RuntimeError('LEM needs CUDA tensors; there is no CPU fallback')
        if self.nhid != 128:            # the GLU classes (164 hidden units): the PyTorch-ROCm restatement of the cell (north_star keeps the encoder in PyTorch)
            return self.rnn(xin.permute(1, 0, 2).contiguous().to(self.rnn.weights.dtype))
        if not self.TRAIN_KERNELS:
            return self.rnn(xin.permute(1, 0, 2).contiguous())
        r = self.rnn
        return _LEMTrainFunction.apply(self, xin, r.weights, r.weights_lin_z, r.bias, r.bias_lin_z)[0]

    def _pack(self, mlp):
        ps = [self.rnn.weights, self.rnn.weights_lin_z, self.rnn.bias, self.rnn.bias_lin_z]
        if mlp is not None:
            ps += [mlp[0].weight, mlp[0].bias, mlp[2].weight, mlp[2].bias]
        key = (PARAM_EPOCH[0],) + tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)
        if key != self._packed_key:
            L = lib()
            blob = torch.empty(L.msmp_packed_lem_floats(), dtype=torch.float32, device=ps[0].device)
            f = [p.detach().to(torch.float32).contiguous() for p in ps]
            args = [ptr(t) for t in f] + [None] * (8 - len(f))
            check(L.msmp_pack_lem_f32(*args, self.ninp, ptr(blob), current_stream()), 'msmp_pack_lem_f32')
            self._packed, self._packed_key = blob, key
        return self._packed

    def encode_nodes(self, u, pos_x, pos_t, variables, dt_cum, two_d, mlp=None):
        """Same as `encode` with the step inputs assembled inside the kernel from the node arrays (models_gnn.py:1357-1360 /
        models_gnn2D.py:429-433).  Returns None when the selected kernel edition has no such entry (msmp_tune "lem" != 3)."""
        L = lib()
        n, nv = u.shape[0], variables.shape[1]
        tw = u.shape[1] // (2 if two_d else 1)
        assert self.nhid == 128 and self.ninp == (3 if two_d else 2) + nv
        t = [x.to(torch.float32).contiguous() for x in (u, pos_x, pos_t, variables, dt_cum)]
        out = torch.empty(n, self.nhid, dtype=torch.float32, device=u.device)
        rc = L.msmp_lem_encoder_nodes_f32(ptr(t[0]), ptr(t[1]), ptr(t[2]), ptr(t[3]), ptr(t[4]), n, tw, nv, int(two_d), self.rnn.dt,
                                          ptr(self._pack(mlp)), int(mlp is not None), ptr(out), current_stream())
        if rc == MSMP_ERR_UNSUPPORTED:
            return None
        check(rc, 'msmp_lem_encoder_nodes_f32')
        return out

    def encode(self, xin, mlp=None):
        """xin [N, T, ninp] float32 CUDA (node-major step inputs) -> [N, nhid]; `mlp` = lemoutput_mlp
        (nn.Sequential(Linear, Swish, Linear, Swish)) to fuse behind the recurrence."""
        assert self.nhid == 128 and xin.shape[2] == self.ninp
        L = lib()
        n, t_len, _ = xin.shape
        stride = L.msmp_lem_input_stride(self.ninp)
        if stride != self.ninp:
            xin = torch.nn.functional.pad(xin, (0, stride - self.ninp))
        xin = xin.to(torch.float32).contiguous()
        out = torch.empty(n, self.nhid, dtype=torch.float32, device=xin.device)
        check(L.msmp_lem_encoder_f32(ptr(xin), n, t_len, self.ninp, self.rnn.dt, ptr(self._pack(mlp)),
                                     int(mlp is not None), ptr(out), current_stream()), 'msmp_lem_encoder_f32')
        return out


class LEMS(LEM):
    """experiments/models_gnn.py:345-362: the LEM that keeps (all_y[-1], all_z[-1]) of a call as the initial states of the next
    (`reset_states()` starts a new unrolling sequence; train_helper.py:144-145, 199-200).  The carried states are constants for
    the next call.  Runs on the exact-fp32 recurrence kernel (msmp_lem_train_fwd_f32; with autograd its BPTT pair), which takes
    initial states; the weight-stationary inference kernel starts from zeros and is not used here."""

    def __init__(self, ninp, nhid, dt=1.):
        super().__init__(ninp, nhid, dt)
        self.states = None

    def reset_states(self):
        self.states = None

    def forward(self, inputs):
        return self.forward_nodes(inputs.permute(1, 0, 2))

    def forward_nodes(self, xin):
        y0, z0 = self.states if self.states is not None else (None, None)
        r = self.rnn
        if not (xin.is_cuda and self.nhid == 128):
            raise RuntimeError('LEMS needs CUDA tensors (HIP path only, no CPU fallback)')
        if torch.is_grad_enabled() and any(p.requires_grad for p in r.parameters()):
            if self.TRAIN_KERNELS:
                y, z = _LEMTrainFunction.apply(self, xin, r.weights, r.weights_lin_z, r.bias, r.bias_lin_z, y0, z0)
            else:
                y, z = r(xin.permute(1, 0, 2).contiguous(), None if y0 is None else (y0, z0), return_state=True)
        else:
            y, z = _train_forward(self, _padded_inputs(xin), xin.shape[2], _state(y0), _state(z0), None)
        self.states = (y.detach(), z.detach())
        return y

    def encode(self, *args, **kwargs):
        raise RuntimeError('LEMS is stateful: use forward / forward_nodes')

    encode_nodes = encode
