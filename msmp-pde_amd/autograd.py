"""Autograd support for the message-passing stack (interim form of SURVEY.md section 8f row 3).

Forward values always come from the HIP kernels (msmp_mp_layer_f32).  The backward pass is a
RECOMPUTE: the layer is re-evaluated from its saved inputs with differentiable PyTorch-ROCm ops on the
GPU and differentiated by torch.autograd; nothing runs on the CPU and nothing here is used by the
inference / rollout path.  Dedicated backward kernels replace the recompute in a later round; the
interface (mp_layer under autograd) stays.  The math below is the same restatement of
experiments/models_gnn.py:61-149 and :1204-1207 as the kernels."""
import torch
import torch.nn.functional as F


def _swish(x):
    return x * torch.sigmoid(x)


def _seg_mean(x, index, n):
    out = torch.zeros(n, x.shape[1], dtype=x.dtype, device=x.device).index_add_(0, index, x)
    cnt = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(
        0, index, torch.ones(index.numel(), dtype=x.dtype, device=x.device))
    return out / cnt.clamp(min=1)[:, None]


def _instance_norm(x, batch, b, eps):
    mean = _seg_mean(x, batch, b)
    xc = x - mean[batch]
    var = _seg_mean(xc * xc, batch, b)
    return xc / torch.sqrt(var + eps)[batch]


def layer_reference(h, u, pos, variables, src, dst, batch, n_graphs, p, lin, eps):
    """One GNN_Layer / GNN_LayerLin with differentiable torch ops; p = (w1,b1,w2,b2,w3,b3,w4,b4)."""
    w1, b1, w2, b2, w3, b3, w4, b4 = p
    cat = torch.cat((h[dst], h[src], u[dst] - u[src], (pos[dst] - pos[src])[:, None], variables[dst]), -1)
    m = _swish(F.linear(_swish(F.linear(cat, w1, b1)), w2, b2))
    agg = _seg_mean(m, dst, h.shape[0])
    upd = F.linear(_swish(F.linear(torch.cat((h, agg, variables), -1), w3, b3)), w4, b4)
    pre = upd if lin else h + _swish(upd)
    return _instance_norm(pre, batch, n_graphs, eps)


class MPLayerFunction(torch.autograd.Function):
    """mp_layer with HIP forward and recompute backward.  Tensor args: h, then the main layer's 8 parameters,
    then (gated) the gate layer's 8 parameters, so autograd routes their gradients."""

    @staticmethod
    def forward(ctx, h, u, pos, variables, structure, main, gate, eps, hip_forward, *params):
        out = hip_forward(h, u, pos, variables, structure, main, gate, eps)
        ctx.save_for_backward(h, u, pos, variables, *params)
        ctx.meta = (structure, main.MODE, gate is not None, eps)
        return out

    @staticmethod
    def backward(ctx, gout):
        h, u, pos, variables, *params = ctx.saved_tensors
        gs, mode, gated, eps = ctx.meta
        src, dst = gs.col[:gs.n_edges].long(), gs.tgt[:gs.n_edges].long()
        sizes = (gs.graph_ptr[1:] - gs.graph_ptr[:-1]).long()
        batch = torch.repeat_interleave(torch.arange(gs.n_graphs, device=h.device), sizes)
        with torch.enable_grad():
            h_ = h.detach().requires_grad_(True)
            ps = [p.detach().requires_grad_(True) for p in params]
            pos1 = pos.reshape(-1)
            if gated:
                tau = torch.sigmoid(layer_reference(h_, u, pos1, variables, src, dst, batch, gs.n_graphs, ps[8:], True, eps))
                out = (1.0 - tau) * h_ + tau * _swish(layer_reference(h_, u, pos1, variables, src, dst, batch, gs.n_graphs,
                                                                      ps[:8], True, eps))
            else:
                out = layer_reference(h_, u, pos1, variables, src, dst, batch, gs.n_graphs, ps, mode == 1, eps)
            grads = torch.autograd.grad(out, [h_] + ps, gout)
        return (grads[0], None, None, None, None, None, None, None, None) + tuple(grads[1:])
