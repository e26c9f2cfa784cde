"""Autograd support for the message-passing stack.

Forward values always come from the HIP kernels (msmp_mp_layer_f32); only the layer's inputs are saved.  The backward
pass RECOMPUTES the layer in materialised form and differentiates it with explicit formulas, all of it behind ONE C-ABI call
per layer / gated pair (msmp_mp_layer_bwd_f32, train_kernels.hip): GEMMs with edge- / node-sized outputs on rocBLAS, everything
else HIP kernels (edge concat, bias + Swish, mean-backward + Swish', InstanceNorm backward, gated-blend backward, scatters,
the batched weight-gradient kernel).  `EXPLICIT_BACKWARD = 1` runs the same algorithm orchestrated from Python (library GEMMs
through torch, ~65 ops per pair), `0` differentiates a PyTorch restatement with torch.autograd (~190 ops): kept for cross-checks.  Nothing runs on
the CPU and nothing here is used by the inference / rollout path.  The math is that of experiments/models_gnn.py:61-149
and :1204-1207."""
import torch
import torch.nn.functional as F

from ._lib import lib, check, ptr, current_stream

# 2: the whole layer backward behind the C-ABI (msmp_mp_layer_bwd_f32, launches issued from native code);
# 1: the same algorithm orchestrated from Python (library GEMMs through torch); 0: torch.autograd over a restatement.
EXPLICIT_BACKWARD = 2

_bwd_ws = {}


def layer_backward_native(gout, h, u, pos, variables, gs, params, mode_lin, gated, eps):
    """dL/dh and the parameter gradients of one layer / gated pair through msmp_mp_layer_bwd_f32."""
    import ctypes
    L = lib()
    dev = h.device
    n, e, tw, nv = h.shape[0], gs.n_edges, u.shape[1], variables.shape[1]
    ps = [q.detach() if (q.dtype == torch.float32 and q.is_contiguous()) else q.detach().to(torch.float32).contiguous() for q in params]
    grads = [torch.empty(q.shape, dtype=torch.float32, device=dev) for q in ps]
    dh = torch.empty(n, h.shape[1], dtype=torch.float32, device=dev)
    g = gout if (gout.dtype == torch.float32 and gout.is_contiguous()) else gout.to(torch.float32).contiguous()
    need = L.msmp_mp_layer_bwd_workspace_bytes(n, e, tw, nv, int(gated))
    ws = _bwd_ws.get(dev)                       # grow-only scratch per device, like the forward's
    if ws is None or ws.numel() < need:
        ws = _bwd_ws[dev] = torch.empty(need, dtype=torch.uint8, device=dev)
    arr = lambda ts: (ctypes.c_void_p * 8)(*[t.data_ptr() for t in ts])
    # edges regrouped by source: the source-side scatter of dL/dh runs in a fixed order (an edgeless batch has nothing to scatter)
    perm32, src_rowptr = gs.by_source32() if e else (None, None)
    check(L.msmp_mp_layer_bwd_f32(ptr(g), ptr(h), ptr(u), ptr(pos), ptr(variables), ptr(gs.rowptr), ptr(gs.col), ptr(gs.tgt),
                                  ptr(src_rowptr), ptr(perm32), ptr(gs.graph_ptr), n, e, gs.n_graphs, tw, nv, arr(ps[:8]), arr(ps[8:]) if gated else None,
                                  1 if (mode_lin or gated) else 0, eps, ptr(dh), arr(grads[:8]), arr(grads[8:]) if gated else None,
                                  ptr(ws), ws.numel(), current_stream()), 'msmp_mp_layer_bwd_f32')
    return dh, grads


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


def _head_recompute(h, u, pos, variables, gs, p):
    """One GNN_LayerLin / GNN_Layer up to its pre-norm update  W4 Swish(W3 [h, mean_j m_ij, var] + b3) + b4, keeping the
    pre-activations.  p = (w1, b1, w2, b2, w3, b3, w4, b4)."""
    L = lib()
    w1, b1, w2, b2, w3, b3, w4, b4 = p
    n, e, tw, nv = h.shape[0], gs.n_edges, u.shape[1], variables.shape[1]
    k = 2 * h.shape[1] + tw + 1 + nv
    ld = (k + 3) // 4 * 4
    cat_e = torch.empty(e, ld, dtype=torch.float32, device=h.device)
    if e:
        check(L.msmp_edge_concat_f32(ptr(h), ptr(u), ptr(pos), ptr(variables), ptr(gs.tgt), ptr(gs.col), e, tw, nv, ld, ptr(cat_e),
                                     current_stream()), 'msmp_edge_concat_f32')
    cat_e = cat_e[:, :k]
    a1 = torch.addmm(b1, cat_e, w1.t())
    m1 = F.silu(a1)
    a2 = torch.addmm(b2, m1, w2.t())
    m2 = F.silu(a2)
    agg = torch.empty_like(h) if e else torch.zeros_like(h)      # no in-edges anywhere: every mean is 0
    if e:
        check(L.msmp_scatter_mean_f32(ptr(m2), ptr(gs.rowptr), n, ptr(agg), current_stream()), 'msmp_scatter_mean_f32')
    cat_n = torch.cat((h, agg, variables), 1)
    a3 = torch.addmm(b3, cat_n, w3.t())
    u1 = F.silu(a3)
    upd = torch.addmm(b4, u1, w4.t())
    return upd, (cat_e, a1, m1, a2, cat_n, a3, u1)


def grad_weights(pairs):
    """[(A [R,128] = dL/d pre-activation, B [R,k2] = the linear layer's input; row-strided views are fine)] -> [dW [128,k2], db [128], ...] through
    msmp_grad_weights_f32 (all pairs in one call; row-split fp32-exact MFMA partials, deterministic).  A pair may be a triple
    (A, B1, B2): B is then the column concatenation [B1 | B2] WITHOUT being materialised (msmp_grad_weights_cat_f32)."""
    import ctypes
    L = lib()
    n = len(pairs)
    fix = lambda t: t if t.stride(1) == 1 else t.contiguous()
    a = [fix(p[0]) for p in pairs]
    b = [fix(p[1]) for p in pairs]
    b2 = [fix(p[2]) if len(p) > 2 else None for p in pairs]
    kk = [b[i].shape[1] + (b2[i].shape[1] if b2[i] is not None else 0) for i in range(n)]
    rows = (ctypes.c_int64 * n)(*[x.shape[0] for x in a])
    k2 = (ctypes.c_int * n)(*kk)
    lda = (ctypes.c_int * n)(*[x.stride(0) for x in a])
    ldb = (ctypes.c_int * n)(*[y.stride(0) for y in b])
    out_w = [torch.empty(a[i].shape[1], kk[i], dtype=torch.float32, device=a[i].device) for i in range(n)]
    out_b = [torch.empty(a[i].shape[1], dtype=torch.float32, device=a[i].device) for i in range(n)]
    ws_floats = L.msmp_grad_weights_workspace_floats(n, rows, k2)
    if ws_floats < 0:
        raise ValueError('grad_weights: unsupported shapes')
    ws = torch.empty(ws_floats, dtype=torch.float32, device=a[0].device)
    vp = lambda ts: (ctypes.c_void_p * n)(*[t.data_ptr() if t is not None else None for t in ts])
    if any(t is not None for t in b2):
        ldb2 = (ctypes.c_int * n)(*[t.stride(0) if t is not None else 0 for t in b2])
        ksplit = (ctypes.c_int * n)(*[b[i].shape[1] if b2[i] is not None else 0 for i in range(n)])
        check(L.msmp_grad_weights_cat_f32(n, vp(a), vp(b), vp(b2), rows, lda, ldb, ldb2, ksplit, k2, vp(out_w), vp(out_b), ptr(ws), ws_floats,
                                          current_stream()), 'msmp_grad_weights_cat_f32')
    else:
        check(L.msmp_grad_weights_f32(n, vp(a), vp(b), rows, lda, ldb, k2, vp(out_w), vp(out_b), ptr(ws), ws_floats, current_stream()),
              'msmp_grad_weights_f32')
    res = []
    for w, bias in zip(out_w, out_b):
        res += [w, bias]
    return res


def _head_backward(d_upd, saved, gs, p, dh):
    """Back-propagates dL/d(pre-norm update) of one head to dL/dh (accumulated into dh) and returns the four
    (dL/d pre-activation, layer input) pairs of its linear layers, in parameter order, for grad_weights."""
    L = lib()
    cat_e, a1, m1, a2, cat_n, a3, u1 = saved
    w1, _, w2, _, w3, _, w4, _ = p
    hd = dh.shape[1]
    silu_backward = torch.ops.aten.silu_backward
    d_a3 = silu_backward(d_upd @ w4, a3)
    d_cat_n = d_a3 @ w3[:, :2 * hd]                       # [dh | dagg]; the variables columns carry no gradient
    dh += d_cat_n[:, :hd]
    dagg = d_cat_n[:, hd:].contiguous()
    d_a2 = torch.empty_like(a2)
    if gs.n_edges:
        check(L.msmp_mean_bwd_dswish_f32(ptr(dagg), ptr(gs.rowptr), ptr(gs.tgt), ptr(a2), gs.n_edges, ptr(d_a2), current_stream()),
              'msmp_mean_bwd_dswish_f32')
    d_a1 = silu_backward(d_a2 @ w2, a1)
    d_cat_e = d_a1 @ w1[:, :2 * hd]                       # [d x_i | d x_j]
    dh.index_add_(0, gs.tgt_long, d_cat_e[:, :hd])
    dh.index_add_(0, gs.col_long, d_cat_e[:, hd:])
    return [(d_a1, cat_e), (d_a2, m1), (d_a3, cat_n), (d_upd, u1)]


def _param_grads(pairs, n_edges):
    if n_edges:
        return grad_weights(pairs)
    res = []                                              # no edges: the message layers get zero gradients
    for k, (a, b) in enumerate(pairs):
        if a.shape[0]:
            res += grad_weights([(a, b)])
        else:
            res += [a.new_zeros(a.shape[1], b.shape[1]), a.new_zeros(a.shape[1])]
    return res


def layer_backward_explicit(gout, h, u, pos, variables, gs, params, mode_lin, gated, eps):
    """dL/dh and dL/d params of one layer (or one gated pair) given gout = dL/d out; see the module docstring."""
    L = lib()
    st = current_stream()
    gout = gout.to(torch.float32).contiguous()
    pos1 = pos.reshape(-1)
    params = [q.detach() for q in params]
    if gated:
        upd_m, sv_m = _head_recompute(h, u, pos1, variables, gs, params[:8])
        upd_g, sv_g = _head_recompute(h, u, pos1, variables, gs, params[8:])
        d_g, d_m, dh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
        check(L.msmp_gate_blend_bwd_f32(ptr(gout), ptr(h), ptr(upd_g), ptr(upd_m), ptr(gs.graph_ptr), gs.n_graphs, eps, ptr(d_g),
                                        ptr(d_m), ptr(dh), st), 'msmp_gate_blend_bwd_f32')
        pairs = _head_backward(d_m, sv_m, gs, params[:8], dh) + _head_backward(d_g, sv_g, gs, params[8:], dh)
        return dh, _param_grads(pairs, gs.n_edges)
    upd, sv = _head_recompute(h, u, pos1, variables, gs, params)
    pre = upd if mode_lin else h + F.silu(upd)
    dh = torch.empty_like(h)
    check(L.msmp_instance_norm_bwd_f32(ptr(pre), ptr(gout), ptr(gs.graph_ptr), gs.n_graphs, eps, ptr(dh), st),
          'msmp_instance_norm_bwd_f32')
    if mode_lin:                                          # out = IN(upd): no direct path to h
        d_upd, dh = dh, torch.zeros_like(h)
    else:                                                 # out = IN(h + Swish(upd))
        d_upd = torch.ops.aten.silu_backward(dh, upd)
    return dh, _param_grads(_head_backward(d_upd, sv, gs, params, dh), gs.n_edges)


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
        if EXPLICIT_BACKWARD:
            with torch.no_grad():
                fn = layer_backward_native if EXPLICIT_BACKWARD == 2 else layer_backward_explicit
                dh, grads = fn(gout, h, u, pos.reshape(-1), variables, gs, params, mode == 1, gated, eps)
            return (dh, None, None, None, None, None, None, None, None) + tuple(grads)
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
