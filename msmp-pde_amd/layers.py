"""Message-passing layers with the reference's module / parameter names, running on the HIP kernels.

Reference: experiments/models_gnn.py:12-21 (Swish), 23-86 (GNN_Layer), 88-149 (GNN_LayerLin).
state_dict keys are identical (`message_net_1.0.weight` [128, 2*128+tw+1+nv], `message_net_2.0.*`,
`update_net_1.0.weight` [128, 2*128+nv], `update_net_2.0.*`; InstanceNorm has no parameters), so
reference checkpoints load unchanged.  Parameters are created in float32 whatever the default dtype
is (the reference's import side effect makes it float64, temporal/solvers.py:10).
"""
import ctypes
import threading

import torch
from torch import nn

from . import _lib
from ._lib import lib, check, ptr, current_stream, HIDDEN
from .graph import GraphStructure


class Swish(nn.Module):
    """x * sigmoid(beta x); experiments/models_gnn.py:12-21."""

    def __init__(self, beta=1):
        super().__init__()
        self.beta = beta

    def forward(self, x):
        return x * torch.sigmoid(self.beta * x)


def _linear(i, o):
    return nn.Linear(i, o, dtype=torch.float32)


class _Workspace(object):
    """Grow-only scratch buffer per (device, stream), shared by all layers (the C-ABI never allocates; forwards issued on different
    streams of one device must not share scratch memory).  `private(device)` is a context
    in which the layers of the calling thread use buffers of their own instead (hipGraph capture: the captured launches must
    point at memory nobody replaces; solvers._GraphedForward keeps those buffers alive as long as the graph)."""
    _bufs = {}
    _tls = threading.local()      # `.private`: the calling THREAD's private buffer (a capture on one thread must not redirect another thread's forward)

    class _Private(object):
        def __init__(self, device):
            self.device, self.buf = device, None

        def __enter__(self):
            self._outer = getattr(_Workspace._tls, 'private', None)
            _Workspace._tls.private = self
            return self

        def __exit__(self, *exc):
            _Workspace._tls.private = self._outer

        def buffers(self):
            return [self.buf]

    @classmethod
    def private(cls, device):
        return cls._Private(device)

    @classmethod
    def get(cls, nbytes, device):
        p = getattr(cls._tls, 'private', None)
        if p is not None and p.device == device:
            if p.buf is None or p.buf.numel() < nbytes:
                assert not torch.cuda.is_current_stream_capturing() or p.buf is None, 'workspace grew during capture'
                p.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            return p.buf
        key = (device, torch.cuda.current_stream(device).cuda_stream)
        buf = cls._bufs.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
            cls._bufs[key] = buf
        return buf


class _MPLayerBase(nn.Module):
    MODE = None

    def __init__(self, in_features, out_features, hidden_features, time_window, n_variables):
        super().__init__()
        if not (in_features == out_features == hidden_features):
            raise ValueError('in_features, hidden_features and out_features must be equal (they are in every class of the reference)')
        # hidden width 128: the fused kernels.  Any other width (the GLU classes: 164): the width-generic layer of wide_kernels.hip
        # around msmp_linear_f32 (same formulas, HBM-bound pieces; `mp_layer` dispatches on this flag)
        self.wide = hidden_features != HIDDEN
        if not 1 <= n_variables <= _lib.MSMP_MAX_VARS:
            raise ValueError(f'n_variables must be in 1..{_lib.MSMP_MAX_VARS}')
        self.in_features, self.out_features, self.hidden_features = in_features, out_features, hidden_features
        self.time_window, self.n_variables = time_window, n_variables
        self.message_net_1 = nn.Sequential(_linear(2 * in_features + time_window + 1 + n_variables, hidden_features), Swish())
        self.message_net_2 = nn.Sequential(_linear(hidden_features, hidden_features), Swish())
        self.update_net_1 = nn.Sequential(_linear(in_features + hidden_features + n_variables, hidden_features), Swish())
        self._make_update_net_2(hidden_features, out_features)
        self._packed = None
        self._packed_key = None
        self._ps = None

    def _params8(self):
        ps = self._ps
        if ps is None or ps[0] is not self.message_net_1[0].weight:      # (re-)collected when a Parameter object was replaced
            ps = self._ps = (self.message_net_1[0].weight, self.message_net_1[0].bias, self.message_net_2[0].weight,
                             self.message_net_2[0].bias, self.update_net_1[0].weight, self.update_net_1[0].bias,
                             self.update_net_2[0].weight, self.update_net_2[0].bias)
        return ps

    def packed(self):
        """Kernel-layout weight blob (msmp_pack_layer_f32), re-packed only when a parameter changed."""
        if self.wide:
            raise _lib.MsmpError(f'the packed layer blob exists for hidden width {HIDDEN} only')
        ps = self._params8()
        key = (_lib.PARAM_EPOCH[0], ps[0].data_ptr(), ps[7].data_ptr(), ps[0].dtype) + tuple(p._version for p in ps)
        if key != self._packed_key:
            dev = ps[0].device
            if dev.type != 'cuda' or any(p.device != dev for p in ps):
                raise _lib.MsmpError('layer parameters must be on the GPU (HIP path only, no CPU fallback)')
            L = lib()
            n = L.msmp_packed_layer_floats(self.time_window, self.n_variables)
            blob = torch.empty(n, dtype=torch.float32, device=dev)
            f = [p.detach().to(torch.float32).contiguous() for p in ps]
            check(L.msmp_pack_layer_f32(*[ptr(t) for t in f], self.time_window, self.n_variables, ptr(blob),
                                        current_stream()), 'msmp_pack_layer_f32')
            self._packed, self._packed_key = blob, key
        return self._packed

    def wide_weights(self):
        """The wide path's operands, cached per parameter version: message_net_1 factorised per node (models_gnn.py:132-138):
        P = Wp [h | u | pos | vars] + b1 for the edge's target, Q = Wq [h | u | pos | vars] for its source, with
        Wp = [W1[:, :W] | W1[:, 2W:]], Wq = [W1[:, W:2W] | -W1[:, 2W:2W+tw+1] | 0]."""
        ps = self._params8()
        key = (_lib.PARAM_EPOCH[0],) + tuple((p.data_ptr(), p._version) for p in ps)
        if key != self._packed_key:
            w1 = ps[0].detach().to(torch.float32)
            W, tw = self.hidden_features, self.time_window
            wp = torch.cat((w1[:, :W], w1[:, 2 * W:]), 1).contiguous()
            wq = torch.cat((w1[:, W:2 * W], -w1[:, 2 * W:2 * W + tw + 1], torch.zeros_like(w1[:, 2 * W + tw + 1:])), 1).contiguous()
            self._packed = (wp, wq) + tuple(p.detach().to(torch.float32).contiguous() for p in ps[1:])
            self._packed_key = key
        return self._packed

    def forward(self, x, u, pos, variables, edge_index, batch, structure=None):
        """Same signature as the reference's layer forward (experiments/models_gnn.py:61-67 / 124-130);
        `structure` lets the solver pass the cached CSR instead of rebuilding it from edge_index."""
        if structure is None:
            structure = GraphStructure(edge_index, batch, x.shape[0])
        return mp_layer(x, u, pos, variables, structure, self, None)


class GNN_Layer(_MPLayerBase):
    """experiments/models_gnn.py:23-86: Swish on the last linear and residual x + update."""
    MODE = _lib.MSMP_LAYER_RESIDUAL_SWISH

    def _make_update_net_2(self, h, o):
        self.update_net_2 = nn.Sequential(_linear(h, o), Swish())


class GNN_LayerLin(_MPLayerBase):
    """experiments/models_gnn.py:88-149: no final activation, no residual."""
    MODE = _lib.MSMP_LAYER_LIN

    def _make_update_net_2(self, h, o):
        self.update_net_2 = nn.Sequential(_linear(h, o))


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


DENSE_MESSAGE = False     # True: evaluate message_net_1 on the per-edge concatenation (reference order of operations)


def node_features(u, pos_x, variables):
    """[u | pos | vars | 0-pad] rows of msmp_pack_node_features_f32: the columns of message_net_1's input that are the same for
    every layer of a forward.  Packed once per forward by the solvers and handed to every layer call (`feat`)."""
    L = lib()
    n, tw, nv = u.shape[0], u.shape[1], variables.shape[1]
    stride = L.msmp_node_feature_stride(tw, nv)
    feat = torch.empty(n, stride, dtype=torch.float32, device=u.device)
    check(L.msmp_pack_node_features_f32(ptr(u), ptr(pos_x), ptr(variables), n, tw, nv, ptr(feat), current_stream()),
          'msmp_pack_node_features_f32')
    return feat


def _mp_layer_hip(h, u, pos_x, variables, gs, main, gate, eps, dense_message=None, feat=None, decode=None):
    """The HIP call proper (no autograd): msmp_mp_layer_f32.  decode = (conv1, conv2, u_or_None, dt, tw): the LAST layer of a 1-D solver
    with the decoder as the node tail's epilogue (msmp_mp_layer_decode_f32): returns (h_out, prediction), or None where the fused
    tail does not apply (the caller then takes the two entry points)."""
    L = lib()
    n = h.shape[0]
    out = torch.empty_like(h)
    gated = gate is not None
    dense = DENSE_MESSAGE if dense_message is None else dense_message
    mode = main.MODE | (_lib.MSMP_LAYER_DENSE_MESSAGE if dense else 0)
    ws_bytes = L.msmp_mp_layer_workspace_bytes(n, gs.n_edges, int(gated), gs.max_in_degree)
    ws = _Workspace.get(ws_bytes, h.device)
    tiles = gs.tiles()
    if decode is not None:
        c1, c2, u_last, dt, tw = decode
        w = [p.detach().to(torch.float32).contiguous() for p in (c1.weight, c1.bias, c2.weight, c2.bias)]       # alive until the call returns (stream-ordered use follows)
        pred = torch.empty(n, tw, dtype=torch.float32, device=h.device)
        dec = _lib.MsmpDecoder(ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(w[3]), ptr(u_last), float(dt), int(tw), ptr(pred))
        rc = L.msmp_mp_layer_decode_f32(ptr(h), ptr(u), ptr(pos_x), ptr(variables), ptr(feat), ptr(gs.rowptr), ptr(gs.col), ptr(gs.tgt),
                                        None if tiles is None else ctypes.byref(tiles[0]), ptr(gs.graph_ptr), n, gs.n_edges, gs.n_graphs, gs.max_in_degree,
                                        gs.max_graph_nodes, main.time_window, main.n_variables, ptr(main.packed()), ptr(gate.packed()) if gated else None,
                                        mode, eps, ptr(out), ctypes.byref(dec), ptr(ws), ws.numel(), current_stream())
        if rc == _lib.MSMP_ERR_UNSUPPORTED:
            return None
        check(rc, 'msmp_mp_layer_decode_f32')
        out._msmp_keep = w          # the kernels read the decoder weights after this returns
        return out, pred
    check(L.msmp_mp_layer_f32(ptr(h), ptr(u), ptr(pos_x), ptr(variables), ptr(feat), ptr(gs.rowptr), ptr(gs.col), ptr(gs.tgt),
                              None if tiles is None else ctypes.byref(tiles[0]), ptr(gs.graph_ptr), n, gs.n_edges, gs.n_graphs, gs.max_in_degree, gs.max_graph_nodes, main.time_window,
                              main.n_variables,
                              ptr(main.packed()), ptr(gate.packed()) if gated else None, mode, eps, ptr(out),
                              ptr(ws), ws.numel(), current_stream()), 'msmp_mp_layer_f32')
    return out


def _wide_linear(x, k, w, bias, n_out, mode, out, ws):
    L = lib()
    check(L.msmp_linear_f32(ptr(x), x.shape[1], x.shape[0], k, ptr(w), w.shape[1], ptr(bias), n_out, mode, ptr(out), out.shape[1],
                            ptr(ws), ws.numel(), current_stream()), 'msmp_linear_f32')


def _wide_head(h, feat_cat, k_feat, variables, gs, layer, ld, ws):
    """One GNN_LayerLin head at hidden width W != 128 up to its pre-norm output [N, ld] (experiments/models_gnn.py:124-149)."""
    L = lib()
    n, W, e = h.shape[0], layer.hidden_features, gs.n_edges
    wp, wq, b1, w2, b2, w3, b3, w4, b4 = layer.wide_weights()
    dev = h.device
    P = torch.empty(n, ld, dtype=torch.float32, device=dev)
    Q = torch.empty(n, ld, dtype=torch.float32, device=dev)
    _wide_linear(feat_cat, k_feat, wp, b1, W, 0, P, ws)
    _wide_linear(feat_cat, k_feat, wq, None, W, 0, Q, ws)
    a1 = torch.empty(max(e, 1), ld, dtype=torch.float32, device=dev)
    check(L.msmp_wide_gather_swish_f32(ptr(P), ptr(Q), ptr(gs.tgt), ptr(gs.col), e, W, ld, ptr(a1), current_stream()), 'msmp_wide_gather_swish_f32')
    msg = torch.empty(max(e, 1), ld, dtype=torch.float32, device=dev)
    if e:
        _wide_linear(a1[:e], W, w2, b2, W, 1, msg[:e], ws)
    agg = torch.empty(n, ld, dtype=torch.float32, device=dev)
    check(L.msmp_wide_scatter_mean_f32(ptr(msg), ptr(gs.rowptr), n, W, ld, ptr(agg), current_stream()), 'msmp_wide_scatter_mean_f32')
    upd_in = torch.cat((h[:, :W], agg[:, :W], variables), 1)
    pad = (-upd_in.shape[1]) % 4
    if pad:
        upd_in = torch.nn.functional.pad(upd_in, (0, pad))
    upd_in = upd_in.contiguous()
    z = torch.empty(n, ld, dtype=torch.float32, device=dev)
    _wide_linear(upd_in, 2 * W + variables.shape[1], w3, b3, W, 1, z, ws)
    y = torch.empty(n, ld, dtype=torch.float32, device=dev)
    _wide_linear(z, W, w4, b4, W, 0, y, ws)
    return y


def _mp_layer_wide(h, u, pos_x, variables, gs, main, gate, eps):
    """GNN_LayerLin (or a gated pair of them) at a hidden width other than 128: the HIP path of wide_kernels.hip.  h [N, W]."""
    L = lib()
    W = main.hidden_features
    if main.MODE != _lib.MSMP_LAYER_LIN:
        raise _lib.MsmpError('the width-generic layer path implements GNN_LayerLin (the layer of the GLU classes)')
    ld = 128 * ((W + 127) // 128)
    n = h.shape[0]
    hp = torch.zeros(n, ld, dtype=torch.float32, device=h.device)
    hp[:, :W] = h
    feat_cat = torch.cat((h, u, pos_x.reshape(-1, 1), variables), 1)
    k_feat = feat_cat.shape[1]
    pad = (-feat_cat.shape[1]) % 4
    if pad:
        feat_cat = torch.nn.functional.pad(feat_cat, (0, pad))
    feat_cat = feat_cat.contiguous()
    k_max = max(feat_cat.shape[1], 2 * W + variables.shape[1] + 3)
    ws = _Workspace.get(L.msmp_linear_workspace_bytes(k_max, W), h.device)
    y_main = _wide_head(hp, feat_cat, k_feat, variables, gs, main, ld, ws)
    y_gate = _wide_head(hp, feat_cat, k_feat, variables, gs, gate, ld, ws) if gate is not None else None
    out = torch.empty(n, ld, dtype=torch.float32, device=h.device)
    check(L.msmp_wide_norm_blend_f32(ptr(hp), ptr(y_gate), ptr(y_main), ptr(gs.graph_ptr), gs.n_graphs, W, ld, eps, ptr(out), current_stream()),
          'msmp_wide_norm_blend_f32')
    return out[:, :W].contiguous()


def _mp_layer_wide_autograd(h, u, pos_x, variables, gs, main, gate, eps):
    """The same layer as differentiable PyTorch-ROCm ops (training of the GLU ablation classes): gathers, F.linear, index_add_
    mean, InstanceNorm and the blend, formula by formula as experiments/models_gnn.py:124-149, 1486-1489."""
    import torch.nn.functional as F
    i, j = gs.tgt_long, gs.col_long
    n = h.shape[0]
    deg = (gs.rowptr[1:] - gs.rowptr[:-1]).clamp(min=1).to(h.dtype)[:, None]
    batch = torch.repeat_interleave(torch.arange(gs.n_graphs, device=h.device), (gs.graph_ptr[1:] - gs.graph_ptr[:-1]).long())
    cnt = (gs.graph_ptr[1:] - gs.graph_ptr[:-1]).clamp(min=1).to(h.dtype)[:, None]
    pos = pos_x.reshape(-1, 1)

    def head(layer):
        sw = lambda x: x * torch.sigmoid(x)
        cat = torch.cat((h[i], h[j], u[i] - u[j], pos[i] - pos[j], variables[i]), -1)
        m = sw(layer.message_net_2[0](sw(layer.message_net_1[0](cat))))
        agg = torch.zeros(n, m.shape[1], dtype=m.dtype, device=m.device).index_add_(0, i, m) / deg
        y = layer.update_net_2[0](sw(layer.update_net_1[0](torch.cat((h, agg, variables), -1))))
        mean = torch.zeros(gs.n_graphs, y.shape[1], dtype=y.dtype, device=y.device).index_add_(0, batch, y) / cnt
        yc = y - mean[batch]
        var = torch.zeros_like(mean).index_add_(0, batch, yc * yc) / cnt
        return yc / torch.sqrt(var + eps)[batch]

    out = head(main)
    if gate is None:
        return out
    tau = torch.sigmoid(head(gate))
    return (1.0 - tau) * h + tau * (out * torch.sigmoid(out))


def mp_layer(h, u, pos_x, variables, structure, main, gate=None, eps=1e-5, dense_message=None, feat=None, decode=None):
    """One message-passing layer (or one gated pair) on the device through msmp_mp_layer_f32.
    h [N,128], u [N,Tw], pos_x [N,1] or [N], variables [N,nv]: float32 CUDA tensors.
    dense_message: None -> module default (factorised message_net_1); True -> literal per-edge GEMM.
    Under autograd (training) the forward is the same HIP call and the backward an explicit recompute with library GEMMs
    and the HIP glue / weight-gradient kernels of train_kernels.hip (msmp_pde_amd.autograd)."""
    gs = structure
    if gs is None or h.device.type != 'cuda':
        raise _lib.MsmpError('mp_layer needs CUDA tensors and a GraphStructure (HIP path only, no CPU fallback)')
    need_grad = torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in main._params8()))
    if main.wide:
        if need_grad:
            return _mp_layer_wide_autograd(h, u.to(h.dtype), pos_x, variables.to(h.dtype), gs, main, gate, eps)
        return _mp_layer_wide(_f32c(h), _f32c(u), _f32c(pos_x).reshape(-1), _f32c(variables), gs, main, gate, eps)
    hd, u, pos_x, variables = _f32c(h), _f32c(u), _f32c(pos_x).reshape(-1), _f32c(variables)
    n = hd.shape[0]
    assert n == gs.n_nodes and hd.shape[1] == HIDDEN and u.shape[1] == main.time_window
    assert variables.shape[1] == main.n_variables and pos_x.numel() == n
    if not need_grad:
        return _mp_layer_hip(hd, u, pos_x, variables, gs, main, gate, eps, dense_message, feat, decode)
    assert decode is None, 'the fused decoder epilogue is an inference path'

    from .autograd import MPLayerFunction
    params = list(main._params8()) + (list(gate._params8()) if gate is not None else [])
    hin = h if (h.dtype == torch.float32 and h.is_contiguous()) else h.to(torch.float32).contiguous()
    return MPLayerFunction.apply(hin, u, pos_x, variables, gs, main, gate, eps, _mp_layer_hip, *params)
