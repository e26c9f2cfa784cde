"""Drop-in solver classes: same constructor, `forward(data)` signature, `repr() == 'GNN'` and
state_dict key names as the reference, with the message-passing stack on the HIP kernels.

Reference classes:
  MP_PDE_Solver               experiments/models_gnn.py:151-281
  MP_PDE_SolverGated          experiments/models_gnn.py:1067-1218
  MP_PDE_SolverLEMLinGated    experiments/models_gnn.py:1220-1377      ("MSMP-PDE", train.py:52-56)
  MP_PDE_Solver2D             experiments/models_gnn2D.py:17-141
  MP_PDE_Solver2DGated        experiments/models_gnn2D.py:143-288
  MP_PDE_Solver2DLEMLinGated  experiments/models_gnn2D.py:290-458      ("MSMP-PDE2D", train.py:116-120)
  MP_PDE_SolverLEMLin / MP_PDE_Solver2DLEMLin   models_gnn.py:619-756 / models_gnn2D.py:920-1057   ("LEM" / "LEM2D" ablations)
Inference (no autograd) runs on HIP kernels end to end: the encoder (msmp_lem_encoder_nodes_f32 with lemoutput_mlp fused, or
msmp_mlp2_swish_f32 for embedding_mlp), the L x [message -> mean -> update -> InstanceNorm (-> gate blend)] loop
(msmp_mp_layer_f32), `double_mlp` of the 2-D classes (msmp_linear_swish_f32) and the decoder CNN with the Euler update
(msmp_decoder_f32 / msmp_decoder2d_f32).  Under autograd the layers and the LEM encoder use their
HIP forward / backward pairs (autograd.py, lem.py) and the small encoder / decoder modules PyTorch-ROCm ops.  `pde.L`,
`pde.tmax`, `pde.dt` are read at call time (they are mutated after construction, experiments/train.py:355-358).  Compute
dtype is float32; the result is returned in the dtype of `data.x`.
"""
import functools

import torch
from torch import nn

from . import _lib
from ._lib import lib, check, ptr, current_stream, PARAM_EPOCH
from .graph import structure_of
from .layers import GNN_Layer, GNN_LayerLin, Swish, mp_layer, node_features
from .lem import LEM, LEMS
from .reductions import bias_add

_DECODER = {20: (15, 4, 10), 25: (16, 3, 14), 50: (12, 2, 10)}   # models_gnn.py:210-224; models_gnn2D.py:79-88


class _Linear(nn.Linear):
    """nn.Linear (same parameters / state_dict keys) whose bias gradient is a deterministic column sum on the library's own kernel
    (reductions.bias_add): the encoder's and decoder's Linear layers then put no library reduction into a captured training step."""

    def forward(self, x):
        if self.bias is None or not (x.is_cuda and x.dtype == torch.float32 and torch.is_grad_enabled() and self.bias.requires_grad):
            return super().forward(x)
        y = torch.nn.functional.linear(x, self.weight)
        return bias_add(y.reshape(-1, y.shape[-1]), self.bias).view(y.shape)


def _lin(i, o):
    return _Linear(i, o, dtype=torch.float32)


_TOEPLITZ = {}


def _conv1d_as_matmul(x, weight, bias, stride):
    """Conv1d(x [N, Cin, Lin]; weight [Cout, Cin, k], stride) as ONE dense GEMM [N, Cin Lin] x T [Cin Lin, Cout Lout], T the
    convolution's Toeplitz matrix, itself the weight times a fixed 0/1 shift tensor (differentiable both ways as small GEMMs).
    The decoder's convolutions have 1-2 input and 8 output channels: as unfold + einsum they are [N L, 16] x [16, 8] products for
    which the library picked a 32x16-tile kernel (3.7 ms of a 48-ms training iteration at batch 512, plus the unfold backward);
    as Toeplitz products they are [N,128] x [128,304] and [N,304] x [304,25]: well-shaped GEMMs forward and backward."""
    cout, cin, k = weight.shape
    n, _, lin = x.shape
    lout = (lin - k) // stride + 1
    key = (k, lin, stride, x.device, weight.dtype)
    if key not in _TOEPLITZ:
        # E[tap, i, l] = 1 where input position i feeds output position l through that tap (i = stride l + tap): T = w . E is a
        # [Cout Cin, k] x [k, Lin Lout] product forward and backward (a gather's backward would be a sorted index_put, a dense
        # selection matrix over all of T's entries a 20-MB GEMV per call)
        i = torch.arange(lin, device=x.device)[None, :, None]
        l = torch.arange(lout, device=x.device)[None, None, :]
        tap = torch.arange(k, device=x.device)[:, None, None]
        _TOEPLITZ[key] = (i == stride * l + tap).to(weight.dtype).reshape(k, lin * lout)
    t = (weight.reshape(cout * cin, k) @ _TOEPLITZ[key]).view(cout, cin, lin, lout).permute(1, 2, 0, 3).reshape(cin * lin, cout * lout)
    out = x.reshape(n, cin * lin) @ t
    return bias_add(out, bias, lout).view(n, cout, lout)


def _decoder_autograd(x, conv1, conv2):
    """output_mlp (Conv1d -> Swish -> Conv1d) for the AUTOGRAD path, written as two Toeplitz GEMMs so that the backward is plain
    GEMM / elementwise work: MIOpen's implicit-GEMM backward-data kernel for this shape faulted on MI355X
    (memory access fault inside igemm_bwd_gtcx35_nhwc_fp32, seen with rocgdb).  x [N, Cin, 128] -> [N, Cout, tw]."""
    mid = _conv1d_as_matmul(x, conv1.weight, conv1.bias, conv1.stride[0])
    mid = mid * torch.sigmoid(mid)
    return _conv1d_as_matmul(mid, conv2.weight, conv2.bias, 1)


class _Conv1dNoMIOpen(nn.Conv1d):
    """nn.Conv1d (same parameters and state_dict keys: `weight`, `bias`) whose forward is a Toeplitz GEMM.  Under autograd the
    library path of these decoder shapes ([N, C, 128], kernel 16 / stride 3 and [N, 8, 38], kernel 14) reaches MIOpen's
    implicit-GEMM backward-data kernel, which faulted on MI355X (igemm_bwd_gtcx35_nhwc_fp32, round 1); `model.output_mlp(x)`
    called the reference's way therefore never dispatches a MIOpen convolution, forward or backward."""

    def forward(self, x):
        return _conv1d_as_matmul(x, self.weight, self.bias, self.stride[0])


class _OutEdgeMean(torch.autograd.Function):
    """Mean of a per-edge tensor [E,128] over each node's OUT-edges (torch_scatter `scatter(..., edge_index[0], reduce='mean')`,
    models_gnn2D.py:607-608) in a fixed summation order: edges regrouped by source once per graph structure, then
    msmp_scatter_mean_f32.  Nodes without out-edges get 0."""

    @staticmethod
    def forward(ctx, msg, gs):
        perm, rowptr = gs.by_source()
        m = msg.detach().to(torch.float32)[perm].contiguous()
        out = torch.empty(gs.n_nodes, m.shape[1], dtype=torch.float32, device=m.device)
        check(lib().msmp_scatter_mean_f32(ptr(m), ptr(rowptr), gs.n_nodes, ptr(out), current_stream()), 'msmp_scatter_mean_f32')
        ctx.gs = gs
        return out

    @staticmethod
    def backward(ctx, g):
        perm, rowptr = ctx.gs.by_source()
        deg = (rowptr[1:] - rowptr[:-1]).clamp(min=1).to(g.dtype)
        return (g / deg[:, None])[ctx.gs.col_long], None


class LSTM(nn.Module):
    """experiments/models_gnn.py:758-767: torch.nn.LSTM(ninp, nhid), returns output[-1].  Parameter names as the reference's
    (`rnn.weight_ih_l0` ...)."""

    def __init__(self, ninp, nhid):
        super().__init__()
        self.ninp, self.nhid = ninp, nhid
        self.rnn = nn.LSTM(ninp, nhid)

    def forward(self, inputs):
        output, _ = self.rnn(inputs.contiguous())
        return output[-1]


class _SolverBase(nn.Module):
    TWO_D = False
    GATED = False
    LEM_ENCODER = False
    ALWAYS_SAVE = False     # MP_PDE_SolverLEMLinGatedSave: LEMS regardless of the constructor argument
    LSTM_ENCODER = False    # the LSTM ablations: torch.nn.LSTM (MIOpen) in place of the LEM, everything after it unchanged
    G2 = False
    RETURN_DIFF = False     # MSSMP_PDE_Solver_sub: forward returns the decoder output, not the Euler update
    LAYER = GNN_Layer

    def __init__(self, pde, time_window=25, hidden_features=128, hidden_layer=6, eq_variables={}, save_state=None):
        super().__init__()
        allowed = (25, 50) if self.TWO_D else (20, 25, 50)
        assert time_window in allowed
        self.save_state = save_state if self.LEM_ENCODER else None        # models_gnn2D.py:325, 360-363: LEM or the stateful LEMS
        self.pde = pde
        self.out_features = time_window
        self.hidden_features = hidden_features
        self.hidden_layer = hidden_layer
        self.time_window = time_window
        self.eq_variables = eq_variables
        comps = 2 if self.TWO_D else 1
        nv = len(eq_variables) + 1
        mk = lambda: self.LAYER(in_features=hidden_features, hidden_features=hidden_features,
                                out_features=hidden_features, time_window=comps * time_window, n_variables=nv)
        self.gnn_layers = nn.ModuleList(mk() for _ in range(hidden_layer))
        if self.GATED or self.G2:
            self.gnn_layers_gate = nn.ModuleList(mk() for _ in range(hidden_layer))
            self.swish = Swish()
        if self.LSTM_ENCODER:
            self.embedding_lstm = LSTM(2 + len(eq_variables) + comps, hidden_features)
            self.lstmoutput_mlp = nn.Sequential(_lin(hidden_features, hidden_features), Swish(),
                                                _lin(hidden_features, hidden_features), Swish())
        elif self.LEM_ENCODER:
            self.embedding_lem = (LEMS if (save_state or self.ALWAYS_SAVE) else LEM)(2 + len(eq_variables) + comps, hidden_features)
            self.lemoutput_mlp = nn.Sequential(_lin(hidden_features, hidden_features), Swish(),
                                               _lin(hidden_features, hidden_features), Swish())
        else:
            self.embedding_mlp = nn.Sequential(_lin(comps * time_window + 2 + len(eq_variables), hidden_features), Swish(),
                                               _lin(hidden_features, hidden_features), Swish())
        k1, s1, k2 = _DECODER[time_window]
        if self.TWO_D:
            self.double_mlp = nn.Sequential(_lin(hidden_features, 2 * hidden_features), Swish(),
                                            nn.Unflatten(1, (2, hidden_features)))
        self.output_mlp = nn.Sequential(_Conv1dNoMIOpen(comps, 8, k1, stride=s1, dtype=torch.float32), Swish(),
                                        _Conv1dNoMIOpen(8, comps, k2, stride=1, dtype=torch.float32))

    def __repr__(self):
        return 'GNN'     # every helper of the reference dispatches on this (train_helper.py:99,110,124,...)

    # -- feature preparation -----------------------------------------------------------------
    def _variable_columns(self, data):
        """(tensor [N,1], divisor) of every equation-variable column after pos_t, in the reference's order."""
        ev = self.eq_variables
        cols = []
        if self.TWO_D:      # models_gnn2D.py:112-116 -- NB 'b' divides data.a (reference behaviour, kept)
            if 'a' in ev:
                cols.append((data.a, ev['a']))
            if 'b' in ev:
                cols.append((data.a, ev['b']))
        else:               # models_gnn.py:250-266
            for k in ('alpha', 'beta', 'gamma'):
                if k in ev:
                    cols.append((getattr(data, k), ev[k]))
            for k in ('bc_left', 'bc_right'):
                if k in ev:
                    cols.append((getattr(data, k), 1))
            for k in ('c', 'D', 'r'):
                if k in ev:
                    cols.append((getattr(data, k), ev[k]))
        return cols

    def _prepare(self, data, want_feat):
        """u, pos_x [N,1], pos_t [N,1], variables [N,nv] (float32) and the packed feature rows of the tile kernel from one HIP
        launch (msmp_prepare_nodes); same values as the tensor expressions of the reference's forward (:1325-1352)."""
        import ctypes
        x, pos = data.x, data.pos
        cols = self._variable_columns(data)
        ok = lambda t: t.is_cuda and t.dtype in (torch.float32, torch.float64)
        if not (ok(x) and ok(pos) and all(ok(c) for c, _ in cols)):
            return None
        # the kernel indexes pos as [N, 2] = (t, x), every column as N contiguous values and x as [N, comps * time_window]:
        # anything else takes the tensor expressions of the reference (which raise or broadcast as they always did)
        comps = 2 if self.TWO_D else 1
        if not (x.dim() == 2 and x.shape[1] == comps * self.time_window and pos.dim() == 2 and pos.shape == (x.shape[0], 2)
                and all(c.numel() == x.shape[0] for c, _ in cols)):
            return None
        x, pos = x.contiguous(), pos.contiguous()
        cs = [c.reshape(-1).contiguous() for c, _ in cols]
        n, tw_feat, nc = x.shape[0], x.shape[1], len(cs)
        dev = x.device
        L = lib()
        u = torch.empty(n, tw_feat, dtype=torch.float32, device=dev)
        pos_x = torch.empty(n, 1, dtype=torch.float32, device=dev)
        pos_t = torch.empty(n, 1, dtype=torch.float32, device=dev)
        variables = torch.empty(n, 1 + nc, dtype=torch.float32, device=dev)
        feat = torch.empty(n, L.msmp_node_feature_stride(tw_feat, 1 + nc), dtype=torch.float32, device=dev) if want_feat else None
        check(L.msmp_prepare_nodes(ptr(x), int(x.dtype == torch.float64), ptr(pos), int(pos.dtype == torch.float64), n, tw_feat,
                                   float(self.pde.L), float(self.pde.tmax), nc, (ctypes.c_void_p * max(nc, 1))(*[ptr(c) for c in cs]),
                                   (ctypes.c_int * max(nc, 1))(*[int(c.dtype == torch.float64) for c in cs]),
                                   (ctypes.c_double * max(nc, 1))(*[float(d) for _, d in cols]), ptr(u), ptr(pos_x), ptr(pos_t),
                                   ptr(variables), ptr(feat), current_stream()), 'msmp_prepare_nodes')
        return u, pos_x, pos_t, variables, feat

    def _variables(self, data, pos_t):
        ev = self.eq_variables
        cols = [pos_t]
        if self.TWO_D:      # models_gnn2D.py:112-116 -- NB 'b' divides data.a (reference behaviour, kept)
            if 'a' in ev:
                cols.append(data.a / ev['a'])
            if 'b' in ev:
                cols.append(data.a / ev['b'])
        else:               # models_gnn.py:250-266
            for k in ('alpha', 'beta', 'gamma'):
                if k in ev:
                    cols.append(getattr(data, k) / ev[k])
            for k in ('bc_left', 'bc_right'):
                if k in ev:
                    cols.append(getattr(data, k))
            for k in ('c', 'D', 'r'):
                if k in ev:
                    cols.append(getattr(data, k) / ev[k])
        return torch.cat([c.to(pos_t.dtype) for c in cols], -1)

    def _encode(self, u, pos_x, pos_t, variables, dt):
        if self.LSTM_ENCODER:           # models_gnn.py:880-887 / 1042-1052; models_gnn2D.py:752-758 / 894-900
            return self.lstmoutput_mlp(self.embedding_lstm(self._step_inputs(u, pos_x, pos_t, variables, dt).permute(1, 0, 2)))
        if not self.LEM_ENCODER:        # models_gnn.py:269-270
            if torch.is_grad_enabled() and any(p.requires_grad for p in self.embedding_mlp.parameters()):
                return self.embedding_mlp(torch.cat((u, pos_x, variables), -1))      # differentiable PyTorch path
            return self._embed_hip(u, pos_x, variables)
        if isinstance(self.embedding_lem, LEMS):       # stateful encoder: always the state-taking recurrence kernel
            return self.lemoutput_mlp(self.embedding_lem.forward_nodes(self._step_inputs(u, pos_x, pos_t, variables, dt)))
        if self.hidden_features != 128:     # the GLU classes: LEM cell and lemoutput_mlp as PyTorch-ROCm ops (any width; lem.LEM.forward_nodes)
            return self.lemoutput_mlp(self.embedding_lem.forward_nodes(self._step_inputs(u, pos_x, pos_t, variables, dt)))
        grad = torch.is_grad_enabled() and any(p.requires_grad for p in self.embedding_lem.parameters())
        if not grad:                    # step inputs assembled inside the kernel (no [N, T, ninp] tensor)
            h = self.embedding_lem.encode_nodes(u, pos_x, pos_t, variables, dt, self.TWO_D, self.lemoutput_mlp)
            if h is not None:
                return h
        lem_in = self._step_inputs(u, pos_x, pos_t, variables, dt)
        if grad:
            h = self.embedding_lem.forward_nodes(lem_in)       # HIP training kernels (recurrence forward + BPTT)
            return self.lemoutput_mlp(h)
        return self.embedding_lem.encode(lem_in, self.lemoutput_mlp)       # fused HIP kernel (recurrence + MLP)

    def _step_inputs(self, u, pos_x, pos_t, variables, dt):
        """The recurrent encoder's per-step inputs, node-major [N, T, ninp]."""
        tw, n = self.time_window, u.shape[0]
        if self.TWO_D:                  # models_gnn2D.py:421-436
            ts = dt.view(1, tw) + pos_t
            x = torch.stack([pos_x.expand(n, tw), u[:, :tw], u[:, tw:], ts], -1)       # [N, tw, 4]
            return torch.cat((x, variables[:, None, 1:].expand(n, tw, variables.shape[1] - 1)), -1)
        t_len = u.shape[1]              # models_gnn.py:1356-1363
        return torch.cat((pos_x[:, None, :].expand(n, t_len, 1), u[:, :, None],
                          variables[:, None, :].expand(n, t_len, variables.shape[1])), -1)

    def _g2_pair(self, h, u, pos_x, variables, gs, i):
        """models_gnn2D.py:606-611 (gradient gating): tau = tanh(mean over the out-edges j -> i of node j of |t_j - t_i|^2) with
        t = Swish(gate layer), then the same blend as the gated classes.  Both layers are the fused HIP layer call
        (GNN_LayerLin incl. its InstanceNorm); the gate statistic and the blend are PyTorch-ROCm ops."""
        t = self.swish(mp_layer(h, u, pos_x, variables, gs, self.gnn_layers_gate[i], None))
        tau = torch.tanh(_OutEdgeMean.apply((t[gs.col_long] - t[gs.tgt_long]) ** 2, gs))
        return (1.0 - tau) * h + tau * self.swish(mp_layer(h, u, pos_x, variables, gs, self.gnn_layers[i], None))

    def _embed_packed(self, dev):
        """Packed embedding_mlp weights (msmp_pack_mlp2_f32), cached per parameter version; built on the CURRENT stream."""
        lin1, lin2 = self.embedding_mlp[0], self.embedding_mlp[2]
        ps = (lin1.weight, lin1.bias, lin2.weight, lin2.bias)
        key = (PARAM_EPOCH[0],) + tuple((p.data_ptr(), p._version, str(p.device)) for p in ps)
        if getattr(self, '_embed_key', None) != key:
            L = lib()
            f = [p.detach().to(torch.float32).contiguous() for p in ps]
            blob = torch.empty(L.msmp_packed_mlp2_floats(lin1.in_features), dtype=torch.float32, device=dev)
            check(L.msmp_pack_mlp2_f32(*[ptr(t) for t in f], lin1.in_features, ptr(blob), current_stream()), 'msmp_pack_mlp2_f32')
            self._embed_blob, self._embed_key = blob, key
        return self._embed_blob

    def _embed_hip(self, u, pos_x, variables):
        """embedding_mlp as one HIP kernel (msmp_mlp2_swish_f32); the packed weights are cached per parameter version."""
        L = lib()
        k_in = self.embedding_mlp[0].in_features
        blob = self._embed_packed(u.device)
        stride = L.msmp_mlp2_input_stride(k_in)
        pad = u.new_zeros(u.shape[0], stride - k_in)
        x = torch.cat((u, pos_x, variables, pad), -1).contiguous()
        assert x.shape[1] == stride
        out = torch.empty(u.shape[0], self.hidden_features, dtype=torch.float32, device=u.device)
        check(L.msmp_mlp2_swish_f32(ptr(x), u.shape[0], k_in, ptr(blob), ptr(out), current_stream()), 'msmp_mlp2_swish_f32')
        return out

    def _dt(self, dev):
        """cumsum(dt) of the Euler update (models_gnn.py:275): constant until pde.dt changes; built on the CURRENT stream."""
        dkey = (self.time_window, float(self.pde.dt), str(dev))
        if getattr(self, '_dt_key', None) != dkey:
            with torch.inference_mode(False):      # a plain tensor: an inference-mode tensor could not enter the autograd decoder later
                self._dt_cum, self._dt_key = torch.cumsum(torch.ones(self.time_window, dtype=torch.float32, device=dev) * self.pde.dt, 0), dkey
        return self._dt_cum

    def warm_caches(self, dev=None):
        """Build every lazily packed operand of an inference forward (layer blobs, encoder blobs, cumsum(dt)) on the CURRENT
        stream.  Callers that evaluate one model on SEVERAL streams (sub_batches > 1, bench.SplitWorkload) call this first and
        make the side streams wait for the current one: a cache is created by whichever stream gets there first, and the others
        would read a blob whose pack kernels they never waited for."""
        dev = dev or next(self.parameters()).device
        for layers in (self.gnn_layers, getattr(self, 'gnn_layers_gate', ())):
            for layer in layers:
                layer.wide_weights() if layer.wide else layer.packed()
        if self.LEM_ENCODER and not self.LSTM_ENCODER and self.hidden_features == 128 and not isinstance(self.embedding_lem, LEMS):
            self.embedding_lem._pack(self.lemoutput_mlp)
        elif not self.LEM_ENCODER and not self.LSTM_ENCODER:
            self._embed_packed(dev)
        self._dt(dev)

    # -- forward -------------------------------------------------------------------------------
    INPUT_RANGE = 255.0

    def input_range(self, data):
        """Largest |value| among the node features the kernels see (u, pos_x, the variables columns) -- one device reduction and a
        host sync, so NOT part of forward(): call it once per dataset.  The default (fp16-split) matrix path carries node rows
        scaled by 2^8 and saturates them at +-65504 (tile_kernels.hip), i.e. it needs |feature| <= 255 after the reference's own
        normalisation (pos / L, variables / eq_variables): PDE data of order one.  Data outside that range must run on the exact-fp32
        MFMA kernels: `msmp_pde_amd.lib().msmp_tune(b'split', 0)` (no range limit, ~2x slower)."""
        prep = self._prepare(data, False)
        if prep is None:
            raise RuntimeError('input_range: device tensors of dtype float32 / float64 expected')
        u, pos_x, _, variables, _ = prep
        return float(max(u.abs().max(), pos_x.abs().max(), variables.abs().max()))

    def validate_inputs(self, data):
        """Raise if `data` is outside the range the default matrix path represents (see input_range)."""
        r = self.input_range(data)
        if not r <= self.INPUT_RANGE:
            raise ValueError(f'node features reach |x| = {r:.4g} > {self.INPUT_RANGE:g}: the fp16-split matrix path would saturate them; '
                             "rescale the data or select the exact-fp32 kernels with lib().msmp_tune(b'split', 0)")
        return r

    # >1: under no_grad, forward() evaluates the batch as that many sub-batches of WHOLE graphs, each on a stream of its own
    # (graphs are independent: same values; bit-identical when a tile of the message kernel divides a graph).  The ~20 dependent
    # kernels of a rollout step then form several chains whose kernels overlap: 1.06 -> 0.91 ms per step at 256 graphs,
    # 6.56 -> 6.43 ms at 2048 (scripts/sub_batches.py).  Set on an instance (`model.sub_batches = 2`) or per class.
    sub_batches = 1

    # What forward() does about the range of the default (fp16-split) matrix path, which carries node rows scaled by 2^8 in fp16
    # (|x| <= 255) and activations scaled by 2^6 (|x| <= 1023) -- the reference has no such limit (models_gnn.py:1315-1377):
    #   'auto' (default)  the FIRST forward of a model (and the first after load_state_dict) is followed by one stream
    #                     synchronisation and a read of the sticky status word (msmp_last_status): if a kernel left the range,
    #                     that forward is evaluated AGAIN on the exact-fp32 kernels (lib().msmp_tune(b'split', 0) for this model
    #                     only) and the model stays on them, with one MsmpRangeWarning.  Later forwards read the word at entry
    #                     without a synchronisation (flags raised by work that has completed since): same switch, from that call
    #                     on -- the call whose kernels raised the flag has returned by then, and the warning says so.
    #   'sync'            every forward is checked like the first one: no call ever returns an out-of-range result (one stream
    #                     synchronisation per forward: ~1.5 % of a 2048-graph rollout step, more on small batches).
    #   'warn'            round 3's behaviour: warn at the next forward, change nothing.
    # Under autograd and during a hipGraph capture the synchronous check is skipped (entry check only).
    range_policy = 'auto'
    _range_exact = False          # this model runs on the exact-fp32 kernels
    _range_probed = False         # the synchronous first-call check has been done

    def _range_switch(self, flags, when):
        self._range_exact = True
        _lib._range_report(flags, f'{type(self).__name__}: {when}; this model now runs on the exact-fp32 MFMA kernels (no range limit, about 2x slower). '
                                  'Rescale the data to keep the fast path; model._range_exact = False switches back.')
        _lib.last_status(reset=True)

    def _dispatch(self, data):
        if self.sub_batches > 1 and not torch.is_grad_enabled() and data.x.is_cuda and not torch.cuda.is_current_stream_capturing() \
                and not isinstance(getattr(self, 'embedding_lem', None), LEMS):
            return self._forward_sub_batches(data, int(self.sub_batches))
        return self._forward(data)

    def forward(self, data):
        policy = self.range_policy
        if policy == 'warn' or not data.x.is_cuda or not lib().msmp_tune_query(b'split'):
            _lib.status_check()
            return self._dispatch(data)
        if not self._range_exact:
            flags = _lib.last_status()          # one host read, no device synchronisation
            if flags:
                self._range_switch(flags, 'raised by kernel work that completed before this call (an EARLIER result is saturated or not finite)')
        if self._range_exact:
            with _lib.exact_fp32():
                return self._dispatch(data)
        probe = (policy == 'sync' or not self._range_probed) and not torch.is_grad_enabled() and not torch.cuda.is_current_stream_capturing()
        stateful = isinstance(getattr(self, 'embedding_lem', None), LEMS)
        states = self.embedding_lem.states if (probe and stateful) else None      # a second evaluation must start from the same carried states
        out = self._dispatch(data)
        if probe:
            self._range_probed = True
            torch.cuda.current_stream().synchronize()
            flags = _lib.last_status()
            if flags:
                self._range_switch(flags, 'raised by this forward, which has been evaluated again')
                if stateful:
                    self.embedding_lem.states = states
                with _lib.exact_fp32():
                    out = self._dispatch(data)
        return out

    def load_state_dict(self, *args, **kwargs):
        self._range_probed = False          # new weights: check the first forward again
        return super().load_state_dict(*args, **kwargs)

    def _sub_batch_plan(self, data, parts):
        """Per-structure split of `data` into `parts` contiguous blocks of graphs (dist.shard_graph: node rows are views, edges
        filtered and rebased), cached on the GraphStructure that travels with the graph through the rollout, with a stream each."""
        from .dist import shard_graph
        gs = structure_of(data)
        plans = gs.__dict__.setdefault('_sub_batch_plans', {})
        plan = plans.get(parts)
        if plan is None:
            n_parts = max(1, min(parts, gs.n_graphs))
            subs = []
            for r in range(n_parts):
                sub = shard_graph(data, r, n_parts)
                subs.append(sub)
            # node offsets of the blocks
            offs, o = [], 0
            for sub in subs:
                offs.append((o, o + sub.x.shape[0]))
                o += sub.x.shape[0]
            assert o == data.x.shape[0]
            for sub in subs:
                structure_of(sub).tiles()                   # CSR + tiles built once, on the calling stream
            plan = plans[parts] = {'subs': subs, 'offs': offs, 'streams': [torch.cuda.Stream(device=data.x.device) for _ in subs]}
        return plan

    def _forward_sub_batches(self, data, parts):
        plan = self._sub_batch_plan(data, parts)
        share = lib().msmp_tune_query(b'lem_share')
        lib().msmp_tune(b'lem_share', max(1, len(plan['subs'])))        # the sub-batches' LEM launches share the CUs (restored below)
        n = data.x.shape[0]
        cur = torch.cuda.current_stream()
        self.warm_caches(data.x.device)      # shared blobs are packed HERE, on the caller's stream; every side stream waits for it below
        out = None
        per_node = [k for k, v in data.__dict__.items() if torch.is_tensor(v) and not k.startswith('_') and k not in ('edge_index', 'batch')
                    and v.dim() >= 1 and v.shape[0] == n]
        outs = []
        try:
            for sub, (n0, n1), st in zip(plan['subs'], plan['offs'], plan['streams']):
                for k in per_node:                              # this step's node rows: views of the caller's tensors
                    setattr(sub, k, getattr(data, k)[n0:n1])
                st.wait_stream(cur)
                with torch.cuda.stream(st):
                    outs.append(self._forward(sub))
        finally:
            lib().msmp_tune(b'lem_share', share)
        for o, st in zip(outs, plan['streams']):
            cur.wait_stream(st)
            o.record_stream(cur)                                # allocated on the side stream, consumed on the caller's
        out = torch.cat(outs, 0)
        return out

    def _forward(self, data):
        u_in = data.x
        pos = data.pos
        gs = structure_of(data)
        tw = self.time_window
        # the [u | pos | vars] columns of message_net_1's input do not change over the layers: packed once for the tile kernel
        want_feat = not torch.is_grad_enabled() and self.hidden_features == 128 and gs.tiles() is not None
        prep = self._prepare(data, want_feat) if u_in.is_cuda else None
        if prep is not None:            # one HIP launch
            u, pos_x, pos_t, variables, feat = prep
        else:                           # odd dtypes: the tensor expressions of the reference
            pos_x = (pos[:, 1][:, None] / self.pde.L)
            pos_t = (pos[:, 0][:, None] / self.pde.tmax)
            variables = self._variables(data, pos_t).float()
            pos_x, pos_t = pos_x.float(), pos_t.float()
            u = u_in.float().contiguous()
            feat = node_features(u, pos_x.reshape(-1).contiguous(), variables.contiguous()) if want_feat else None
        dt = self._dt(u.device)

        h = self._encode(u, pos_x, pos_t, variables, dt)
        # msmp_tune("dec_fuse", 1): the decoder as the epilogue of the LAST layer's node tail (1-D classes, time_window
        # 25, inference): bit-identical, measured in round 4 -- off by default
        fuse = (not self.TWO_D and not self.G2 and tw == 25 and self.hidden_features == 128 and not torch.is_grad_enabled()
                and u.is_cuda and bool(lib().msmp_tune_query(b'dec_fuse')))
        for i in range(self.hidden_layer):
            if self.G2:
                h = self._g2_pair(h, u, pos_x, variables, gs, i)
                continue
            gate = self.gnn_layers_gate[i] if self.GATED else None
            if fuse and i == self.hidden_layer - 1:
                both = mp_layer(h, u, pos_x, variables, gs, self.gnn_layers[i], gate, feat=feat,
                                decode=(self.output_mlp[0], self.output_mlp[2], None if self.RETURN_DIFF else u, float(self.pde.dt), tw))
                if both is not None:
                    return both[1].to(u_in.dtype)
            h = mp_layer(h, u, pos_x, variables, gs, self.gnn_layers[i], gate, feat=feat)

        return self._decode(h, u, u_in, dt, tw)

    def _decode(self, h, u, u_in, dt, tw):
        grad_path = torch.is_grad_enabled() and (h.requires_grad or any(p.requires_grad for p in self.output_mlp.parameters()))
        if self.TWO_D and grad_path:    # models_gnn2D.py:125-141
            diff = _decoder_autograd(self.double_mlp(h), self.output_mlp[0], self.output_mlp[2])
            out = (u.view(-1, 2, tw) + dt.view(1, 1, tw) * diff).flatten(1, 2)
        elif self.TWO_D:                # fused: conv(2->8) -> Swish -> conv(8->2) -> u + cumsum(dt) * diff
            hd = self._double_mlp_hip(h)
            out = torch.empty_like(u)
            c1, c2 = self.output_mlp[0], self.output_mlp[2]
            w = [p.detach().to(torch.float32).contiguous() for p in (c1.weight, c1.bias, c2.weight, c2.bias)]   # kept alive
            check(lib().msmp_decoder2d_f32(ptr(hd), ptr(u), u.shape[0], tw, ptr(w[0]), ptr(w[1]), ptr(w[2]), ptr(w[3]),
                                           float(self.pde.dt), ptr(out), current_stream()), 'msmp_decoder2d_f32')
        elif grad_path:
            diff = _decoder_autograd(h[:, None], self.output_mlp[0], self.output_mlp[2]).squeeze(1)   # differentiable decoder
            out = diff if self.RETURN_DIFF else u[:, -1:] + dt.view(1, tw) * diff
        else:                           # models_gnn.py:275-279, fused: conv -> Swish -> conv -> u + cumsum(dt) * diff
            out = torch.empty_like(u)
            c1, c2 = self.output_mlp[0], self.output_mlp[2]
            w = [p.detach().to(torch.float32).contiguous() for p in (c1.weight, c1.bias, c2.weight, c2.bias)]   # kept alive
            h = h.contiguous()
            check(lib().msmp_decoder_f32(ptr(h), None if self.RETURN_DIFF else ptr(u), u.shape[0], tw, ptr(w[0]), ptr(w[1]), ptr(w[2]),
                                         ptr(w[3]), float(self.pde.dt), ptr(out), current_stream()), 'msmp_decoder_f32')
        return out.to(u_in.dtype)


    def _double_mlp_hip(self, h):
        """double_mlp (models_gnn2D.py:66-70: Linear(128, 256) + Swish + Unflatten) as one HIP row GEMM with the bias and the Swish in its
        epilogue (msmp_linear_swish_f32); returns [N, 2, 128]."""
        lin = self.double_mlp[0]
        w, b = lin.weight.detach().to(torch.float32).contiguous(), lin.bias.detach().to(torch.float32).contiguous()
        h = h.contiguous()
        n_out, k = w.shape
        need = lib().msmp_linear_swish_workspace_bytes(k, n_out)
        if not need:
            raise RuntimeError(f'double_mlp of {k} -> {n_out} features is outside msmp_linear_swish_f32')
        from .layers import _Workspace
        ws = _Workspace.get(need, h.device)         # per (device, stream); the layers are done with it by now
        out = torch.empty(h.shape[0], n_out, dtype=torch.float32, device=h.device)
        check(lib().msmp_linear_swish_f32(ptr(h), h.shape[0], k, ptr(w), ptr(b), n_out, ptr(out), ptr(ws), ws.numel(), current_stream()),
              'msmp_linear_swish_f32')
        self._dmlp_keep = (w, b)          # alive until the stream has consumed them
        return out.view(h.shape[0], 2, n_out // 2)

    def capture(self, data):
        """hipGraph of `forward` for a fixed graph batch (inference): returns `step(data) -> prediction` that copies the
        per-step inputs (`data.x`, `data.pos`) into static buffers and replays ONE graph launch instead of the ~60 kernel
        launches of the eager forward (the rollout is launch-bound between its kernels: ~0.5 ms of a 7.4 ms step).  The
        structure tensors (edge_index, batch, equation variables) are those of `data` at capture time."""
        return _GraphedForward(self, data)


class _GraphedForward:
    """hipGraph of one forward.  The captured kernels hold RAW pointers, so everything they point at is owned or pinned here:
    * the static input buffers (x, pos) and the output;
    * a PRIVATE layer workspace (layers._Workspace.private): the shared grow-only workspace of the eager path is replaced, and
      its old buffer freed, as soon as any later call needs a larger one;
    * the packed weight blobs the capture read (`layer._packed`, the encoder blobs): they are replaced after an optimizer step,
      load_state_dict or invalidate_packed_weights().  References keep their memory alive, and `__call__` compares what the
      blobs' caches key on (optimizer epoch, parameter storages and versions) with the capture-time values: on a mismatch the
      graph is re-captured, never replayed against stale weights."""

    def __init__(self, model, data):
        import copy
        assert not torch.is_grad_enabled(), 'capture() is for inference: wrap it in torch.no_grad()'
        self.model = model
        self.data = copy.copy(data)                       # shallow: shares edge_index / batch / variables / cached structure
        self.data.x = data.x.clone()
        self.data.pos = data.pos.clone()
        self._capture()

    def _weight_state(self):
        """What the packed-weight caches key on: the global epoch (optimizer steps, invalidate_packed_weights) and every
        parameter's storage and version (load_state_dict, in-place edits)."""
        pde = self.model.pde          # dt, L, tmax are baked into the captured kernel arguments (train.py:355-358 mutates them after construction)
        return (PARAM_EPOCH[0], tuple((p.data_ptr(), p._version) for p in self.model.parameters()),
                (float(pde.dt), float(pde.L), float(pde.tmax)))

    def _weight_blobs(self):
        return [getattr(m, attr) for m in self.model.modules() for attr in ('_packed', '_embed_blob') if getattr(m, attr, None) is not None]

    @torch.no_grad()          # a re-capture from __call__ may come from a training loop: never capture the autograd path
    def _capture(self):
        from .layers import _Workspace
        model = self.model
        self.graph = None
        with _Workspace.private(self.data.x.device) as ws:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                 # warm-up on a side stream: weight packing, workspaces, CSR build
                for _ in range(2):
                    # through _forward, like the capture below: with sub_batches > 1 the eager forward() would fan out over streams
                    # that share this ONE private workspace (sized for a sub-batch) and the capture would then outgrow it
                    model._forward(self.data)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                self.out = model(self.data)
            self.graph = graph
        self._ws = ws.buffers()                           # pinned: the captured launches point into them
        self._blobs = self._weight_blobs()                # pinned
        self._state = self._weight_state()

    def __call__(self, data):
        if self._state != self._weight_state():           # the weights changed since the capture: the eager path would re-pack them
            self._capture()
        self.data.x.copy_(data.x)
        self.data.pos.copy_(data.pos)
        self.graph.replay()
        return self.out.clone()


class MP_PDE_Solver(_SolverBase):
    pass


class MP_PDE_SolverGated(_SolverBase):
    GATED, LAYER = True, GNN_LayerLin


class MP_PDE_SolverLEMLinGated(_SolverBase):
    GATED, LEM_ENCODER, LAYER = True, True, GNN_LayerLin


class MP_PDE_SolverLEMLin(_SolverBase):
    """experiments/models_gnn.py:619-756: LEM encoder in front of the plain GNN_Layer stack (no gate; train.py name 'LEM')."""
    LEM_ENCODER = True


class MP_PDE_Solver2D(_SolverBase):
    TWO_D = True


class MP_PDE_Solver2DGated(_SolverBase):
    TWO_D, GATED, LAYER = True, True, GNN_LayerLin


class MP_PDE_Solver2DLEMLinGated(_SolverBase):
    TWO_D, GATED, LEM_ENCODER, LAYER = True, True, True, GNN_LayerLin


class MP_PDE_Solver2DLEMLin(_SolverBase):
    """experiments/models_gnn2D.py:920-1057 (train.py name 'LEM2D')."""
    TWO_D, LEM_ENCODER = True, True


class MP_PDE_Solver2DLEMLinG2(_SolverBase):
    """experiments/models_gnn2D.py:460-620 (train.py name 'MSG2-PDE2D'): the gated 2-D LEM model with gradient gating."""
    TWO_D, LEM_ENCODER, G2, LAYER = True, True, True, GNN_LayerLin


class _GLUBase(_SolverBase):
    """The 'GLU' classes (train.py names 'MSGMP-PDE' / 'MSGMP-PDE2D'): the gated LEM model at hidden width 164 whose decoder is a
    gated pair of CNNs on the two halves of the hidden state,  out = (1 - scale) u_last + cumsum(dt) scale diff  with
    scale = output_mlp_gate(h[..., :82]), diff = output_mlp_diff(h[..., 82:])  (experiments/models_gnn.py:1379-1523,
    models_gnn2D.py:1198-1366; no sigmoid on `scale`, as in the reference).  Layers: the width-generic HIP path
    (layers._mp_layer_wide); LEM cell, lemoutput_mlp, double_mlp and the two small CNNs: PyTorch-ROCm ops."""
    GATED, LEM_ENCODER, LAYER = True, True, GNN_LayerLin

    def __init__(self, pde, time_window=25, hidden_features=164, hidden_layer=6, eq_variables={}, save_state=None):
        super().__init__(pde, time_window, hidden_features, hidden_layer, eq_variables, save_state)
        comps = 2 if self.TWO_D else 1
        assert time_window == 25, 'the GLU decoder of the reference exists for time_window 25 (Conv1d(.., 8, 6, stride 2) -> Conv1d(8, .., 15) on 82 features)'
        del self.output_mlp
        mk = lambda: nn.Sequential(_Conv1dNoMIOpen(comps, 8, 6, stride=2, dtype=torch.float32), Swish(),
                                   _Conv1dNoMIOpen(8, comps, 15, stride=1, dtype=torch.float32))
        self.output_mlp_gate = mk()
        self.output_mlp_diff = mk()

    def _decode(self, h, u, u_in, dt, tw):
        half = h.shape[1] // 2
        if self.TWO_D:                  # models_gnn2D.py:1349-1366
            hd = self.double_mlp(h)                                             # [N, 2, W]
            half = hd.shape[2] // 2
            diff = self.output_mlp_diff(hd[:, :, half:])
            scale = self.output_mlp_gate(hd[:, :, :half])
            out = ((1.0 - scale) * u.view(-1, 2, tw) + dt.view(1, 1, tw) * scale * diff).flatten(1, 2)
        else:                           # models_gnn.py:1511-1521
            scale = self.output_mlp_gate(h[:, :half][:, None]).squeeze(1)
            diff = self.output_mlp_diff(h[:, half:][:, None]).squeeze(1)
            out = (1.0 - scale) * u[:, -1:] + dt.view(1, tw) * (scale * diff)
        return out.to(u_in.dtype)


class MP_PDE_SolverLEMLinGatedGLU(_GLUBase):
    """experiments/models_gnn.py:1379-1523 (train.py name 'MSGMP-PDE')."""


class MP_PDE_Solver2DLEMLinGatedGLU(_GLUBase):
    """experiments/models_gnn2D.py:1198-1366 (train.py name 'MSGMP-PDE2D')."""
    TWO_D = True


class MP_PDE_SolverLEMLinGatedSave(_SolverBase):
    """experiments/models_gnn.py:1747-1905 (train.py name 'SaveMSMP-PDE'): MSMP-PDE whose LEM keeps its hidden states from one
    call of a rollout to the next (`model.embedding_lem.reset_states()` between sequences)."""
    GATED, LEM_ENCODER, ALWAYS_SAVE, LAYER = True, True, True, GNN_LayerLin


class MP_PDE_SolverLSTMLin(_SolverBase):
    """experiments/models_gnn.py:770-907 (train.py name 'LSTM')."""
    LSTM_ENCODER = True


class MP_PDE_SolverLSTMLinGated(_SolverBase):
    """experiments/models_gnn.py:909-1065 (train.py name 'LSTMGated')."""
    GATED, LSTM_ENCODER, LAYER = True, True, GNN_LayerLin


class MP_PDE_Solver2DLSTMLin(_SolverBase):
    """experiments/models_gnn2D.py:782-918 (train.py name 'LSTM2D')."""
    TWO_D, LSTM_ENCODER = True, True


class MP_PDE_Solver2DLSTMLinGated(_SolverBase):
    """experiments/models_gnn2D.py:622-780 (train.py name 'LSTMGated2D')."""
    TWO_D, GATED, LSTM_ENCODER, LAYER = True, True, True, GNN_LayerLin


class MSSMP_PDE_Solver_sub(_SolverBase):
    """experiments/models_gnn.py:1525-1682: the MSMP-PDE network whose forward returns the decoder output `diff`."""
    GATED, LEM_ENCODER, RETURN_DIFF, LAYER = True, True, True, GNN_LayerLin


class MSSMP_PDE_Solver(nn.Module):
    """experiments/models_gnn.py:1684-1745 (train.py name 'MSSMP-PDE'): two MSMP-PDE networks, `diff` and `scale`;
    out = (1 - scale) * u[:, -1] + cumsum(dt) * (scale * diff)."""

    def __init__(self, pde, time_window=25, hidden_features=128, hidden_layer=6, eq_variables={}):
        super().__init__()
        self.pde, self.time_window, self.eq_variables = pde, time_window, eq_variables
        self.out_features, self.hidden_features, self.hidden_layer = time_window, hidden_features, hidden_layer
        self.diff = MSSMP_PDE_Solver_sub(pde, time_window, hidden_features, hidden_layer, eq_variables)
        self.scale = MSSMP_PDE_Solver_sub(pde, time_window, hidden_features, hidden_layer, eq_variables)

    def __repr__(self):
        return 'GNN'

    def forward(self, data):
        scale = self.scale(data)
        diff = self.diff(data)
        u = data.x
        dt = torch.cumsum(torch.ones(1, self.time_window, dtype=u.dtype, device=u.device) * self.pde.dt, 1)
        return (1.0 - scale) * u[:, -1:] + dt * (scale * diff)


MODEL_NAMES = {   # experiments/train.py:34-183 getModel names -> class
    'MP-PDE': MP_PDE_Solver, 'Gated': MP_PDE_SolverGated, 'MSMP-PDE': MP_PDE_SolverLEMLinGated,
    'MP-PDE2D': MP_PDE_Solver2D, 'Gated2D': MP_PDE_Solver2DGated, 'MSMP-PDE2D': MP_PDE_Solver2DLEMLinGated,
    'LEM': MP_PDE_SolverLEMLin, 'LEM2D': MP_PDE_Solver2DLEMLin, 'MSG2-PDE2D': MP_PDE_Solver2DLEMLinG2,
    'MSSMP-PDE': MSSMP_PDE_Solver, 'SaveMSMP-PDE': MP_PDE_SolverLEMLinGatedSave,
    'LSTM': MP_PDE_SolverLSTMLin, 'LSTMGated': MP_PDE_SolverLSTMLinGated, 'LSTM2D': MP_PDE_Solver2DLSTMLin,
    'LSTMGated2D': MP_PDE_Solver2DLSTMLinGated,
    'MSGMP-PDE': MP_PDE_SolverLEMLinGatedGLU, 'MSGMP-PDE2D': MP_PDE_Solver2DLEMLinGatedGLU,
    'SaveMSMP-PDE2D': functools.partial(MP_PDE_Solver2DLEMLinGated, save_state=True),       # train.py:126-131
}
# Not here: 'GLEMGated2D' (G_PDE_Solver2DLEMLinGated, models_gnn2D.py:1058: layers are torch_geometric's RGATConv, a third-party
# attention layer outside this path's message / update functions) and the grid models BaseCNN / FNO / VNO (no message passing).
