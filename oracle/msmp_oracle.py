"""CPU ORACLE for the MSMP-PDE message-passing rollout step.  TEST INFRASTRUCTURE ONLY.

This file is a plain-numpy (float64 by default) restatement of the reference algorithm for the
hot path of SURVEY.md section 8.  It is the *checker*: only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it.  The product (`msmp-pde_amd/`) never does and
fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * PINNED against the reference's own model / graph code: `tests/golden/gen_golden.py` imports
    /root/reference/experiments/models_gnn.py, models_gnn2D.py and common/utils.py unchanged (on
    pure-torch stand-ins for the third-party packages this image lacks) and stores inputs/outputs in
    tests/golden/*.npz; tests/test_oracle_golden.py checks every function below against them.
  * UNPINNED: the third-party primitives themselves (PyG propagate/mean, PyG InstanceNorm,
    torch_cluster radius_graph/knn_graph) and the LEM recurrence (`lem_cuda`, source absent from the
    reference tree).  The reference holds no fixture for them; they follow the published behaviour
    of those packages.

Every function cites the reference file:line it follows (paths relative to /root/reference).
All arrays are row-major numpy; `edge_index` is int64 [2, E] with row 0 = source j, row 1 = target i.
"""
from types import SimpleNamespace

import numpy as np

H = 128  # hidden_features default, experiments/models_gnn.py:158


# --------------------------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------------------------
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def swish(x):
    """Swish, beta=1: x * sigmoid(x).  experiments/models_gnn.py:12-21."""
    return x * sigmoid(x)


def linear(x, w, b):
    """nn.Linear with weight [out, in]: x @ w.T + b."""
    return x @ w.T + b


def _segment_sum(x, index, n):
    """sum of the rows of x sharing an index -> [n, C] (index_add semantics; rows visited in index order)."""
    out = np.zeros((n, x.shape[1]), dtype=x.dtype)
    if len(index) == 0:
        return out
    order = np.argsort(index, kind='stable')
    idx = index[order]
    starts = np.flatnonzero(np.concatenate(([True], idx[1:] != idx[:-1])))
    out[idx[starts]] = np.add.reduceat(x[order], starts, axis=0)
    return out


def scatter_mean(msg, target, n):
    """PyG aggr='mean' (experiments/models_gnn.py:42,107): sum over edges sharing a target divided
    by max(in-degree, 1); nodes without in-edges give 0.  Row L2 of SURVEY.md section 8a."""
    out = _segment_sum(msg, target, n)
    cnt = np.bincount(target, minlength=n).astype(msg.dtype)
    return out / np.maximum(cnt, 1.0)[:, None]


def instance_norm(x, batch, eps=1e-5):
    """PyG InstanceNorm(affine=False, no running stats), experiments/models_gnn.py:59,66,122,129:
    per graph and channel (x - mean) / sqrt(biased var + eps).  Row L4."""
    b = int(batch.max()) + 1 if batch.size else 0
    cnt = np.maximum(np.bincount(batch, minlength=b).astype(x.dtype), 1.0)[:, None]
    mean = _segment_sum(x, batch, b) / cnt
    xc = x - mean[batch]
    var = _segment_sum(xc * xc, batch, b) / cnt
    return xc / np.sqrt(var + eps)[batch]


def layer_params(sd, prefix):
    """The eight tensors of one GNN_Layer / GNN_LayerLin from a state_dict (reference key names,
    experiments/models_gnn.py:47-58)."""
    g = lambda k: np.asarray(sd[prefix + k])
    return SimpleNamespace(
        w1=g('message_net_1.0.weight'), b1=g('message_net_1.0.bias'),
        w2=g('message_net_2.0.weight'), b2=g('message_net_2.0.bias'),
        w3=g('update_net_1.0.weight'), b3=g('update_net_1.0.bias'),
        w4=g('update_net_2.0.weight'), b4=g('update_net_2.0.bias'))


def edge_messages(p, x, u, pos, variables, edge_index):
    """GNN_Layer.message / GNN_LayerLin.message, experiments/models_gnn.py:69-75 / 132-138.
    i = target = edge_index[1], j = source = edge_index[0].  Row L1."""
    j, i = edge_index[0], edge_index[1]
    cat = np.concatenate((x[i], x[j], u[i] - u[j], pos[i] - pos[j], variables[i]), axis=-1)
    m = swish(linear(cat, p.w1, p.b1))
    return swish(linear(m, p.w2, p.b2))


def node_update(p, x, agg, variables, lin):
    """update(): GNN_Layer (lin=False) experiments/models_gnn.py:77-86: x + Swish(W4 Swish(W3 .)+b4);
    GNN_LayerLin (lin=True) :140-149: W4 Swish(W3 .) + b4, no final activation, no residual.  Row L3."""
    upd = swish(linear(np.concatenate((x, agg, variables), axis=-1), p.w3, p.b3))
    upd = linear(upd, p.w4, p.b4)
    if lin:
        return upd
    return x + swish(upd)


def mp_layer(p, x, u, pos, variables, edge_index, batch, lin, parts=False):
    """GNN_Layer.forward / GNN_LayerLin.forward, experiments/models_gnn.py:61-67 / 124-130:
    propagate (message -> mean over targets -> update) then InstanceNorm over `batch`."""
    msg = edge_messages(p, x, u, pos, variables, edge_index)
    agg = scatter_mean(msg, edge_index[1], x.shape[0])
    pre = node_update(p, x, agg, variables, lin)
    out = instance_norm(pre, batch)
    if parts:
        return SimpleNamespace(msg=msg, agg=agg, pre=pre, out=out)
    return out


def conv1d(x, w, b, stride):
    """nn.Conv1d (valid padding): x [N, Cin, L], w [Cout, Cin, K] -> [N, Cout, (L-K)//stride+1]."""
    n, cin, length = x.shape
    cout, _, k = w.shape
    lo = (length - k) // stride + 1
    idx = np.arange(lo)[:, None] * stride + np.arange(k)[None, :]
    win = x[:, :, idx]                      # [N, Cin, Lo, K]
    return np.einsum('nclk,ock->nol', win, w, optimize=True) + b[None, :, None]


def lem_forward(inputs, weights, weights_lin_z, bias, bias_lin_z, dt=1.0, states=None, return_state=False):
    """LEM recurrence behind `lem_cuda.forward` (call site experiments/models_gnn.py:285-342; the
    extension's source is absent, so this follows the published LEM cell):
      X = [y, x_t]; g = X W^T + b split in 3; dt_bar = dt s(g1); dt_ = dt s(g2);
      z <- (1-dt_) z + dt_ tanh(g3); X2 = [z, x_t]; y <- (1-dt_bar) y + dt_bar tanh(X2 Wz^T + bz).
    inputs [T, N, ninp]; returns all_y[-1] (LEM.forward, :340-342).  `states` = (y0, z0) as LEMcuda.forward takes them
    (:325-332; the stateful LEMS carries (all_y[-1], all_z[-1]) from call to call, :345-357); return_state adds all_z[-1].
    PARITY UNPINNED."""
    t_len, n, _ = inputs.shape
    nh = weights_lin_z.shape[0]
    y = np.zeros((n, nh), dtype=inputs.dtype) if states is None else np.asarray(states[0], dtype=inputs.dtype)
    z = np.zeros((n, nh), dtype=inputs.dtype) if states is None else np.asarray(states[1], dtype=inputs.dtype)
    for t in range(t_len):
        g = np.concatenate((y, inputs[t]), axis=1) @ weights.T + bias
        dt_bar = dt * sigmoid(g[:, :nh])
        dt_ = dt * sigmoid(g[:, nh:2 * nh])
        z = (1.0 - dt_) * z + dt_ * np.tanh(g[:, 2 * nh:])
        lin = np.concatenate((z, inputs[t]), axis=1) @ weights_lin_z.T + bias_lin_z
        y = (1.0 - dt_bar) * y + dt_bar * np.tanh(lin)
    return (y, z) if return_state else y


def lstm_forward(inputs, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTM(ninp, nhid) with zero initial state, output[-1] (experiments/models_gnn.py:758-767): gates in the order
    i, f, g, o;  c <- s(f) c + s(i) tanh(g);  h <- s(o) tanh(c).  inputs [T, N, ninp]."""
    t_len, n, _ = inputs.shape
    nh = w_hh.shape[1]
    h = np.zeros((n, nh), dtype=inputs.dtype)
    c = np.zeros((n, nh), dtype=inputs.dtype)
    for t in range(t_len):
        g = inputs[t] @ w_ih.T + b_ih + h @ w_hh.T + b_hh
        c = sigmoid(g[:, nh:2 * nh]) * c + sigmoid(g[:, :nh]) * np.tanh(g[:, 2 * nh:3 * nh])
        h = sigmoid(g[:, 3 * nh:]) * np.tanh(c)
    return h


# --------------------------------------------------------------------------------------------
# solver forward passes
# --------------------------------------------------------------------------------------------
KINDS_1D = ('MP_PDE_Solver', 'MP_PDE_SolverGated', 'MP_PDE_SolverLEMLinGated', 'MP_PDE_SolverLEMLin', 'MSSMP_PDE_Solver',
            'MP_PDE_SolverLEMLinGatedSave', 'MP_PDE_SolverLEMLinGatedGLU',
            'MP_PDE_SolverLSTMLin', 'MP_PDE_SolverLSTMLinGated')
KINDS_2D = ('MP_PDE_Solver2D', 'MP_PDE_Solver2DGated', 'MP_PDE_Solver2DLEMLinGated', 'MP_PDE_Solver2DLEMLin',
            'MP_PDE_Solver2DLEMLinG2', 'MP_PDE_Solver2DLSTMLin', 'MP_PDE_Solver2DLSTMLinGated', 'MP_PDE_Solver2DLEMLinGatedGLU')

_DECODER = {  # time_window -> (k1, stride1, k2); experiments/models_gnn.py:210-224, models_gnn2D.py:79-88
    20: (15, 4, 10), 25: (16, 3, 14), 50: (12, 2, 10)}


def build_variables(kind, data, pde, eq_variables):
    """Feature preparation shared by all solvers.  1-D: experiments/models_gnn.py:239-266;
    2-D: experiments/models_gnn2D.py:103-116 (NB the reference divides data.a, not data.b, for 'b')."""
    pos = np.asarray(data.pos)
    pos_x = pos[:, 1][:, None] / pde.L
    pos_t = pos[:, 0][:, None] / pde.tmax
    cols = [pos_t]
    if kind in KINDS_1D:
        for k in ('alpha', 'beta', 'gamma'):
            if k in eq_variables:
                cols.append(np.asarray(getattr(data, k)) / eq_variables[k])
        for k in ('bc_left', 'bc_right'):
            if k in eq_variables:
                cols.append(np.asarray(getattr(data, k)))
        for k in ('c', 'D', 'r'):
            if k in eq_variables:
                cols.append(np.asarray(getattr(data, k)) / eq_variables[k])
    else:
        if 'a' in eq_variables:
            cols.append(np.asarray(data.a) / eq_variables['a'])
        if 'b' in eq_variables:
            cols.append(np.asarray(data.a) / eq_variables['b'])   # sic: models_gnn2D.py:116
    return pos_x, pos_t, np.concatenate(cols, axis=-1)


def solver_forward(kind, sd, data, pde, time_window, eq_variables, hidden_layer=6, dtype=np.float64,
                   parts=False, decoder_diff=False, lem_states=None):
    """forward(data) of the in-scope solver classes.
    MP_PDE_Solver               experiments/models_gnn.py:229-281
    MP_PDE_SolverGated          experiments/models_gnn.py:1162-1218
    MP_PDE_SolverLEMLinGated    experiments/models_gnn.py:1315-1377
    MP_PDE_Solver2D             experiments/models_gnn2D.py:93-141
    MP_PDE_Solver2DGated        experiments/models_gnn2D.py:238-288
    MP_PDE_Solver2DLEMLinGated  experiments/models_gnn2D.py:396-458
    MP_PDE_SolverLEMLin         experiments/models_gnn.py:696-756    (LEM encoder + plain GNN_Layer stack; train.py 'LEM')
    MP_PDE_Solver2DLEMLin       experiments/models_gnn2D.py:1003-1057 (same, 2-D; train.py 'LEM2D')
    MP_PDE_Solver2DLEMLinG2     experiments/models_gnn2D.py:565-620   (gradient-gated blend; train.py 'MSG2-PDE2D')
    MP_PDE_SolverLEMLinGatedSave  experiments/models_gnn.py:1747-1905 (and save_state=True of the 2-D class): pass a dict as
                                `lem_states`; it carries the LEM states from call to call ({} or reset = a new sequence)
    MP_PDE_SolverLEMLinGatedGLU   experiments/models_gnn.py:1379-1523   (hidden width 164, gated CNN decoder; train.py 'MSGMP-PDE')
    MP_PDE_Solver2DLEMLinGatedGLU experiments/models_gnn2D.py:1198-1366 (same, 2-D; train.py 'MSGMP-PDE2D')
    MSSMP_PDE_Solver            experiments/models_gnn.py:1721-1745   (two MSMP-PDE networks `diff.*`, `scale.*`, each returning
                                its decoder output (:1679-1682, decoder_diff=True); train.py 'MSSMP-PDE')
    `sd` maps the reference's state_dict key names to arrays.  Row S1."""
    sd = {k: np.asarray(v, dtype=dtype) for k, v in sd.items()}
    tw = time_window
    if kind == 'MSSMP_PDE_Solver':
        sub = lambda pre: solver_forward('MP_PDE_SolverLEMLinGated', {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)},
                                         data, pde, time_window, eq_variables, hidden_layer, dtype, decoder_diff=True)
        scale, diff = sub('scale.'), sub('diff.')
        u = np.asarray(data.x, dtype=dtype)
        dt = np.cumsum(np.ones(time_window, dtype=dtype) * pde.dt)
        return (1.0 - scale) * u[:, -1:] + dt[None, :] * (scale * diff)
    two_d = kind in KINDS_2D
    u = np.asarray(data.x, dtype=dtype)
    ei = np.asarray(data.edge_index)
    batch = np.asarray(data.batch)
    pos_x, pos_t, variables = build_variables(kind, data, pde, eq_variables)
    pos_x, pos_t, variables = pos_x.astype(dtype), pos_t.astype(dtype), variables.astype(dtype)
    dt = np.cumsum(np.ones(tw, dtype=dtype) * pde.dt)

    gated = 'Gated' in kind
    g2 = kind.endswith('G2')
    if 'LEM' in kind or 'LSTM' in kind:
        if two_d:   # models_gnn2D.py:421-436
            ts = dt[None, :] + pos_t
            steps = [np.concatenate((pos_x, u[:, t:t + 1], u[:, t + tw:t + tw + 1], ts[:, t:t + 1],
                                     variables[:, 1:]), axis=-1) for t in range(tw)]
        else:       # models_gnn.py:1356-1363
            steps = [np.concatenate((pos_x, u[:, t:t + 1], variables), axis=-1) for t in range(u.shape[1])]
        lem_in = np.stack(steps, axis=0)
        if 'LSTM' in kind:      # the LSTM ablations (models_gnn.py:770-1065, models_gnn2D.py:622-918): same inputs, nn.LSTM encoder
            r = 'embedding_lstm.rnn.'
            h = lstm_forward(lem_in, sd[r + 'weight_ih_l0'], sd[r + 'weight_hh_l0'], sd[r + 'bias_ih_l0'], sd[r + 'bias_hh_l0'])
            mlp = 'lstmoutput_mlp'
        else:
            h = lem_forward(lem_in, sd['embedding_lem.rnn.weights'], sd['embedding_lem.rnn.weights_lin_z'],
                            sd['embedding_lem.rnn.bias'], sd['embedding_lem.rnn.bias_lin_z'], 1.0,
                            states=None if lem_states is None else lem_states.get('states'), return_state=lem_states is not None)
            if lem_states is not None:      # the Save variants: (all_y[-1], all_z[-1]) go into the next call (models_gnn.py:350-353)
                lem_states['states'] = h
                h = h[0]
            mlp = 'lemoutput_mlp'
        h = swish(linear(h, sd[mlp + '.0.weight'], sd[mlp + '.0.bias']))
        h = swish(linear(h, sd[mlp + '.2.weight'], sd[mlp + '.2.bias']))
    else:
        node_input = np.concatenate((u, pos_x, variables), axis=-1)
        h = swish(linear(node_input, sd['embedding_mlp.0.weight'], sd['embedding_mlp.0.bias']))
        h = swish(linear(h, sd['embedding_mlp.2.weight'], sd['embedding_mlp.2.bias']))
    h_enc = h

    hs = []
    for i in range(hidden_layer):
        if g2:      # models_gnn2D.py:606-611: tau = tanh(mean over the out-edges (j -> i) of j of |tau_j - tau_i|^2), tau = Swish(gate layer)
            pg = layer_params(sd, f'gnn_layers_gate.{i}.')
            pm = layer_params(sd, f'gnn_layers.{i}.')
            tau = swish(mp_layer(pg, h, u, pos_x, variables, ei, batch, lin=True))
            d2 = np.abs(tau[ei[0]] - tau[ei[1]]) ** 2
            acc = np.zeros_like(tau)
            np.add.at(acc, ei[0], d2)
            cnt = np.bincount(ei[0], minlength=tau.shape[0]).astype(dtype)
            tau = np.tanh(acc / np.maximum(cnt, 1.0)[:, None])        # torch_scatter mean: empty segments -> 0
            h = (1.0 - tau) * h + tau * swish(mp_layer(pm, h, u, pos_x, variables, ei, batch, lin=True))
        elif gated:   # models_gnn.py:1204-1207 / 1365-1368; models_gnn2D.py:266-269 / 438-441.  Row L5.
            pg = layer_params(sd, f'gnn_layers_gate.{i}.')
            pm = layer_params(sd, f'gnn_layers.{i}.')
            tau = sigmoid(mp_layer(pg, h, u, pos_x, variables, ei, batch, lin=True))
            h = (1.0 - tau) * h + tau * swish(mp_layer(pm, h, u, pos_x, variables, ei, batch, lin=True))
        else:       # models_gnn.py:271-272
            pm = layer_params(sd, f'gnn_layers.{i}.')
            h = mp_layer(pm, h, u, pos_x, variables, ei, batch, lin=False)
        hs.append(h)

    if kind.endswith('GLU'):
        # the GLU classes (hidden width 164): a gated pair of CNNs on the two halves of the hidden state; `scale` enters WITHOUT a
        # sigmoid.  1-D: experiments/models_gnn.py:1511-1521; 2-D: models_gnn2D.py:1349-1366 (double_mlp first, halves of each component)
        dec = lambda x, pre: conv1d(swish(conv1d(x, sd[pre + '.0.weight'], sd[pre + '.0.bias'], 2)), sd[pre + '.2.weight'], sd[pre + '.2.bias'], 1)
        if two_d:
            hd = swish(linear(h, sd['double_mlp.0.weight'], sd['double_mlp.0.bias'])).reshape(-1, 2, h.shape[1])
            half = hd.shape[2] // 2
            diff, scale = dec(hd[:, :, half:], 'output_mlp_diff'), dec(hd[:, :, :half], 'output_mlp_gate')
            out = ((1.0 - scale) * u.reshape(-1, 2, tw) + dt[None, None, :] * scale * diff).reshape(-1, 2 * tw)
        else:
            half = h.shape[1] // 2
            scale = dec(h[:, None, :half], 'output_mlp_gate')[:, 0, :]
            diff = dec(h[:, None, half:], 'output_mlp_diff')[:, 0, :]
            out = (1.0 - scale) * u[:, -1:] + dt[None, :] * (scale * diff)
        if parts:
            return SimpleNamespace(out=out, h_enc=h_enc, hs=hs)
        return out
    k1, s1, k2 = _DECODER[tw]
    if two_d:       # models_gnn2D.py:125-141
        hd = swish(linear(h, sd['double_mlp.0.weight'], sd['double_mlp.0.bias'])).reshape(-1, 2, h.shape[1])
        diff = conv1d(swish(conv1d(hd, sd['output_mlp.0.weight'], sd['output_mlp.0.bias'], s1)),
                      sd['output_mlp.2.weight'], sd['output_mlp.2.bias'], 1)
        out = (u.reshape(-1, 2, tw) + dt[None, None, :] * diff).reshape(-1, 2 * tw)
    else:           # models_gnn.py:275-279
        diff = conv1d(swish(conv1d(h[:, None, :], sd['output_mlp.0.weight'], sd['output_mlp.0.bias'], s1)),
                      sd['output_mlp.2.weight'], sd['output_mlp.2.bias'], 1)[:, 0, :]
        out = diff if decoder_diff else u[:, -1:] + dt[None, :] * diff
    if parts:
        return SimpleNamespace(out=out, h_enc=h_enc, hs=hs)
    return out


# --------------------------------------------------------------------------------------------
# graph construction (common/utils.py GraphCreator)
# --------------------------------------------------------------------------------------------
def torch_linspace(start, end, steps):
    """torch.linspace in float64 (used for the time axis at common/utils.py:340,456).  ATen fills
    the first half as start + i*step and the second half as end - (steps-1-i)*step, which differs
    from numpy.linspace in the last bit for some i; the restatement keeps ATen's formula so that
    pos[:,0] is bit-identical to the reference's."""
    step = (float(end) - float(start)) / (steps - 1)
    i = np.arange(steps, dtype=np.float64)
    half = steps // 2
    return np.where(np.arange(steps) < half, float(start) + step * i, float(end) - step * (steps - 1 - i))


def radius_graph(x, r, batch, max_num_neighbors=32):
    """torch_cluster.radius_graph(x, r, batch, loop=False) as called at common/utils.py:368:
    pair kept iff same graph, i != j, (x_i - x_j)^2 < r*r (float64, strict), at most 32 sources per
    target (lowest index first).  Canonical order: ascending target, then ascending source.  Row G1."""
    x = np.asarray(x, dtype=np.float64).reshape(len(x), -1)
    batch = np.asarray(batch)
    src, dst = [], []
    r2 = r * r
    start = 0
    n = len(x)
    while start < n:
        end = start
        while end < n and batch[end] == batch[start]:
            end += 1
        xb = x[start:end]
        d2 = ((xb[:, None, :] - xb[None, :, :]) ** 2).sum(-1)
        m = d2 < r2
        np.fill_diagonal(m, False)
        for ti in range(end - start):
            js = np.nonzero(m[ti])[0][:max_num_neighbors]
            src.append(js + start)
            dst.append(np.full(len(js), ti + start))
        start = end
    return np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)


def knn_graph(x, k, batch):
    """torch_cluster.knn_graph(x, k, batch, loop=False) as called at common/utils.py:377,380: per
    target i the k nearest same-graph nodes j != i (float64 squared distance), ascending distance,
    ties -> lower index; grouped by ascending target.  Row G1."""
    x = np.asarray(x, dtype=np.float64).reshape(len(x), -1)
    batch = np.asarray(batch)
    src, dst = [], []
    start = 0
    n = len(x)
    while start < n:
        end = start
        while end < n and batch[end] == batch[start]:
            end += 1
        xb = x[start:end]
        d2 = ((xb[:, None, :] - xb[None, :, :]) ** 2).sum(-1)
        np.fill_diagonal(d2, np.inf)
        order = np.argsort(d2, axis=1, kind='stable')[:, :k]
        for ti in range(end - start):
            js = order[ti]
            js = js[np.isfinite(d2[ti, js])]
            src.append(js + start)
            dst.append(np.full(len(js), ti + start))
        start = end
    return np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64)


def create_data(datapoints, steps, tw):
    """GraphCreator.create_data, common/utils.py:300-317: data = dp[step-tw:step], labels = dp[step:step+tw]."""
    data = np.stack([dp[s - tw:s] for dp, s in zip(datapoints, steps)])
    labels = np.stack([dp[s:s + tw] for dp, s in zip(datapoints, steps)])
    return data, labels


def _flatten_uy(block, is_ad):
    """common/utils.py:350-357: non-AD: [tw, nx] -> [nx, tw]; AD: [tw, 2, nx] -> [nx, 2*tw] component-major."""
    if is_ad:
        return np.transpose(block, (1, 0, 2)).reshape(-1, block.shape[-1]).T
    return block.T


def create_graph(pde_name, pde, neighbors, tw, data, labels, x, variables, steps, unstructured=False):
    """GraphCreator.create_graph, common/utils.py:320-428.  `x` is [B, nx] (only x[0] is used, as in
    the reference); `variables` maps names to length-B sequences.  Rows G1 + G2."""
    nt, nx = pde.grid_size[0], pde.grid_size[1]
    t = torch_linspace(pde.tmin, pde.tmax, nt)
    x0 = np.asarray(x[0], dtype=np.float64)
    is_ad = pde_name == 'AD'
    bsz = len(data)
    u = np.concatenate([_flatten_uy(np.asarray(d), is_ad) for d in data])
    y = np.concatenate([_flatten_uy(np.asarray(l), is_ad) for l in labels])
    x_pos = np.tile(x0, bsz)
    t_pos = np.repeat(t[np.asarray(steps)], nx)
    batch = np.repeat(np.arange(bsz), nx).astype(np.int64)
    if pde_name in ('CE', 'KF', 'KS', 'AD'):
        if is_ad and unstructured:       # common/utils.py:343-346,376-377
            xx = 2 * np.pi * x0 / (x0.max() - 1e-3)
            x_per = np.tile(np.stack([np.cos(xx), np.sin(xx)], 1), (bsz, 1))
            edge_index = knn_graph(x_per, neighbors, batch)
        else:                            # common/utils.py:366-368
            dx = x0[1] - x0[0]
            edge_index = radius_graph(x_pos, neighbors * dx + 0.0001, batch)
    elif pde_name == 'WE':               # common/utils.py:379-380
        edge_index = knn_graph(x_pos, neighbors, batch)
    else:
        raise ValueError(pde_name)
    g = SimpleNamespace(x=u, y=y, edge_index=edge_index, batch=batch,
                        pos=np.stack([t_pos, x_pos], 1))
    col = lambda name, sign=1.0: (sign * np.asarray(variables[name], dtype=np.float64))[batch][:, None]
    if pde_name == 'CE':                 # common/utils.py:388-397 (beta negated, :392)
        g.alpha, g.beta, g.gamma = col('alpha'), col('beta', -1.0), col('gamma')
    elif pde_name == 'KF':
        g.r, g.D = col('r'), col('D')
    elif pde_name == 'WE':
        g.bc_left, g.bc_right, g.c = col('bc_left'), col('bc_right'), col('c')
    elif is_ad:
        g.a, g.b = col('a'), col('b')
    return g


def create_next_graph(pde_name, pde, tw, graph, pred, labels, steps):
    """GraphCreator.create_next_graph, common/utils.py:431-471: x <- pred (the concatenate-then-slice
    of :448-452 keeps exactly `pred`), y <- new labels, pos[:,0] <- t[step]; rest reused.  Row R1."""
    nt, nx = pde.grid_size[0], pde.grid_size[1]
    t = torch_linspace(pde.tmin, pde.tmax, nt)
    is_ad = pde_name == 'AD'
    keep = 2 * tw if is_ad else tw
    graph.x = np.concatenate((graph.x, pred), 1)[:, keep:]
    graph.y = np.concatenate([_flatten_uy(np.asarray(l), is_ad) for l in labels])
    graph.pos = graph.pos.copy()
    graph.pos[:, 0] = np.repeat(t[np.asarray(steps)], nx)
    return graph


def rollout(kind, sd, graph, pde_name, pde, tw, eq_variables, hidden_layer, traj, first_step, n_steps,
            dtype=np.float64):
    """Unrolled evaluation, experiments/train_helper.py:233-273: pred = model(graph); then repeatedly
    create_next_graph + model(graph).  Returns the list of predictions."""
    preds = [solver_forward(kind, sd, graph, pde, tw, eq_variables, hidden_layer, dtype)]
    step = first_step
    for _ in range(n_steps):
        step += tw
        _, labels = create_data(traj, [step] * len(traj), tw)
        graph = create_next_graph(pde_name, pde, tw, graph, preds[-1], labels, [step] * len(traj))
        preds.append(solver_forward(kind, sd, graph, pde, tw, eq_variables, hidden_layer, dtype))
    return preds
