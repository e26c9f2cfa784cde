"""CPU ORACLE, torch edition.  TEST INFRASTRUCTURE ONLY (same rules as msmp_oracle.py).

The same restatement as `msmp_oracle.solver_forward`, written with torch CPU float64 ops so that it
runs multi-threaded the way the reference itself does on `--device=cpu` (the reference IS PyTorch;
numpy's elementwise ops are single-threaded and would understate the CPU).  Used by bench.py as the
`cpu_baseline` ("port") and pinned in tests/test_oracle_golden.py against the same golden vectors
and against the numpy oracle.  Citations as in msmp_oracle.py (paths relative to /root/reference).
"""
import torch
import torch.nn.functional as F

from . import msmp_oracle as O


def swish(x):                                   # experiments/models_gnn.py:20-21
    return x * torch.sigmoid(x)


def _seg_mean(x, index, n):                     # PyG aggr='mean' / scatter-mean
    out = torch.zeros(n, x.shape[1], dtype=x.dtype, device=x.device).index_add_(0, index, x)
    cnt = torch.zeros(n, dtype=x.dtype, device=x.device).index_add_(0, index, torch.ones(index.numel(), dtype=x.dtype, device=x.device))
    return out / cnt.clamp(min=1)[:, None]


def instance_norm(x, batch, b, eps=1e-5):       # PyG InstanceNorm, experiments/models_gnn.py:59,66
    mean = _seg_mean(x, batch, b)
    xc = x - mean[batch]
    var = _seg_mean(xc * xc, batch, b)
    return xc / torch.sqrt(var + eps)[batch]


def mp_layer(sd, prefix, x, u, pos, variables, ei, batch, b, lin):
    """experiments/models_gnn.py:61-86 / 124-149."""
    g = lambda k: sd[prefix + k]
    j, i = ei[0], ei[1]
    cat = torch.cat((x[i], x[j], u[i] - u[j], pos[i] - pos[j], variables[i]), -1)
    m = swish(F.linear(cat, g('message_net_1.0.weight'), g('message_net_1.0.bias')))
    m = swish(F.linear(m, g('message_net_2.0.weight'), g('message_net_2.0.bias')))
    agg = _seg_mean(m, i, x.shape[0])
    upd = swish(F.linear(torch.cat((x, agg, variables), -1), g('update_net_1.0.weight'), g('update_net_1.0.bias')))
    upd = F.linear(upd, g('update_net_2.0.weight'), g('update_net_2.0.bias'))
    pre = upd if lin else x + swish(upd)
    return instance_norm(pre, batch, b)


def lem_forward(inputs, w, wz, bias, bz, dt=1.0, states=None):
    """See msmp_oracle.lem_forward (PARITY UNPINNED)."""
    t_len, n, _ = inputs.shape
    nh = wz.shape[0]
    y, z = (torch.zeros(n, nh, dtype=inputs.dtype, device=inputs.device), torch.zeros(n, nh, dtype=inputs.dtype, device=inputs.device)) if states is None else states
    for t in range(t_len):
        g = F.linear(torch.cat((y, inputs[t]), 1), w, bias)
        dt_bar = dt * torch.sigmoid(g[:, :nh])
        dt_ = dt * torch.sigmoid(g[:, nh:2 * nh])
        z = (1.0 - dt_) * z + dt_ * torch.tanh(g[:, 2 * nh:])
        y = (1.0 - dt_bar) * y + dt_bar * torch.tanh(F.linear(torch.cat((z, inputs[t]), 1), wz, bz))
    return y


def solver_forward(kind, sd, data, pde, time_window, eq_variables, hidden_layer=6, as_numpy=True, decoder_diff=False,
                   lem_initial_states=None, dtype=torch.float64, device='cpu'):
    """forward(data) of the six in-scope solver classes (see msmp_oracle.solver_forward for the line map).
    `sd` values may be float64 torch tensors that require grad (as_numpy=False keeps the autograd graph:
    used to check the product's gradients).  dtype / device: float64 on the CPU is the oracle; the tests also evaluate it
    in float32 on the GPU (PyTorch-ROCm, rocBLAS GEMMs) as a second measurement of what float32 arithmetic can deliver."""
    t64 = lambda a: (a if torch.is_tensor(a) else torch.as_tensor(a)).to(device=device, dtype=dtype)
    sd = {k: t64(v) for k, v in sd.items()}
    tw = time_window
    if kind == 'MSSMP_PDE_Solver':
        sub = lambda pre: solver_forward('MP_PDE_SolverLEMLinGated', {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)},
                                         data, pde, tw, eq_variables, hidden_layer, as_numpy=False, decoder_diff=True, dtype=dtype, device=device)
        scale, diff = sub('scale.'), sub('diff.')
        dt = torch.cumsum(torch.ones(tw, dtype=dtype, device=device) * pde.dt, 0)
        out = (1.0 - scale) * t64(data.x)[:, -1:] + dt[None, :] * (scale * diff)
        return out.detach().cpu().numpy() if as_numpy else out
    two_d = kind in O.KINDS_2D
    u = t64(data.x)
    ei = torch.as_tensor(data.edge_index).long().to(device)
    batch = torch.as_tensor(data.batch).long().to(device)
    b = int(batch.max()) + 1
    pos_x, pos_t, variables = (t64(a) for a in O.build_variables(kind, data, pde, eq_variables))
    dt = torch.cumsum(torch.ones(tw, dtype=dtype, device=device) * pde.dt, 0)
    if 'LEM' in kind or 'LSTM' in kind:
        if two_d:
            ts = dt[None, :] + pos_t
            steps = [torch.cat((pos_x, u[:, t:t + 1], u[:, t + tw:t + tw + 1], ts[:, t:t + 1], variables[:, 1:]), -1)
                     for t in range(tw)]
        else:
            steps = [torch.cat((pos_x, u[:, t:t + 1], variables), -1) for t in range(u.shape[1])]
        if 'LSTM' in kind:
            r = 'embedding_lstm.rnn.'
            x, nh = torch.stack(steps, 0), sd[r + 'weight_hh_l0'].shape[1]
            h = torch.zeros(x.shape[1], nh, dtype=x.dtype, device=x.device)
            c = torch.zeros_like(h)
            for t in range(x.shape[0]):
                g = F.linear(x[t], sd[r + 'weight_ih_l0'], sd[r + 'bias_ih_l0']) + F.linear(h, sd[r + 'weight_hh_l0'], sd[r + 'bias_hh_l0'])
                c = torch.sigmoid(g[:, nh:2 * nh]) * c + torch.sigmoid(g[:, :nh]) * torch.tanh(g[:, 2 * nh:3 * nh])
                h = torch.sigmoid(g[:, 3 * nh:]) * torch.tanh(c)
            mlp = 'lstmoutput_mlp'
        else:
            h = lem_forward(torch.stack(steps, 0), sd['embedding_lem.rnn.weights'], sd['embedding_lem.rnn.weights_lin_z'],
                            sd['embedding_lem.rnn.bias'], sd['embedding_lem.rnn.bias_lin_z'], 1.0,
                            states=None if lem_initial_states is None else tuple(t64(a) for a in lem_initial_states))
            mlp = 'lemoutput_mlp'
        h = swish(F.linear(h, sd[mlp + '.0.weight'], sd[mlp + '.0.bias']))
        h = swish(F.linear(h, sd[mlp + '.2.weight'], sd[mlp + '.2.bias']))
    else:
        h = swish(F.linear(torch.cat((u, pos_x, variables), -1), sd['embedding_mlp.0.weight'], sd['embedding_mlp.0.bias']))
        h = swish(F.linear(h, sd['embedding_mlp.2.weight'], sd['embedding_mlp.2.bias']))
    for i in range(hidden_layer):
        if kind.endswith('G2'):
            tau = swish(mp_layer(sd, f'gnn_layers_gate.{i}.', h, u, pos_x, variables, ei, batch, b, True))
            d2 = (tau[ei[0]] - tau[ei[1]]).abs() ** 2
            cnt = torch.bincount(ei[0], minlength=tau.shape[0]).clamp(min=1).to(tau.dtype)
            tau = torch.tanh(torch.zeros_like(tau).index_add_(0, ei[0], d2) / cnt[:, None])
            h = (1.0 - tau) * h + tau * swish(mp_layer(sd, f'gnn_layers.{i}.', h, u, pos_x, variables, ei, batch, b, True))
        elif 'Gated' in kind:
            tau = torch.sigmoid(mp_layer(sd, f'gnn_layers_gate.{i}.', h, u, pos_x, variables, ei, batch, b, True))
            h = (1.0 - tau) * h + tau * swish(mp_layer(sd, f'gnn_layers.{i}.', h, u, pos_x, variables, ei, batch, b, True))
        else:
            h = mp_layer(sd, f'gnn_layers.{i}.', h, u, pos_x, variables, ei, batch, b, False)
    if kind.endswith('GLU'):        # msmp_oracle.solver_forward, GLU branch
        dec = lambda x, pre: F.conv1d(swish(F.conv1d(x, sd[pre + '.0.weight'], sd[pre + '.0.bias'], stride=2)), sd[pre + '.2.weight'], sd[pre + '.2.bias'])
        if two_d:
            hd = swish(F.linear(h, sd['double_mlp.0.weight'], sd['double_mlp.0.bias'])).reshape(-1, 2, h.shape[1])
            half = hd.shape[2] // 2
            diff, scale = dec(hd[:, :, half:], 'output_mlp_diff'), dec(hd[:, :, :half], 'output_mlp_gate')
            out = ((1.0 - scale) * u.reshape(-1, 2, tw) + dt[None, None, :] * scale * diff).reshape(-1, 2 * tw)
        else:
            half = h.shape[1] // 2
            scale = dec(h[:, None, :half], 'output_mlp_gate')[:, 0, :]
            diff = dec(h[:, None, half:], 'output_mlp_diff')[:, 0, :]
            out = (1.0 - scale) * u[:, -1:] + dt[None, :] * (scale * diff)
        return out.detach().cpu().numpy() if as_numpy else out
    k1, s1, k2 = O._DECODER[tw]
    if two_d:
        hd = swish(F.linear(h, sd['double_mlp.0.weight'], sd['double_mlp.0.bias'])).reshape(-1, 2, h.shape[1])
        diff = F.conv1d(swish(F.conv1d(hd, sd['output_mlp.0.weight'], sd['output_mlp.0.bias'], stride=s1)),
                        sd['output_mlp.2.weight'], sd['output_mlp.2.bias'])
        out = (u.reshape(-1, 2, tw) + dt[None, None, :] * diff).reshape(-1, 2 * tw)
    else:
        diff = F.conv1d(swish(F.conv1d(h[:, None, :], sd['output_mlp.0.weight'], sd['output_mlp.0.bias'], stride=s1)),
                        sd['output_mlp.2.weight'], sd['output_mlp.2.bias'])[:, 0, :]
        out = diff if decoder_diff else u[:, -1:] + dt[None, :] * diff
    return out.detach().cpu().numpy() if as_numpy else out
