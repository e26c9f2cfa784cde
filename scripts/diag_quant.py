"""CPU diagnostic (no GPU): which operand of the fp16-split GEMMs sets the full-depth error?
The float64 oracle's gated layer pair is re-evaluated with the 2-way fp16 split's operand rounding (hi + lo of a value scaled
by a power of two: 22 significant bits) applied to ONE operand class at a time, in the factorised form the kernels use
(P_i + Q_j), everything else in float64.  Reported: rms error of the layer output (after InstanceNorm + blend) against the
exact float64 layer, for layer pairs 0..2 fed with the exact input ("fresh" error, as scripts/diag_lolo.py), next to a plain
float32 evaluation.  Reference formulas: experiments/models_gnn.py:124-149, 1365-1368.
    python scripts/diag_quant.py [E2|WE3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import msmp_pde_amd as mp
from helpers import synthetic_case
from oracle import msmp_oracle as O

TW = 25


def q22(x, scale):
    """hi + lo of x * scale in fp16 (round to nearest even), back in float64; saturating like v_med3 at +-65504."""
    xs = np.clip(np.asarray(x, dtype=np.float64) * scale, -65504.0, 65504.0)
    hi = xs.astype(np.float16).astype(np.float64)
    lo = (xs - hi).astype(np.float16).astype(np.float64)
    return (hi + lo) / scale


def q33(x, scale):
    xs = np.asarray(x, dtype=np.float64) * scale
    hi = xs.astype(np.float16).astype(np.float64)
    lo = (xs - hi).astype(np.float16).astype(np.float64)
    lo2 = (xs - hi - lo).astype(np.float16).astype(np.float64)
    return (hi + lo + lo2) / scale


def qw(w):
    s = 2.0 ** np.floor(np.log2(31.999 / np.abs(w).max()))
    return q22(w, s)


def f32(x):
    return np.asarray(x, dtype=np.float32).astype(np.float64)


def layer(p, h, u, pos, var, ei, batch, Q, center=False):
    """GNN_LayerLin forward in the kernels' factorised form; Q: dict of operand class -> quantiser (identity when absent)."""
    g = lambda k: Q.get(k, lambda x: x)
    j, i = ei[0], ei[1]
    tw, nv = u.shape[1], var.shape[1]
    w1 = g('w')(p.w1) if 'w' in Q else p.w1
    w2 = g('w')(p.w2) if 'w' in Q else p.w2
    w3 = g('w')(p.w3) if 'w' in Q else p.w3
    w4 = g('w')(p.w4) if 'w' in Q else p.w4
    hq = g('h_msg')(h)
    feat = np.concatenate((u, pos, var), 1)
    fq = g('feat')(feat)
    wa, wb, wt = w1[:, :128], w1[:, 128:256], w1[:, 256:]
    P = hq @ wa.T + fq @ wt.T + p.b1
    fqq = fq.copy(); fqq[:, tw + 1:] = 0.0
    Qn = hq @ wb.T - fqq @ wt.T
    P, Qn = g('pq')(P), g('pq')(Qn)
    a1 = g('act1')(O.swish(P[i] + Qn[j]))
    msg = O.swish(a1 @ w2.T + p.b2)
    agg = O.scatter_mean(g('msg')(msg), i, h.shape[0])
    ht, aggq = g('h_upd')(h), g('agg')(agg)
    z = O.swish(ht @ w3[:, :128].T + aggq @ w3[:, 128:256].T + g('feat')(var) @ w3[:, 256:].T + p.b3)
    return g('act2')(z) @ w4.T + p.b4


def pair(sd, i, h, u, pos, var, ei, batch, Q):
    pg, pm = O.layer_params(sd, f'gnn_layers_gate.{i}.'), O.layer_params(sd, f'gnn_layers.{i}.')
    tau = O.sigmoid(O.instance_norm(layer(pg, h, u, pos, var, ei, batch, Q), batch))
    return (1.0 - tau) * h + tau * O.swish(O.instance_norm(layer(pm, h, u, pos, var, ei, batch, Q), batch))


def main(exp):
    torch.manual_seed(3)
    case = synthetic_case(mp, exp, bsz=8, seed=11, device='cpu')
    kind = 'MP_PDE_SolverLEMLinGated'
    model = getattr(mp, kind)(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=6)
    sd = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    g = case.graph_np()
    r64 = O.solver_forward(kind, sd, g, case.pde, TW, case.eqv, 6, parts=True)
    r32 = O.solver_forward(kind, sd, g, case.pde, TW, case.eqv, 6, dtype=np.float32, parts=True)
    u = np.asarray(g.x, dtype=np.float64)
    pos_x, pos_t, var = O.build_variables(kind, g, case.pde, case.eqv)
    ei, batch = np.asarray(g.edge_index), np.asarray(g.batch)
    NODE, ACT = 256.0, 64.0
    variants = {
        'exact (factorised, float64)': {},
        'h rows in message proj (22 bit)': {'h_msg': lambda x: q22(x, NODE)},
        'h rows in update (22 bit)': {'h_upd': lambda x: q22(x, NODE)},
        'agg rows in update (22 bit)': {'agg': lambda x: q22(x, NODE)},
        'u|pos|vars rows (22 bit)': {'feat': lambda x: q22(x, NODE)},
        'weights (22 bit)': {'w': qw},
        'Swish(P+Q) operand (22 bit)': {'act1': lambda x: q22(x, ACT)},
        'update hidden operand (22 bit)': {'act2': lambda x: q22(x, ACT)},
        'all of the above': {'h_msg': lambda x: q22(x, NODE), 'h_upd': lambda x: q22(x, NODE), 'agg': lambda x: q22(x, NODE),
                             'feat': lambda x: q22(x, NODE), 'w': qw, 'act1': lambda x: q22(x, ACT), 'act2': lambda x: q22(x, ACT)},
        'all, node rows 33 bit': {'h_msg': lambda x: q33(x, NODE), 'h_upd': lambda x: q33(x, NODE), 'agg': lambda x: q33(x, NODE),
                                  'feat': lambda x: q33(x, NODE), 'w': qw, 'act1': lambda x: q22(x, ACT), 'act2': lambda x: q22(x, ACT)},
        'all operands rounded to float32 (24 bit)': {k: f32 for k in ('h_msg', 'h_upd', 'agg', 'feat', 'w', 'act1', 'act2', 'pq', 'msg')},
    }
    print(f'{exp}: |h_enc| max {np.abs(r64.h_enc).max():.3g}, per-graph std of h_enc (median over channels) '
          f'{np.median(r64.h_enc[:100].std(0)):.3g}')
    for li in range(3):
        hin = r64.hs[li - 1] if li else r64.h_enc
        ref = r64.hs[li]
        e32 = np.sqrt(np.mean((r32.hs[li] - r64.hs[li]) ** 2))
        print(f'layer pair {li}: |h_in| rms {np.sqrt(np.mean(hin ** 2)):.3g}, per-graph std (median channel) {np.median(hin[:100].std(0)):.3g}; '
              f'accumulated float32-oracle error rms {e32:.2e}')
        for name, Q in variants.items():
            out = pair(sd, li, hin, u, pos_x, var, ei, batch, Q)
            print(f'    {name:45s} rms {np.sqrt(np.mean((out - ref) ** 2)):.2e}  max {np.abs(out - ref).max():.2e}')


if __name__ == '__main__' and len(sys.argv) <= 2:
    main(sys.argv[1] if len(sys.argv) > 1 else 'E2')


# ---- second experiment: the fp32 ACCUMULATION of the split GEMMs (python scripts/diag_quant.py E2 acc) --------------------------
def split16(x, scale):
    xs = np.clip(np.asarray(x, dtype=np.float64) * scale, -65504.0, 65504.0)
    hi = xs.astype(np.float16).astype(np.float64)
    lo = (xs - hi).astype(np.float16).astype(np.float64)
    return hi, lo


def mfma_gemm(x, w, b, xscale, mode):
    """x [n, K] @ w[out, K].T + b the way the kernels do it: operands as fp16 pairs, K in steps of 16, each MFMA = exact sum of
    its 16 products added to the fp32 accumulator with ONE rounding.  mode: '3' three MFMAs per step into one accumulator (the
    shipped kernels); '1' one rounding per step (cross terms kept in a second accumulator, added once at the end);
    'x' cross terms of all steps first, then the hi*hi terms (no second accumulator); '0' float64 accumulation."""
    ws = 2.0 ** np.floor(np.log2(31.999 / np.abs(w).max()))
    xh, xl = split16(x, xscale)
    wh, wl = split16(w, ws)
    K = x.shape[1]
    pad = (-K) % 16
    if pad:
        z = lambda a: np.concatenate((a, np.zeros((a.shape[0], pad))), 1)
        xh, xl, wh, wl = z(xh), z(xl), z(wh), z(wl)
    f = lambda a: a.astype(np.float32).astype(np.float64)
    acc = f(np.broadcast_to(b * ws * xscale, (x.shape[0], w.shape[0])).copy())
    if mode == '0':
        return ((xh + xl) @ (wh + wl).T + b * ws * xscale) / (ws * xscale)
    cross = np.zeros_like(acc)
    steps = range(0, xh.shape[1], 16)
    if mode == 'x':
        for k in steps:
            s = slice(k, k + 16)
            acc = f(acc + xl[:, s] @ wh[:, s].T)
            acc = f(acc + xh[:, s] @ wl[:, s].T)
        for k in steps:
            s = slice(k, k + 16)
            acc = f(acc + xh[:, s] @ wh[:, s].T)
        return acc / (ws * xscale)
    for k in steps:
        s = slice(k, k + 16)
        t1, t2, t3 = xl[:, s] @ wh[:, s].T, xh[:, s] @ wl[:, s].T, xh[:, s] @ wh[:, s].T
        if mode == '3':
            acc = f(f(f(acc + t1) + t2) + t3)
        else:
            cross = f(f(cross + t1) + t2)
            acc = f(acc + t3)
    return f(acc + cross) / (ws * xscale) if mode == '1' else acc / (ws * xscale)


def layer_acc(p, h, u, pos, var, ei, batch, modes):
    """GNN_LayerLin in the kernels' factorised form with emulated accumulation; modes = dict GEMM name -> mode ('0' exact)."""
    j, i = ei[0], ei[1]
    tw = u.shape[1]
    feat = np.concatenate((u, pos, var), 1)
    fq = feat.copy(); fq[:, tw + 1:] = 0.0
    m = lambda k: modes.get(k, '0')
    P = mfma_gemm(np.concatenate((h, feat), 1), np.concatenate((p.w1[:, :128], p.w1[:, 256:]), 1), p.b1, 256.0, m('w1'))
    Qn = mfma_gemm(np.concatenate((h, -fq), 1), np.concatenate((p.w1[:, 128:256], p.w1[:, 256:]), 1), 0.0 * p.b1, 256.0, m('w1'))
    a1 = O.swish(P[i] + Qn[j])
    msg = O.swish(mfma_gemm(a1, p.w2, p.b2, 64.0, m('w2')))
    agg = O.scatter_mean(msg, i, h.shape[0])
    z = O.swish(mfma_gemm(np.concatenate((h, agg, var), 1), p.w3, p.b3, 256.0, m('w3')))
    return mfma_gemm(z, p.w4, p.b4, 64.0, m('w4'))


def main_acc(exp):
    torch.manual_seed(3)
    case = synthetic_case(mp, exp, bsz=8, seed=11, device='cpu')
    kind = 'MP_PDE_SolverLEMLinGated'
    model = getattr(mp, kind)(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=6)
    sd = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    g = case.graph_np()
    r64 = O.solver_forward(kind, sd, g, case.pde, TW, case.eqv, 6, parts=True)
    r32 = O.solver_forward(kind, sd, g, case.pde, TW, case.eqv, 6, dtype=np.float32, parts=True)
    u = np.asarray(g.x, dtype=np.float64)
    pos_x, pos_t, var = O.build_variables(kind, g, case.pde, case.eqv)
    ei, batch = np.asarray(g.edge_index), np.asarray(g.batch)
    variants = {'all exact accumulation (operands split)': {},
                'w1 (P, Q) 3 roundings / step': {'w1': '3'}, 'w2 3 / step': {'w2': '3'}, 'w3 3 / step': {'w3': '3'}, 'w4 3 / step': {'w4': '3'},
                'all four, 3 / step (= shipped kernels)': {k: '3' for k in ('w1', 'w2', 'w3', 'w4')},
                'all four, 1 / step (second accumulator)': {k: '1' for k in ('w1', 'w2', 'w3', 'w4')},
                'all four, cross terms first': {k: 'x' for k in ('w1', 'w2', 'w3', 'w4')},
                'w3, w4: 1 / step; w1, w2: 3 / step': {'w1': '3', 'w2': '3', 'w3': '1', 'w4': '1'},
                'w4: 1 / step; rest 3 / step': {'w1': '3', 'w2': '3', 'w3': '3', 'w4': '1'}}
    for li in range(2):
        hin = r64.hs[li - 1] if li else r64.h_enc
        ref = r64.hs[li]
        pg, pm = O.layer_params(sd, f'gnn_layers_gate.{li}.'), O.layer_params(sd, f'gnn_layers.{li}.')
        # float32 numpy evaluation of this pair on the exact input (the "fresh" float32 floor)
        h32 = hin.astype(np.float32)
        sd32 = {k: v.astype(np.float32) for k, v in sd.items()}
        pg32, pm32 = O.layer_params(sd32, f'gnn_layers_gate.{li}.'), O.layer_params(sd32, f'gnn_layers.{li}.')
        a32 = [x.astype(np.float32) for x in (u, pos_x, var)]
        tau = O.sigmoid(O.mp_layer(pg32, h32, *a32, ei, batch, lin=True))
        o32 = (1.0 - tau) * h32 + tau * O.swish(O.mp_layer(pm32, h32, *a32, ei, batch, lin=True))
        print(f'layer pair {li}: float32 numpy on the exact input: rms {np.sqrt(np.mean((o32 - ref) ** 2)):.2e}')
        for name, modes in variants.items():
            tau = O.sigmoid(O.instance_norm(layer_acc(pg, hin, u, pos_x, var, ei, batch, modes), batch))
            out = (1.0 - tau) * hin + tau * O.swish(O.instance_norm(layer_acc(pm, hin, u, pos_x, var, ei, batch, modes), batch))
            print(f'    {name:45s} rms {np.sqrt(np.mean((out - ref) ** 2)):.2e}  max {np.abs(out - ref).max():.2e}')


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[2] == 'acc':
    main_acc(sys.argv[1])


# ---- third experiment (python scripts/diag_quant.py E2 center): update_net_2 / update_net_1 on differences to a reference node ---
def layer_center(p, h, u, pos, var, ei, batch, modes, c4, c3):
    j, i = ei[0], ei[1]
    tw = u.shape[1]
    feat = np.concatenate((u, pos, var), 1)
    fq = feat.copy(); fq[:, tw + 1:] = 0.0
    m = lambda k: modes.get(k, '0')
    P = mfma_gemm(np.concatenate((h, feat), 1), np.concatenate((p.w1[:, :128], p.w1[:, 256:]), 1), p.b1, 256.0, m('w1'))
    Qn = mfma_gemm(np.concatenate((h, -fq), 1), np.concatenate((p.w1[:, 128:256], p.w1[:, 256:]), 1), 0.0 * p.b1, 256.0, m('w1'))
    msg = O.swish(mfma_gemm(O.swish(P[i] + Qn[j]), p.w2, p.b2, 64.0, m('w2')))
    agg = O.scatter_mean(msg, i, h.shape[0])
    x = np.concatenate((h, agg, var), 1)
    first = np.concatenate(([0], np.flatnonzero(batch[1:] != batch[:-1]) + 1))
    ref = first[batch]                                    # first node of each node's graph
    if c3:      # pre_n = pre_ref + W3 (x_n - x_ref): ONE fp32 rounding at full magnitude instead of 50
        pre_ref = mfma_gemm(x[first], p.w3, p.b3, 256.0, m('w3'))[batch]
        d = mfma_gemm((x - x[ref]).astype(np.float32).astype(np.float64), p.w3, 0.0 * p.b3, 256.0, m('w3'))
        pre = (pre_ref + d).astype(np.float32).astype(np.float64)
    else:
        pre = mfma_gemm(x, p.w3, p.b3, 256.0, m('w3'))
    z = O.swish(pre)
    if c4:
        return mfma_gemm((z - z[ref]).astype(np.float32).astype(np.float64), p.w4, 0.0 * p.b4, 64.0, m('w4'))
    return mfma_gemm(z, p.w4, p.b4, 64.0, m('w4'))


def main_center(exp):
    torch.manual_seed(3)
    case = synthetic_case(mp, exp, bsz=8, seed=11, device='cpu')
    kind = 'MP_PDE_SolverLEMLinGated'
    model = getattr(mp, kind)(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=6)
    sd = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    g = case.graph_np()
    r64 = O.solver_forward(kind, sd, g, case.pde, TW, case.eqv, 6, parts=True)
    u = np.asarray(g.x, dtype=np.float64)
    pos_x, pos_t, var = O.build_variables(kind, g, case.pde, case.eqv)
    ei, batch = np.asarray(g.edge_index), np.asarray(g.batch)
    all3 = {k: '3' for k in ('w1', 'w2', 'w3', 'w4')}
    for li in (0, 1, 3):
        hin = r64.hs[li - 1] if li else r64.h_enc
        ref = r64.hs[li]
        pg, pm = O.layer_params(sd, f'gnn_layers_gate.{li}.'), O.layer_params(sd, f'gnn_layers.{li}.')
        print(f'layer pair {li}')
        for name, c4, c3 in (('shipped before round 3', False, False), ('update_net_2 centred (round 3)', True, False),
                             ('update_net_2 and update_net_1 centred', True, True)):
            tau = O.sigmoid(O.instance_norm(layer_center(pg, hin, u, pos_x, var, ei, batch, all3, c4, c3), batch))
            out = (1.0 - tau) * hin + tau * O.swish(O.instance_norm(layer_center(pm, hin, u, pos_x, var, ei, batch, all3, c4, c3), batch))
            print(f'    {name:45s} rms {np.sqrt(np.mean((out - ref) ** 2)):.2e}  max {np.abs(out - ref).max():.2e}')


if __name__ == '__main__' and len(sys.argv) > 2 and sys.argv[2] == 'center':
    main_center(sys.argv[1])
