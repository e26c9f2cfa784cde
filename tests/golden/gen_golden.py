#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own code.

Runs only in the build container (needs /root/reference, which never travels to the GPU box):

    python tests/golden/gen_golden.py

It imports /root/reference/experiments/models_gnn.py, models_gnn2D.py and common/utils.py
unchanged, with `tests/golden/standins/` satisfying the third-party module names this image lacks
(torch_geometric, torch_cluster, torch_scatter, lem_cuda, h5py; see standins/README.md), runs the
reference classes in the reference's effective dtype (float64, temporal/solvers.py:10) and stores
inputs + outputs as small .npz files.  All inputs and weights are first rounded to
float32-representable values, so the float64 oracle and the float32 HIP path see identical numbers
and any difference is arithmetic error only.

Fixtures written:
  graph_{E2,WE3,RPU,MSWG3}.npz   GraphCreator.create_data/create_graph/create_next_graph tensors
                                 (edge_index, x, y, pos, batch, parameter columns)
  layer_{GNN_Layer,GNN_LayerLin}.npz   one message-passing layer: inputs, weights, pre-/post-norm out
  solver_{MP_PDE_Solver,MP_PDE_SolverGated,MP_PDE_Solver2D,MP_PDE_Solver2DGated}.npz
                                 state_dict + graph batch + forward output (+ 3-step rollout for E2)
  deep_{class}_{experiment}.npz  FULL-DEPTH (hidden_layer = 6, the reference's default) forward + one rollout step of the
                                 LEM-free classes; the 0.6-1.3 M parameters are not stored but re-created from
                                 seeded_weights.seeded_state_dict (same function in the tests), so a fixture is ~100 KB

    python tests/golden/gen_golden.py            # everything
    python tests/golden/gen_golden.py deep       # only the full-depth fixtures
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = '/root/reference'
if not os.path.isdir(REF):
    sys.exit('reference tree not present: golden vectors can only be regenerated in the build container')
sys.path.insert(0, os.path.join(HERE, 'standins'))
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import experiments.models_gnn as M  # noqa: E402
import experiments.models_gnn2D as M2  # noqa: E402
import common.utils as U  # noqa: E402
from equations.PDEs import CE, WE, AD  # noqa: E402

assert torch.get_default_dtype() == torch.float64  # side effect of temporal/solvers.py:10

NT, NX, TW = 250, 100, 25


def f32r(a):
    """Round to float32-representable float64."""
    return np.asarray(a, dtype=np.float64).astype(np.float32).astype(np.float64)


def cheb_grid(xmin, xmax, n):
    """Our restatement of the Chebyshev grid the reference generator uses (generate/generate_data.py:64-78)."""
    x = np.cos(np.arange(0, n) * np.pi / (n - 1))[::-1]
    return (xmax - xmin) * ((x + 1.) / 2.) + xmin


def pseudo_random_grid(xmin, xmax, n):
    """Our restatement of generate/generate_data.py:80-113 (LCG a=75, c=74, p=65537)."""
    a, c, p = 75, 74, 2 ** 16 + 1
    ns = [(a * 0 + c) % p]
    for _ in range(n - 1):
        ns.append((a * ns[-1] + c) % p)
    ns = np.array(ns, dtype=np.float64)
    ns = ns / ns.max() * (xmax - xmin) + xmin
    ns = np.sort(ns)
    ns[0], ns[-1] = xmin, xmax
    return ns


def sine_traj(rng, b, nt, x, tmax, length):
    """Smooth O(1) trajectories shaped like the reference's initial conditions (sum of 5 sines,
    generate/generate_data.py:131-151), translated in time.  Not a PDE solution; only shape/scale matter."""
    t = np.linspace(0, tmax, nt)
    out = np.zeros((b, nt, len(x)))
    for i in range(b):
        A = rng.uniform(-0.5, 0.5, 5)
        om = 0.8 * rng.uniform(-0.5, 0.5, 5)
        phi = rng.uniform(0, 2 * np.pi, 5)
        l = rng.integers(1, 3, 5)
        for k in range(5):
            out[i] += A[k] * np.sin(om[k] * t[:, None] + 2 * np.pi * l[k] * x[None, :] / length + phi[k])
    return out


def make_case(name, rng, bsz):
    """(pde, creator, u_super, x, variables, eq_variables) for the four in-scope experiment shapes."""
    if name == 'E2':
        pde = CE(device='cpu')
        pde.tmin, pde.tmax, pde.grid_size = 0.0, 4.0, [NT, NX]
        pde.dt = pde.tmax / (NT - 1)
        x = np.linspace(0, 16, NX)
        u = sine_traj(rng, bsz, NT, x, pde.tmax, 16.0)
        variables = {'alpha': np.ones(bsz), 'beta': f32r(rng.uniform(0, 0.2, bsz)), 'gamma': np.zeros(bsz)}
        eqv = {'beta': 0.2}
    elif name == 'WE3':
        pde = WE(device='cpu')
        pde.tmin, pde.tmax, pde.grid_size = 0.0, 100.0, [NT, NX]
        pde.dt = pde.tmax / (NT - 1)
        x = cheb_grid(-8, 8, NX)
        t = np.linspace(0, 1, NT)
        s = rng.uniform(-4, 4, bsz)
        u = np.exp(-(x[None, None, :] - s[:, None, None] - 2 * t[None, :, None]) ** 2)
        variables = {'bc_left': rng.integers(0, 2, bsz).astype(np.float64),
                     'bc_right': rng.integers(0, 2, bsz).astype(np.float64), 'c': np.ones(bsz) * 2.0}
        eqv = {'bc_left': 1, 'bc_right': 1}
    elif name in ('RPU', 'MSWG3'):
        pde = AD(device='cpu')
        pde.tmin, pde.tmax, pde.grid_size = 0.0, 1.0, [NT, NX]
        pde.dt = pde.tmax / (NT - 1)
        if name == 'RPU':
            pde.untructured_grid = True
            x = pseudo_random_grid(0, 16, NX)
        else:
            x = np.linspace(0, 2 * np.pi, NX)
        u1 = sine_traj(rng, bsz, NT, x, pde.tmax, x[-1])
        u2 = sine_traj(rng, bsz, NT, x, pde.tmax, x[-1])
        u = np.stack([u1, u2], axis=2)          # [B, nt, 2, nx] as HDF5Dataset returns (common/utils.py:260-261)
        variables = {'a': f32r(rng.uniform(.1, 1., bsz)), 'b': f32r(rng.uniform(1., 10., bsz))}
        eqv = {'a': 1., 'b': 1.}
    else:
        raise ValueError(name)
    creator = U.GraphCreator(pde=pde, neighbors=3, time_window=TW, t_resolution=NT, x_resolution=NX)
    x = f32r(x)
    return pde, creator, f32r(u), x, variables, eqv


F32_KEYS = ('u_super', 'pred', 'g_x', 'g_y', 'n_x', 'n_y', 'h', 'u', 'variables')


def save(path, d):
    """Arrays whose values are float32-representable by construction are stored as float32 (half the bytes);
    reference OUTPUTS stay float64."""
    out = {}
    for k, v in d.items():
        v = np.asarray(v)
        if k in F32_KEYS:
            assert np.array_equal(v.astype(np.float32).astype(np.float64), v), k
            v = v.astype(np.float32)
        out[k] = v
    np.savez_compressed(path, **out)


def graph_to_np(g, prefix=''):
    d = {}
    for k in ('x', 'y', 'pos', 'batch', 'edge_index', 'alpha', 'beta', 'gamma', 'bc_left', 'bc_right', 'c', 'a', 'b'):
        if hasattr(g, k) and getattr(g, k) is not None:
            d[prefix + k] = getattr(g, k).detach().numpy().copy()
    return d


def build_graph(creator, u, x, variables, steps):
    ut = torch.tensor(u)
    xt = torch.tensor(np.tile(x[None], (len(u), 1)))
    vt = {k: torch.tensor(v) for k, v in variables.items()}
    data, labels = creator.create_data(ut, steps)
    return creator.create_graph(data, labels, xt, vt, steps), ut


def round_params(model):
    with torch.no_grad():
        for p in model.parameters():
            p.copy_(p.float().double())


def gen_graphs(rng):
    for name in ('E2', 'WE3', 'RPU', 'MSWG3'):
        bsz = 3
        pde, creator, u, x, variables, eqv = make_case(name, rng, bsz)
        steps = [50, 75, 100]
        g, ut = build_graph(creator, u, x, variables, steps)
        out = {'u_super': u[:, :175], 'x_grid': x, 'steps': np.array(steps),
               'tmin': pde.tmin, 'tmax': pde.tmax, 'dt': pde.dt, 'L': float(pde.L)}
        out.update({'var_' + k: v for k, v in variables.items()})
        out.update(graph_to_np(g, 'g_'))
        # one create_next_graph step with a deterministic fake prediction
        pred = torch.tensor(f32r(rng.standard_normal(tuple(g.x.shape))))
        steps2 = [s + TW for s in steps]
        _, labels2 = creator.create_data(ut, steps2)
        g2 = creator.create_next_graph(g, pred, labels2, steps2)
        out['pred'] = pred.numpy()
        out['steps2'] = np.array(steps2)
        out.update(graph_to_np(g2, 'n_'))
        save(os.path.join(HERE, f'graph_{name}.npz'), out)
        print(name, 'edges', g.edge_index.shape[1], 'nodes', g.x.shape)


def gen_layers(rng):
    torch.manual_seed(1)
    pde, creator, u, x, variables, eqv = make_case('E2', rng, 2)
    g, _ = build_graph(creator, u, x, variables, [50, 60])
    n = g.x.shape[0]
    v = 2
    for cls in (M.GNN_Layer, M.GNN_LayerLin):
        layer = cls(128, 128, 128, TW, v)
        round_params(layer)
        h = torch.tensor(f32r(rng.standard_normal((n, 128))))
        pos_x = g.pos[:, 1][:, None] / 16.0
        pos_x = pos_x.float().double()
        var = torch.tensor(f32r(rng.uniform(0, 1, (n, v))))
        uu = g.x.float().double()
        with torch.no_grad():
            pre = layer.propagate(g.edge_index, x=h, u=uu, pos=pos_x, variables=var)
            out = layer(h, uu, pos_x, var, g.edge_index, g.batch)
        d = {'h': h.numpy(), 'u': uu.numpy(), 'pos_x': pos_x.numpy(), 'variables': var.numpy(),
             'edge_index': g.edge_index.numpy(), 'batch': g.batch.numpy(), 'pre': pre.numpy(), 'out': out.numpy()}
        for k, p in layer.state_dict().items():
            d['sd_' + k] = p.numpy().astype(np.float32)
        save(os.path.join(HERE, f'layer_{cls.__name__}.npz'), d)
        print(cls.__name__, float(out.abs().max()))


def gen_solvers(rng):
    cases = [('MP_PDE_Solver', M.MP_PDE_Solver, 'E2', 2, 3),
             ('MP_PDE_SolverGated', M.MP_PDE_SolverGated, 'E2', 2, 0),
             ('MP_PDE_Solver2D', M2.MP_PDE_Solver2D, 'MSWG3', 1, 0),
             ('MP_PDE_Solver2DGated', M2.MP_PDE_Solver2DGated, 'RPU', 1, 2)]
    for seed, (name, cls, exp, layers, n_roll) in enumerate(cases):
        torch.manual_seed(10 + seed)
        bsz = 4
        pde, creator, u, x, variables, eqv = make_case(exp, rng, bsz)
        model = cls(pde, time_window=TW, eq_variables=eqv, hidden_layer=layers)
        round_params(model)
        model.eval()
        assert repr(model) == 'GNN'
        steps = [50] * bsz
        g, ut = build_graph(creator, u, x, variables, steps)
        d = {'experiment': exp, 'hidden_layer': layers, 'u_super': u[:, :50 + TW * (n_roll + 2)], 'x_grid': x,
             'tmin': pde.tmin, 'tmax': pde.tmax, 'dt': pde.dt, 'L': float(pde.L), 'steps': np.array(steps)}
        d.update({'var_' + k: v for k, v in variables.items()})
        d.update(graph_to_np(g, 'g_'))
        with torch.no_grad():
            pred = model(g)
            d['out'] = pred.numpy().copy()
            step = 50
            for r in range(n_roll):     # experiments/train_helper.py:255-261
                step += TW
                same = [step] * bsz
                _, labels = creator.create_data(ut, same)
                g = creator.create_next_graph(g, pred, labels, same)
                pred = model(g)
                d[f'roll{r}'] = pred.numpy().copy()
        d['n_roll'] = n_roll
        for k, p in model.state_dict().items():
            d['sd_' + k] = p.numpy().astype(np.float32)
        save(os.path.join(HERE, f'solver_{name}.npz'), d)
        print(name, exp, 'params', sum(p.numel() for p in model.parameters()), 'out max', float(pred.abs().max()))


DEEP_CASES = [('MP_PDE_Solver', 'E2', 1), ('MP_PDE_SolverGated', 'E2', 1), ('MP_PDE_SolverGated', 'WE3', 1),
              ('MP_PDE_Solver2DGated', 'MSWG3', 0), ('MP_PDE_Solver2DGated', 'RPU', 0)]


def gen_deep():
    """Full-depth fixtures: the reference's classes at hidden_layer = 6 with seeded parameters."""
    sys.path.insert(0, HERE)
    from seeded_weights import seeded_state_dict
    rng = np.random.default_rng(20261005)
    for idx, (name, exp, n_roll) in enumerate(DEEP_CASES):
        cls = getattr(M2 if '2D' in name else M, name)
        bsz = 4
        pde, creator, u, x, variables, eqv = make_case(exp, rng, bsz)
        model = cls(pde, time_window=TW, eq_variables=eqv, hidden_layer=6)
        seed = 7000 + idx
        shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
        sd = seeded_state_dict(shapes, seed)
        model.load_state_dict({k: torch.tensor(v.astype(np.float64)) for k, v in sd.items()}, strict=True)
        model.eval()
        steps = [50] * bsz
        g, ut = build_graph(creator, u, x, variables, steps)
        d = {'experiment': exp, 'hidden_layer': 6, 'weights_seed': seed, 'u_super': u[:, :50 + TW * (n_roll + 2)], 'x_grid': x,
             'tmin': pde.tmin, 'tmax': pde.tmax, 'dt': pde.dt, 'L': float(pde.L), 'steps': np.array(steps),
             'param_names': np.array(list(shapes.keys())), 'param_shapes': np.array([','.join(str(n) for n in s) for s in shapes.values()]),
             'param_checksum': np.float64(sum(float(np.abs(v.astype(np.float64)).sum()) for v in sd.values()))}
        d.update({'var_' + k: v for k, v in variables.items()})
        d.update(graph_to_np(g, 'g_'))
        with torch.no_grad():
            pred = model(g)
            d['out'] = pred.numpy().copy()
            step = 50
            for r in range(n_roll):
                step += TW
                same = [step] * bsz
                _, labels = creator.create_data(ut, same)
                g = creator.create_next_graph(g, pred, labels, same)
                pred = model(g)
                d[f'roll{r}'] = pred.numpy().copy()
        d['n_roll'] = n_roll
        save(os.path.join(HERE, f'deep_{name}_{exp}.npz'), d)
        print('deep', name, exp, 'params', sum(p.numel() for p in model.parameters()), 'out max', float(pred.abs().max()))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'deep':
        gen_deep()
    else:
        rng = np.random.default_rng(20261004)
        gen_graphs(rng)
        gen_layers(rng)
        gen_solvers(rng)
        gen_deep()
