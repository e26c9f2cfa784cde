"""Shared helpers for the tests: golden loading and synthetic E2-shaped batches (no reference needed)."""
import os
from types import SimpleNamespace

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def sd_of(d):
    return {k[3:]: v for k, v in d.items() if k.startswith('sd_')}


def graph_of(d, prefix='g_'):
    g = SimpleNamespace()
    for k, v in d.items():
        if k.startswith(prefix):
            setattr(g, k[len(prefix):], v.astype(np.float64) if v.dtype == np.float32 else v)
    return g


def pde_of(d, nt=250, nx=100):
    return SimpleNamespace(tmin=float(d['tmin']), tmax=float(d['tmax']), dt=float(d['dt']), L=float(d['L']),
                           grid_size=[nt, nx])


EXPERIMENTS = {  # experiment -> (pde name, eq_variables, unstructured)
    'E2': ('CE', {'beta': 0.2}, False),
    'WE3': ('WE', {'bc_left': 1, 'bc_right': 1}, False),
    'RPU': ('AD', {'a': 1., 'b': 1.}, True),
    'MSWG3': ('AD', {'a': 1., 'b': 1.}, False),
}


class _Case(object):
    pass


def synthetic_case(mp, exp, bsz, seed, device='cuda', step=50):
    """An `exp`-shaped batch built by the product's GraphCreator mirror from synthetic trajectories.
    On the CPU (host-logic tests) the edge_index comes from the oracle's builder."""
    import torch
    from msmp_pde_amd.synthetic import make_case
    from oracle import msmp_oracle as O
    c = make_case(exp, bsz, seed=seed, device=device, dtype=torch.float64)
    steps = [step] * bsz
    data, labels = c.creator.create_data(c.u_super, steps)
    ei = None
    if device == 'cpu':
        pde_name, _, unstructured = EXPERIMENTS[exp]
        nx = c.pde.grid_size[1]
        x0 = c.x[0].numpy()
        batch = np.repeat(np.arange(bsz), nx)
        if pde_name == 'AD' and unstructured:
            xx = 2 * np.pi * x0 / (x0.max() - 1e-3)
            ei = O.knn_graph(np.tile(np.stack([np.cos(xx), np.sin(xx)], 1), (bsz, 1)), 3, batch)
        elif pde_name == 'WE':
            ei = O.knn_graph(np.tile(x0, bsz), 3, batch)
        else:
            ei = O.radius_graph(np.tile(x0, bsz), 3 * (x0[1] - x0[0]) + 0.0001, batch)
        ei = torch.tensor(ei)
    out = _Case()
    out.pde, out.eqv, out.creator, out.u_super = c.pde, c.eqv, c.creator, c.u_super
    out.graph = c.creator.create_graph(data, labels, c.x, c.variables, steps, edge_index=ei)

    def graph_np():
        g = SimpleNamespace()
        for k, v in out.graph.__dict__.items():
            if torch.is_tensor(v):
                setattr(g, k, v.detach().cpu().numpy())
        return g
    out.graph_np = graph_np
    return out
