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
