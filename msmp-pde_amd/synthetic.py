"""Synthetic inputs of the reference's shapes and value ranges (there are no datasets in the image
and h5py is absent): grids, smooth O(1) trajectories, equation parameters.  Used by bench.py, smoke
and the tests.  Shapes/distributions follow SURVEY.md section 8d; grids restate
generate/generate_data.py:64-78 (Chebyshev) and :80-113 (pseudo-random LCG grid)."""
import math
from types import SimpleNamespace

import numpy as np
import torch

from .graph import GraphCreator
from .pde import CE, WE, AD

EXPERIMENTS = {   # experiment -> eq_variables (experiments/train.py:374-395)
    'E2': {'beta': 0.2},
    'WE3': {'bc_left': 1, 'bc_right': 1},
    'RPU': {'a': 1., 'b': 1.},
    'MSWG3': {'a': 1., 'b': 1.},
}


def cheb_grid(xmin, xmax, n):
    x = np.cos(np.arange(0, n) * np.pi / (n - 1))[::-1]
    return (xmax - xmin) * ((x + 1.) / 2.) + xmin


def pseudo_random_grid(xmin, xmax, n):
    a, c, p = 75, 74, 2 ** 16 + 1
    ns = [c % p]
    for _ in range(n - 1):
        ns.append((a * ns[-1] + c) % p)
    ns = np.array(ns, dtype=np.float64)
    ns = np.sort(ns / ns.max() * (xmax - xmin) + xmin)
    ns[0], ns[-1] = xmin, xmax
    return ns


def make_pde(exp, nt=250, nx=100):
    if exp == 'E2':
        return CE(tmin=0.0, tmax=4.0, grid_size=[nt, nx], L=16.0)
    if exp == 'WE3':
        return WE(tmin=0.0, tmax=100.0, grid_size=[nt, nx])
    if exp in ('RPU', 'MSWG3'):
        return AD(tmin=0.0, tmax=1.0, grid_size=[nt, nx], L=16.0, unstructured=(exp == 'RPU'))
    raise ValueError(exp)


def make_grid(exp, nx=100):
    if exp == 'E2':
        return np.linspace(0, 16, nx)
    if exp == 'WE3':
        return cheb_grid(-8, 8, nx)
    if exp == 'RPU':
        return pseudo_random_grid(0, 16, nx)
    if exp == 'MSWG3':
        return np.linspace(0, 2 * np.pi, nx)
    raise ValueError(exp)


def _sines(gen, bsz, t, x, length, device):
    """Sum of 5 sines, A in U(-.5,.5), omega in .8 U(-.5,.5), phi in U(0,2pi), l in {1,2}: the family the
    reference draws its initial conditions from (generate/generate_data.py:131-151), moved in time."""
    r = lambda *s: torch.rand(*s, generator=gen, dtype=torch.float64)
    A = (r(bsz, 5) - 0.5)
    om = 0.8 * (r(bsz, 5) - 0.5)
    phi = 2 * math.pi * r(bsz, 5)
    l = torch.randint(1, 3, (bsz, 5), generator=gen).to(torch.float64)
    A, om, phi, l = (v.to(device) for v in (A, om, phi, l))
    t = t.to(device)[None, :, None, None]
    x = x.to(device)[None, None, :, None]
    arg = om[:, None, None, :] * t + 2 * math.pi * l[:, None, None, :] * x / length + phi[:, None, None, :]
    return (A[:, None, None, :] * torch.sin(arg)).sum(-1)      # [B, nt, nx]


def make_case(exp, bsz, seed=0, device='cuda', nt=250, nx=100, tw=25, neighbors=3, dtype=torch.float32):
    """pde, GraphCreator, trajectories u_super ([B,nt,nx] or [B,nt,2,nx], `dtype`, values rounded to
    float32), grid x [B,nx] float64, equation parameters and eq_variables for one experiment."""
    gen = torch.Generator().manual_seed(seed)
    pde = make_pde(exp, nt, nx)
    xg = torch.tensor(make_grid(exp, nx)).float().double()
    t = torch.linspace(pde.tmin, pde.tmax, nt, dtype=torch.float64)
    if exp == 'E2':
        u = _sines(gen, bsz, t, xg, 16.0, device)
        variables = {'alpha': torch.ones(bsz, dtype=torch.float64),
                     'beta': (0.2 * torch.rand(bsz, generator=gen, dtype=torch.float64)).float().double(),
                     'gamma': torch.zeros(bsz, dtype=torch.float64)}
    elif exp == 'WE3':
        s = (8 * torch.rand(bsz, generator=gen, dtype=torch.float64) - 4).to(device)
        tt = torch.linspace(0, 1, nt, dtype=torch.float64).to(device)
        u = torch.exp(-(xg.to(device)[None, None, :] - s[:, None, None] - 2 * tt[None, :, None]) ** 2)
        variables = {'bc_left': torch.randint(0, 2, (bsz,), generator=gen).double(),
                     'bc_right': torch.randint(0, 2, (bsz,), generator=gen).double(),
                     'c': 2.0 * torch.ones(bsz, dtype=torch.float64)}
    else:
        u = torch.stack([_sines(gen, bsz, t, xg, float(xg[-1]), device),
                         _sines(gen, bsz, t, xg, float(xg[-1]), device)], 2)       # [B, nt, 2, nx]
        variables = {'a': (0.1 + 0.9 * torch.rand(bsz, generator=gen, dtype=torch.float64)).float().double(),
                     'b': (1 + 9 * torch.rand(bsz, generator=gen, dtype=torch.float64)).float().double()}
    u = u.float().to(dtype)
    creator = GraphCreator(pde, neighbors=neighbors, time_window=tw, t_resolution=nt, x_resolution=nx, device=device)
    return SimpleNamespace(exp=exp, pde=pde, creator=creator, u_super=u, x=xg[None].repeat(bsz, 1),
                           variables=variables, eqv=dict(EXPERIMENTS[exp]), tw=tw, bsz=bsz)
