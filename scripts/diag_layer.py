"""Diagnostic (not a test): error of each piece of ONE gated layer pair with the exact (float64-oracle) input h, HIP vs the
float64 oracle, beside the same pieces of the oracle run in float32 (numpy): aggregate (message + mean), pre-norm update,
InstanceNorm, blend.  Shows which stage of the HIP layer is less accurate than a float32 evaluation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import msmp_pde_amd as mp
from msmp_pde_amd import _lib
from msmp_pde_amd.graph import structure_of
from helpers import synthetic_case
from oracle import msmp_oracle as O

kind, exp = 'MP_PDE_SolverGated', sys.argv[1] if len(sys.argv) > 1 else 'E2'
torch.manual_seed(3)
case = synthetic_case(mp, exp, bsz=8, seed=11)
model = getattr(mp, kind)(case.pde, time_window=25, eq_variables=case.eqv, hidden_layer=6).cuda().eval()
data = case.graph.to('cuda')
sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
g = case.graph_np()
r64 = O.solver_forward(kind, sd, g, case.pde, 25, case.eqv, 6, parts=True)
L = mp.lib(); ptr, cs = _lib.ptr, _lib.current_stream
mx = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max())
with torch.no_grad():
    u = data.x.float().contiguous()
    pos_x = (data.pos[:, 1] / case.pde.L).float().contiguous()
    pos_t = (data.pos[:, 0][:, None] / case.pde.tmax)
    var = model._variables(data, pos_t).float().contiguous()
    gs = structure_of(data)
    n, e = gs.n_nodes, gs.n_edges
    nv = var.shape[1]
    for li in (0, 3):
        h64 = r64.h_enc if li == 0 else r64.hs[li - 1]
        hin = torch.tensor(h64).float().cuda().contiguous()
        h32 = hin.cpu().numpy()           # float32-rounded input, what every float32 evaluation starts from
        out = {}
        for dt in (np.float64, np.float32):
            f = lambda a: np.asarray(a, dtype=dt)
            pg = O.layer_params({k: f(v) for k, v in sd.items()}, f'gnn_layers_gate.{li}.')
            pm = O.layer_params({k: f(v) for k, v in sd.items()}, f'gnn_layers.{li}.')
            ei, batch = np.asarray(g.edge_index), np.asarray(g.batch)
            a = (f(h32), f(u.cpu().numpy()), f(pos_x.cpu().numpy())[:, None], f(var.cpu().numpy()), ei, batch)
            rg = O.mp_layer(pg, *a, lin=True, parts=True)
            rm = O.mp_layer(pm, *a, lin=True, parts=True)
            tau = O.sigmoid(rg.out)
            out[dt] = dict(agg_g=rg.agg, agg_m=rm.agg, pre_g=rg.pre, pre_m=rm.pre, in_g=rg.out, in_m=rm.out,
                           blend=(1.0 - tau) * f(h32) + tau * O.swish(rm.out))
        ref, f32 = out[np.float64], out[np.float32]
        # HIP pieces
        res = {}
        for name, layer in (('g', model.gnn_layers_gate[li]), ('m', model.gnn_layers[li])):
            packed = layer.packed()
            P = torch.empty(n, 128, device='cuda'); Q = torch.empty(n, 128, device='cuda'); agg = torch.empty(n, 128, device='cuda'); pre = torch.empty(n, 128, device='cuda')
            _lib.check(L.msmp_node_project_f32(ptr(hin), ptr(u), ptr(pos_x), ptr(var), n, 25, nv, ptr(packed), ptr(P), ptr(Q), cs()), 'proj')
            _lib.check(L.msmp_edge_aggregate_projected_f32(ptr(P), ptr(Q), ptr(gs.rowptr), ptr(gs.col), ptr(gs.tgt), n, e, gs.max_in_degree, 25, nv, ptr(packed), ptr(agg), cs()), 'edge')
            res['agg_' + name] = agg.cpu().numpy()
            # update on the EXACT aggregate (isolates the node kernel)
            agg_x = torch.tensor(ref['agg_' + name]).float().cuda().contiguous()
            _lib.check(L.msmp_node_update_f32(ptr(hin), ptr(agg_x), ptr(var), n, nv, ptr(packed), 1, ptr(pre), cs()), 'upd')
            res['pre_' + name] = pre.cpu().numpy()
            nrm = torch.empty(n, 128, device='cuda')
            pre_x = torch.tensor(ref['pre_' + name]).float().cuda().contiguous()
            _lib.check(L.msmp_instance_norm_f32(ptr(pre_x), ptr(gs.graph_ptr), gs.n_graphs, gs.max_graph_nodes, 1e-5, ptr(nrm), cs()), 'norm')
            res['in_' + name] = nrm.cpu().numpy()
        full = mp.mp_layer(hin, u, pos_x, var, gs, model.gnn_layers[li], model.gnn_layers_gate[li])
        res['blend'] = full.cpu().numpy()
        print(f'--- {exp} layer pair {li} (exact input; every stage fed with the exact output of the stage before, except "blend" = whole HIP layer)')
        for k in ('agg_g', 'agg_m', 'pre_g', 'pre_m', 'in_g', 'in_m', 'blend'):
            scale = np.abs(ref[k]).max()
            print(f'  {k:6s} max|ref| {scale:8.3g}   HIP err {mx(res[k], ref[k]):.3e}   numpy-f32 err (whole chain in f32) {mx(f32[k], ref[k]):.3e}')
        # InstanceNorm conditioning: smallest per-graph std of the pre-norm tensors
        b = int(batch.max()) + 1
        std = np.stack([ref['pre_m'][batch == i].std(0) for i in range(b)])
        print(f'  smallest per-graph channel std of pre_m: {std.min():.3e} (median {np.median(std):.3e})')
