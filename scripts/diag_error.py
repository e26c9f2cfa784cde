"""Diagnostic (not a test): where does the fp32 error of the HIP path come from?  Per-layer hidden
state error vs the float64 oracle for (a) the HIP path, (b) the oracle in float32, (c) a plain torch
fp32 GPU evaluation of the same math."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import msmp_pde_amd as mp
from msmp_pde_amd.graph import structure_of
from helpers import synthetic_case
from oracle import msmp_oracle as O

kind, exp = sys.argv[1] if len(sys.argv) > 1 else 'MP_PDE_SolverGated', sys.argv[2] if len(sys.argv) > 2 else 'E2'
torch.manual_seed(3)
case = synthetic_case(mp, exp, bsz=8, seed=11)
model = getattr(mp, kind)(case.pde, time_window=25, eq_variables=case.eqv, hidden_layer=6).cuda().eval()
data = case.graph.to('cuda')
sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
g = case.graph_np()
r64 = O.solver_forward(kind, sd, g, case.pde, 25, case.eqv, 6, parts=True)
r32 = O.solver_forward(kind, sd, g, case.pde, 25, case.eqv, 6, dtype=np.float32, parts=True)

with torch.no_grad():
    u = data.x.float()
    pos_x = (data.pos[:, 1][:, None] / case.pde.L).float()
    pos_t = (data.pos[:, 0][:, None] / case.pde.tmax)
    var = model._variables(data, pos_t).float()
    pos_t = pos_t.float()
    gs = structure_of(data)
    dt = torch.cumsum(torch.ones(25, device='cuda') * case.pde.dt, 0)
    h = model._encode(u, pos_x, pos_t, var, dt)
    print('encoder err hip/torch:', np.abs(h.double().cpu().numpy() - r64.h_enc).max(), ' f32 oracle:', np.abs(r32.h_enc - r64.h_enc).max())
    # (a) HIP path fed with its own h; (d) HIP layer fed with the exact (f64-rounded) previous h: fresh error only
    for i in range(6):
        gate = model.gnn_layers_gate[i] if model.GATED else None
        h = mp.mp_layer(h, u, pos_x, var, gs, model.gnn_layers[i], gate)
        hin = torch.tensor(r64.hs[i - 1] if i else r64.h_enc).float().cuda()
        hfresh = mp.mp_layer(hin, u, pos_x, var, gs, model.gnn_layers[i], gate)
        e_hip = np.abs(h.double().cpu().numpy() - r64.hs[i]).max()
        e_fresh = np.abs(hfresh.double().cpu().numpy() - r64.hs[i]).max()
        e_f32 = np.abs(r32.hs[i] - r64.hs[i]).max()
        print(f'layer {i}: hip {e_hip:.3e}   hip-fresh(exact input) {e_fresh:.3e}   f32-oracle {e_f32:.3e}')
