"""Phase profile of the tile message kernel (library built with MSMP_PROF=tile -> libmsmp_pde_prof.so): cycles of wave 0 per workgroup.
Run on the GPU box:  MSMP_LIB_PATH=$PWD/msmp-pde_amd/libmsmp_pde_prof.so python scripts/prof_tile.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd import _lib
from msmp_pde_amd.graph import structure_of
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
L = mp.lib()
L.msmp_debug_prof_tile.argtypes = [ctypes.c_void_p, ctypes.c_int]
ptr, cs = _lib.ptr, _lib.current_stream
bsz = 2048
case = make_case('E2', bsz, seed=1, device='cuda', dtype=torch.float32)
model = mp.MODEL_NAMES['Gated'](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=1).cuda().eval()
data, labels = case.creator.create_data(case.u_super, [50] * bsz)
graph = case.creator.create_graph(data, labels, case.x, case.variables, [50] * bsz)
gs = structure_of(graph)
n, e = gs.n_nodes, gs.n_edges
layer = model.gnn_layers[0]
packed = layer.packed()
h = torch.randn(n, 128, device='cuda'); u = graph.x.float().contiguous(); pos = torch.rand(n, device='cuda'); var = torch.rand(n, 2, device='cuda')
P = torch.empty(n, 128, device='cuda'); Q = torch.empty(n, 128, device='cuda'); agg = torch.empty(n, 128, device='cuda')
_lib.check(L.msmp_node_project_f32(ptr(h), ptr(u), ptr(pos), ptr(var), n, 25, 2, ptr(packed), ptr(P), ptr(Q), cs()), 'proj')
tiles = gs.tiles(); tb = ctypes.byref(tiles[0])
from msmp_pde_amd.layers import node_features
FEAT = node_features(u, pos, var)
names = ['prologue: index + row loads, staging, barrier', 'projection MFMAs (fold)', 'barrier, P/Q to LDS, barrier (fold)', 'first activation step + gathers',
         'K = 16 step: MFMAs + activation of next (x8)', 'step: weight store + barrier (x7)', 'epilogue: Swish of the messages', 'epilogue: selection matrix, split, mean MFMAs',
         '(unused)', 'epilogue: scale + store']
for fold in (False, True):
    args = (ptr(h), ptr(u), ptr(pos), ptr(var), ptr(FEAT), None, None) if fold else (None, None, None, None, None, ptr(P), ptr(Q))
    run = lambda: _lib.check(L.msmp_edge_aggregate_tiled_f32(*args, ptr(gs.rowptr), tb, n, e, 25, 2, ptr(packed), ptr(agg), cs()), 'tiled')
    run(); torch.cuda.synchronize()
    L.msmp_debug_prof_tile(None, 1)
    for _ in range(5): run()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    L.msmp_debug_prof_tile(buf, 0)
    tot = sum(buf[i] for i in range(10)) or 1
    n_wg = buf[15] or 1        # workgroups that reported (one in 64)
    print(f'edge_tile_kernel<fold={fold}>: {tot / n_wg:.0f} cycles per workgroup (wave 0)')
    for i, nm in enumerate(names):
        print(f'  {nm:48s} {100.0 * buf[i] / tot:5.1f} %   {buf[i] / n_wg:8.0f} cycles')
