"""Phase profile of node_proj_split_kernel (library built with MSMP_PROF=proj): share of wave 0's cycles per phase."""
import sys, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
L = mp.lib()
L.msmp_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.msmp_tune(b'tail', 0)     # the tail kernel shares the counters
case = make_case('E2', 2048, seed=1000, device='cuda', dtype=torch.float32)
model = mp.MODEL_NAMES['MSMP-PDE'](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=6).cuda().eval()
data, labels = case.creator.create_data(case.u_super, [50] * 2048)
graph = case.creator.create_graph(data, labels, case.x, case.variables, [50] * 2048)
with torch.no_grad():
    model(graph); torch.cuda.synchronize()
    L.msmp_debug_prof(None, 1)
    for _ in range(3): model(graph)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 16)()
    L.msmp_debug_prof(buf, 0)
    names = ['prologue (first W chunk + h tile, barrier)', 'chunk: issue loads + B fragment read (x8)', 'chunk: 24 MFMAs (x8)', 'chunk: W / h-tile LDS stores (x8)',
             'chunk: barrier (x8)', 'tail chunk: u / pos / vars loads + split', 'tail chunk: 48 MFMAs + barrier', 'epilogue: P, Q stores']
    tot = sum(buf[i] for i in range(8))
    print('node_proj_split_kernel: %.0f cycles per workgroup (wave 0)' % (tot / (3 * 12 * 1600)))
    for i, nm in enumerate(names): print(f'{nm:46s} {100.0 * buf[i] / tot:5.1f} %')
