"""Phase profile of the edge kernel (library built with MSMP_PROF=edge): cycles of wave 0 per workgroup."""
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
    n_wg = 3 * 12 * ((204800 + 20) // 21)      # approx: node tiles of ~21 targets (128 edges)
    names = ['prologue (idx, W chunk 0, first gather)', 'swish (x4)', 'gather issue + split (x4)', 'mma (x4)', 'weights store + barrier (x4)',
             'rowptr fetch', 'round barrier A', 'swish + LDS stage', 'round barrier B', 'segmented mean + store']
    tot = sum(buf[i] for i in range(10))
    for i, nm in enumerate(names): print(f'{nm:42s} {100.0 * buf[i] / tot:5.1f} %')
