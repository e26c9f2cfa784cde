"""Phase profile of the edge kernel (library built with MSMP_PROF=edge): cycles of wave 0 per workgroup."""
import sys, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
L = mp.lib()
L.msmp_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.msmp_tune(b'tail', 0)     # the tail kernel shares the counters
if '--streamed' in sys.argv: L.msmp_tune(b'edge_ws', 0)
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
    if tot:
        print('streamed-weight edge kernel (msmp_tune edge_ws 0):')
        for i, nm in enumerate(names): print(f'{nm:42s} {100.0 * buf[i] / tot:5.1f} %')

    try:
        L.msmp_debug_prof_ws
        have_ws = True
    except AttributeError:
        have_ws = False
    if have_ws:
        L.msmp_debug_prof_ws.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.msmp_debug_prof_ws(None, 1)
        for _ in range(3): model(graph)
        torch.cuda.synchronize()
        L.msmp_debug_prof_ws(buf, 0)
        names = ['block head (edge range, bias init)', 'chunk: swish + issue loads (x4)', 'chunk: split + 24 MFMAs (x4)', 'S fragments (bpermute)',
                 'mean: swish + split + MFMAs (x4)', 'stores (x4, inside mean loop tail)', 'rotate + index loads']
        tot = sum(buf[i] for i in range(7)) or 1
        n_blk = 3 * 12 * 40960 / 8        # blocks seen by wave 0 of each workgroup
        print('weight-stationary edge kernel, cycles per block of wave 0: %.0f' % (tot / n_blk))
        for i, nm in enumerate(names): print(f'{nm:42s} {100.0 * buf[i] / tot:5.1f} %')
