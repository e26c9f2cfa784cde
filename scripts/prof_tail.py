import sys, ctypes
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
L = mp.lib()
L.msmp_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
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
    n = buf[15] or 1        # workgroups that reported (one in 16)
    names = ['gate head rest (GEMM4)', 'gate norm+sigmoid', 'main head rest', 'main norm', 'blend+store', 'head prologue (x2)', 'chunk: split+rowload (x16)', 'chunk: mma (x16)', 'chunk: weight wait+store (x16)', 'chunk: barrier (x16)', 'swish z (x2)']
    tot = sum(buf[i] for i in range(11))
    print(f'node_tail_split_kernel: {tot / n:.0f} cycles per workgroup (wave 0)')
    for i, nm in enumerate(names): print(f'  {nm:32s} {buf[i] / n:10.0f} cycles  {100.0 * buf[i] / tot:5.1f} %')
