"""Phase profile of grad_weight_kernel (library built with MSMP_PROF=gw -> libmsmp_pde_prof.so): cycles of wave 0 of the long workgroups
(>= 100 blocks of 16 rows: the edge-sized jobs) per block.   MSMP_LIB_PATH=$PWD/msmp-pde_amd/libmsmp_pde_prof.so python scripts/prof_gw.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
from msmp_pde_amd.train import training_step
L = mp.lib()
L.msmp_debug_prof_gw.argtypes = [ctypes.c_void_p, ctypes.c_int]
bsz = 512
torch.manual_seed(0)
case = make_case('E2', bsz, seed=1, device='cuda', dtype=torch.float32)
model = mp.MODEL_NAMES['MSMP-PDE'](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=6).cuda().train()
opt = mp.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-8)
steps = [60] * bsz
for _ in range(2): training_step(model, case.creator, case.u_super, case.x, case.variables, steps, 1, opt)
torch.cuda.synchronize(); L.msmp_debug_prof_gw(None, 1)
for _ in range(3): training_step(model, case.creator, case.u_super, case.x, case.variables, steps, 1, opt)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 8)(); L.msmp_debug_prof_gw(buf, 0)
nb = buf[6] or 1
names = ['issue the next block\'s loads', 'wait for this block + publish B + split A', 'barrier', '15 fragment reads + 30 MFMAs + rotate', 'epilogue (once)']
print(f'{buf[7]} workgroups reported, {nb / max(buf[7], 1):.0f} blocks each; cycles per block (wave 0):')
for i, n in enumerate(names): print(f'  {n:44s} {buf[i] / nb:8.0f}')
print(f'  sum {sum(buf[i] for i in range(4)) / nb:8.0f}   (MFMA pipe time of the 30 MFMAs: 960)')
