"""Phase profile of the anti-phased LEM encoder (library built with MSMP_PROF=lem -> libmsmp_pde_prof.so): cycles wave 0 (role A) and
wave 4 (role B) of one workgroup in 32 spend in matrix halves, vector halves, arriving at barriers and waiting in them.
    MSMP_LIB_PATH=$PWD/msmp-pde_amd/libmsmp_pde_prof.so python scripts/prof_lem.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
L = mp.lib()
L.msmp_debug_prof_lem.argtypes = [ctypes.c_void_p, ctypes.c_int]
EDITION = int(os.environ.get('LEM_EDITION', '4'))      # 4: anti-phased three-tile kernel; 5: one wave per SIMD, two tiles (round 4)
L.msmp_tune(b'lem', EDITION)
bsz = 2048
EXP, MODEL = os.environ.get('LEM_EXP', 'E2'), os.environ.get('LEM_MODEL', 'MSMP-PDE')       # e.g. LEM_EXP=MSWG3 LEM_MODEL=MSMP-PDE2D
case = make_case(EXP, bsz, seed=1, device='cuda', dtype=torch.float32)
model = mp.MODEL_NAMES[MODEL](case.pde, time_window=25, eq_variables=EXPERIMENTS[EXP], hidden_layer=1).cuda().eval()
data, labels = case.creator.create_data(case.u_super, [50] * bsz)
graph = case.creator.create_graph(data, labels, case.x, case.variables, [50] * bsz)
with torch.no_grad():
    model(graph); torch.cuda.synchronize()
    L.msmp_debug_prof_lem(None, 1)
    for _ in range(3): model(graph)
    torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
L.msmp_debug_prof_lem(buf, 0)
if EDITION == 5:
    n = buf[3] or 1
    m, v, b, w = buf[0] / n, buf[1] / n, buf[2] / n, buf[8] / n
    print(f'ws1 (one wave per SIMD): per workgroup (64 nodes, 25 steps = 100 fused phases): fused matrix || vector phases {m:9.0f} cycles ({m / 100:6.0f} per phase)   '
          f'step-input fragments {v:9.0f} ({v / 50:6.0f} per tile and step)   barrier wait {b:9.0f} ({b / 100:5.0f} per phase)   arrive {w:8.0f}   total {m + v + b + w:9.0f}')
for role in ((0, 1) if EDITION != 5 else ()):
    n = buf[4 * role + 3] or 1
    m, v, b, w = buf[4 * role] / n, buf[4 * role + 1] / n, buf[4 * role + 2] / n, buf[8 + role] / n
    print(f'role {"AB"[role]}: per workgroup (96 nodes, 25 steps = 75 items): matrix halves {m:9.0f} cycles ({m / 75:6.0f} per item)   vector halves {v:9.0f} ({v / 75:6.0f} per item)'
          f'   barrier wait {b:9.0f}   arrive {w:8.0f}   total {m + v + b + w:9.0f}')
mp.last_status(reset=True)
