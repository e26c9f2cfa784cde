"""msmp_linear_f32 (bf16x3 row GEMM, fp32-exact) against torch.addmm (hipBLASLt fp32) at the shapes of the width-164 LEM cell."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd._lib import lib, check, ptr, current_stream
L = lib()
n = 204800
for k, n_out in ((168, 492), (168, 164), (164, 492), (164, 164)):
    x = torch.randn(n, 168, device='cuda'); w = torch.randn(n_out, 168, device='cuda') * 0.1; b = torch.randn(n_out, device='cuda')
    ld = 128 * ((n_out + 127) // 128)
    out = torch.empty(n, ld, device='cuda')
    ws = torch.empty(L.msmp_linear_workspace_bytes(k, n_out) // 4 + 64, device='cuda')
    f = lambda: check(L.msmp_linear_f32(ptr(x), 168, n, k, ptr(w), 168, ptr(b), n_out, 0, ptr(out), ld, ptr(ws), ws.numel() * 4, current_stream()), 'lin')
    wt = w[:, :k].t().contiguous(); xk = x[:, :k].contiguous()
    g = lambda: torch.addmm(b, xk, wt)
    res = []
    for fn in (f, g):
        for _ in range(3): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20): fn()
        torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 20 * 1e6)
    ref = torch.addmm(b.double(), xk.double(), wt.double())
    e1 = (out[:, :n_out].double() - ref).abs().max().item(); e2 = (g().double() - ref).abs().max().item()
    print(f'k {k} n_out {n_out}: msmp_linear {res[0]:7.1f} us ({2 * n * k * n_out / res[0] / 1e6:6.1f} TFLOP/s, max err {e1:.1e})   addmm {res[1]:7.1f} us ({2 * n * k * n_out / res[1] / 1e6:6.1f} TFLOP/s, max err {e2:.1e})')
