"""Standalone row L2 (msmp_scatter_mean_f32, the CSR segmented mean) at the E2-2048 size: achieved HBM GB/s against SURVEY 8(d)'s
algorithmic bytes (read msg E*H*4 + rowptr, write N*H*4 = 722 MB).  The default layer path fuses the mean into the message kernel
and never launches it; it serves graphs with a target of more than 256 in-edges, the layer backward's recompute, and the
out-edge mean of the G2 variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd._lib import lib, check, ptr, current_stream
from msmp_pde_amd.synthetic import make_case
from msmp_pde_amd.graph import structure_of
c = make_case('E2', 2048, seed=1, device='cuda', dtype=torch.float32)
data, labels = c.creator.create_data(c.u_super, [50] * 2048)
g = c.creator.create_graph(data, labels, c.x, c.variables, [50] * 2048)
gs = structure_of(g)
n, e = gs.n_nodes, gs.n_edges
msg = torch.randn(e, 128, device='cuda')
out = torch.empty(n, 128, device='cuda')
L = lib()
for _ in range(3):
    check(L.msmp_scatter_mean_f32(ptr(msg), ptr(gs.rowptr), n, ptr(out), current_stream()), 'scatter')
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
a.record()
for _ in range(reps):
    check(L.msmp_scatter_mean_f32(ptr(msg), ptr(gs.rowptr), n, ptr(out), current_stream()), 'scatter')
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / reps
nbytes = e * 128 * 4 + (n + 1) * 4 + n * 128 * 4
print(f'scatter_mean_kernel: N = {n}, E = {e}: {ms * 1e3:.1f} us per launch, {nbytes / 1e6:.0f} MB algorithmic -> {nbytes / ms / 1e6:.0f} GB/s '
      f'= {nbytes / ms / 1e6 / 8000:.2f} of the 8 TB/s HBM3E peak')
