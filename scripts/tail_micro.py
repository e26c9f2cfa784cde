"""Times msmp_node_tail_f32 (gated) at the bench size with HIP events; used with the ablation builds (MSMP_LIB_PATH)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd import _lib
from msmp_pde_amd.graph import structure_of
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
L = mp.lib(); ptr, cs = _lib.ptr, _lib.current_stream
bsz = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
case = make_case('E2', bsz, seed=1, device='cuda', dtype=torch.float32)
model = mp.MODEL_NAMES['Gated'](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=1).cuda().eval()
data, labels = case.creator.create_data(case.u_super, [50] * bsz)
graph = case.creator.create_graph(data, labels, case.x, case.variables, [50] * bsz)
gs = structure_of(graph)
n = gs.n_nodes
h = torch.randn(n, 128, device='cuda'); am = torch.randn(n, 128, device='cuda'); ag = torch.randn(n, 128, device='cuda')
var = torch.rand(n, 2, device='cuda'); out = torch.empty(n, 128, device='cuda')
pm, pg = model.gnn_layers[0].packed(), model.gnn_layers_gate[0].packed()
run = lambda: _lib.check(L.msmp_node_tail_f32(ptr(h), ptr(am), ptr(ag), ptr(var), ptr(gs.graph_ptr), n, gs.n_graphs, gs.max_graph_nodes, 2,
                                              ptr(pm), ptr(pg), 1, 1e-5, ptr(out), cs()), 'tail')
for _ in range(5): run()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(40): run()
e1.record(); torch.cuda.synchronize()
print(f'{os.path.basename(os.environ.get("MSMP_LIB_PATH", "libmsmp_pde.so"))}: node tail (gated, {bsz} graphs) {e0.elapsed_time(e1) / 40 * 1e3:.1f} us')
mp.last_status(reset=True)
