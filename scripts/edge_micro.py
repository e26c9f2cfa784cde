"""What bounds the message kernel?  Times msmp_edge_aggregate_projected_f32 (and msmp_node_project_f32) at the bench size with
  (a) the real E2 structure: P / Q rows gathered from 2 x 105 MB tensors (beyond the per-XCD L2, i.e. HBM / MALL latency),
  (b) the same instruction stream with every gather folded onto the first 64 rows (tgt % 64, col % 64 -> L1 / L2 hits):
      the difference is what the row-gather MISS LATENCY costs (the kernel's arithmetic, LDS and launch shape are unchanged),
  (c) P / Q of a small batch (256 graphs: 2 x 13 MB, L2 / MALL resident) at the real structure.
Run on the GPU box:  python scripts/edge_micro.py [--graphs 2048]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import msmp_pde_amd as mp
from msmp_pde_amd import _lib
from msmp_pde_amd.graph import structure_of
from msmp_pde_amd.synthetic import make_case, EXPERIMENTS

ap = argparse.ArgumentParser()
ap.add_argument('--graphs', type=int, default=2048)
ap.add_argument('--experiment', default='E2')
ap.add_argument('--reps', type=int, default=30)
args = ap.parse_args()
L = mp.lib()
ptr, cs = _lib.ptr, _lib.current_stream


def timed(fn, reps=args.reps):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3     # us


def run(bsz):
    case = make_case(args.experiment, bsz, seed=1, device='cuda', dtype=torch.float32)
    model = mp.MODEL_NAMES['Gated' if args.experiment in ('E2', 'WE3') else 'Gated2D'](case.pde, time_window=25,
                                                                                  eq_variables=EXPERIMENTS[args.experiment], hidden_layer=1).cuda().eval()
    data, labels = case.creator.create_data(case.u_super, [50] * bsz)
    graph = case.creator.create_graph(data, labels, case.x, case.variables, [50] * bsz)
    gs = structure_of(graph)
    n, e = gs.n_nodes, gs.n_edges
    layer = model.gnn_layers[0]
    packed = layer.packed()
    tw, nv = layer.time_window, layer.n_variables
    h = torch.randn(n, 128, device='cuda')
    u = graph.x.float().contiguous()
    pos = torch.rand(n, device='cuda')
    var = torch.rand(n, nv, device='cuda')
    P = torch.empty(n, 128, device='cuda')
    Q = torch.empty(n, 128, device='cuda')
    agg = torch.empty(n, 128, device='cuda')
    proj = lambda: _lib.check(L.msmp_node_project_f32(ptr(h), ptr(u), ptr(pos), ptr(var), n, tw, nv, ptr(packed), ptr(P), ptr(Q), cs()), 'proj')
    proj()
    from msmp_pde_amd.layers import node_features
    global FEAT
    FEAT = node_features(u, pos, var)

    def edge(col, tgt):
        return lambda: _lib.check(L.msmp_edge_aggregate_projected_f32(ptr(P), ptr(Q), ptr(gs.rowptr), ptr(col), ptr(tgt), n, e, gs.max_in_degree,
                                                                       tw, nv, ptr(packed), ptr(agg), cs()), 'edge')
    t_proj = timed(proj)
    t_real = timed(edge(gs.col, gs.tgt))
    tiles = gs.tiles()
    t_staged = t_fold = None
    if tiles is not None:
        import ctypes
        tb = ctypes.byref(tiles[0])
        staged = lambda: _lib.check(L.msmp_edge_aggregate_tiled_f32(None, None, None, None, None, ptr(P), ptr(Q), ptr(gs.rowptr), tb, n, e, tw, nv,
                                                                     ptr(packed), ptr(agg), cs()), 'tiled')
        folded = lambda: _lib.check(L.msmp_edge_aggregate_tiled_f32(ptr(h), ptr(u), ptr(pos), ptr(var), ptr(FEAT), None, None, ptr(gs.rowptr), tb, n, e, tw, nv,
                                                                     ptr(packed), ptr(agg), cs()), 'folded')
        t_staged, t_folded = timed(staged), timed(folded)
    col64, tgt64 = (gs.col % 64).contiguous(), (gs.tgt % 64).contiguous()
    t_fold = timed(edge(col64, tgt64))
    print(f'{args.experiment} x{bsz}: N={n} E={e} max in-degree {gs.max_in_degree}')
    if tiles is not None:
        print(f'  tile kernel (tile_nodes {tiles[0].tile_nodes}): staged P/Q {t_staged:8.1f} us   folded projections {t_folded:8.1f} us   '
              f'(gather kernels: proj + edge = {t_proj + t_real:.1f} us)')
    print(f'  node_proj                      {t_proj:8.1f} us   ({(n * 128 * 4 * 3 + n * (tw + 1 + nv) * 4) / t_proj / 1e3:.0f} GB/s algorithmic)')
    print(f'  edge+mean, real gathers        {t_real:8.1f} us')
    print(f'  edge+mean, gathers folded x64  {t_fold:8.1f} us   -> gather-miss latency share {100 * (1 - t_fold / t_real):.0f} %')
    return t_proj, t_real, t_fold


if __name__ == '__main__':
    with torch.no_grad():
        run(args.graphs)
        if args.graphs > 256:
            run(256)
