"""Socket power and shader clock while ONE kernel runs back to back for a few seconds (round 4: are the hot kernels at the power cap?).
    python scripts/power_kernel.py tile|lem|tail|mfma [seconds]
tile: edge_tile_kernel<2> (E2, 2048 graphs);  lem: the LEM encoder;  tail: node_tail_split_kernel;  mfma: scripts/micro/mfma_acc_file.bin
(the bare MFMA loop with constant operands).  rocm-smi is polled twice a second from a side thread; the kernel's own launch time comes
from HIP events around the whole loop."""
import ctypes, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
which = sys.argv[1] if len(sys.argv) > 1 else 'tile'
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
samples, stop = [], threading.Event()


def poll():
    while not stop.is_set():
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True).stdout
        p = re.search(r'Power \(W\): ([0-9.]+)', out)
        s = re.search(r'sclk clock level: \S+ \((\d+)Mhz\)', out)
        if p and s:
            samples.append((float(p.group(1)), int(s.group(1))))
        time.sleep(0.4)


if which == 'mfma':
    th = threading.Thread(target=poll); th.start()
    t0 = time.time()
    while time.time() - t0 < secs:
        subprocess.run([os.path.join(os.path.dirname(os.path.abspath(__file__)), 'micro', 'mfma_acc_file.bin')], capture_output=True)
    stop.set(); th.join()
else:
    import torch
    import msmp_pde_amd as mp
    from msmp_pde_amd import _lib
    from msmp_pde_amd.graph import structure_of
    from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
    from msmp_pde_amd.layers import node_features
    L = mp.lib(); ptr, cs = _lib.ptr, _lib.current_stream
    bsz = 2048
    case = make_case('E2', bsz, seed=1, device='cuda', dtype=torch.float32)
    model = mp.MODEL_NAMES['MSMP-PDE'](case.pde, time_window=25, eq_variables=EXPERIMENTS['E2'], hidden_layer=1).cuda().eval()
    data, labels = case.creator.create_data(case.u_super, [50] * bsz)
    graph = case.creator.create_graph(data, labels, case.x, case.variables, [50] * bsz)
    gs = structure_of(graph); n, e = gs.n_nodes, gs.n_edges
    with torch.no_grad():
        model(graph)
    h = torch.randn(n, 128, device='cuda'); u = graph.x.float().contiguous(); pos = torch.rand(n, device='cuda'); var = torch.rand(n, 2, device='cuda')
    agg = torch.empty(n, 128, device='cuda'); agg2 = torch.randn(n, 128, device='cuda'); out = torch.empty(n, 128, device='cuda')
    feat = node_features(u, pos, var); tiles = gs.tiles(); tb = ctypes.byref(tiles[0])
    layer, gate = model.gnn_layers[0], model.gnn_layers_gate[0]
    pk, pg = layer.packed(), gate.packed()
    if which == 'tile':
        run = lambda: _lib.check(L.msmp_edge_aggregate_tiled_f32(ptr(h), ptr(u), ptr(pos), ptr(var), ptr(feat), None, None, ptr(gs.rowptr), tb, n, e, 25, 2, ptr(pk), ptr(agg), cs()), 'tile')
    elif which == 'tail':
        run = lambda: _lib.check(L.msmp_node_tail_f32(ptr(h), ptr(agg2), ptr(agg2), ptr(var), ptr(gs.graph_ptr), n, gs.n_graphs, gs.max_graph_nodes, 2, ptr(pk), ptr(pg), 1, 1e-5, ptr(out), cs()), 'tail')
    else:
        lem = model.embedding_lem
        dt = model._dt(u.device); pos_t = torch.rand(n, 1, device='cuda')
        run = lambda: lem.encode_nodes(u, pos[:, None], pos_t, var, dt, False, model.lemoutput_mlp)
    for _ in range(3): run()
    torch.cuda.synchronize()
    th = threading.Thread(target=poll); th.start()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    launches, t0 = 0, time.time()
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50): run()
        launches += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    stop.set(); th.join()
    print(f'{which}: {launches} launches, {e0.elapsed_time(e1) * 1e3 / launches:.1f} us per launch (incl. the sync every 50)')
    mp.last_status(reset=True)
if samples:
    mid = samples[len(samples) // 4:]          # skip the ramp
    print(f'{which}: {len(samples)} rocm-smi samples; power mean {sum(p for p, _ in mid) / len(mid):.0f} W, max {max(p for p, _ in mid):.0f} W; '
          f'sclk mean {sum(s for _, s in mid) / len(mid):.0f} MHz, min {min(s for _, s in mid)} MHz, max {max(s for _, s in mid)} MHz')
    print('  samples (W, MHz):', ' '.join(f'{p:.0f}/{s}' for p, s in samples))
