"""CPU tests (-m "not gpu"): host logic of the product, C-ABI surface, loud failure without the GPU,
and the multi-process sharding path on gloo (world_size 2).  No kernel is launched here."""
import os
import re
import sys

import numpy as np
import pytest
import torch

from helpers import load, sd_of, graph_of, EXPERIMENTS, synthetic_case

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def mp():
    import msmp_pde_amd
    if not os.path.exists(msmp_pde_amd.LIB_PATH):       # hipcc cross-compiles gfx950 without a GPU
        sys.path.insert(0, ROOT)
        import __graft_entry__
        __graft_entry__.build()
    return msmp_pde_amd


def test_cabi_exports_every_declared_symbol(mp):
    """The shared library loads and exports every function include/msmp_pde.h declares, and the ctypes
    table binds exactly that set."""
    from msmp_pde_amd import _lib
    hdr = open(os.path.join(ROOT, 'include', 'msmp_pde.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    declared = set(re.findall(r'\b(msmp_[a-z0-9_]+)\s*\(', hdr))
    assert declared, 'no declarations parsed'
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    L = mp.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert L.msmp_version() >= 100
    assert L.msmp_packed_layer_floats(25, 2) == (2 * (12 + 13) + 9) * 4096 + 4 * 128 + 128 * 8 + 8 + 2048 + 8 * 4096     # fp32 chunks + fp16-split copies + biases + w3v + scales + variable slot fragments + w4t + w2t (+ w1t: 9 more chunks)
    assert L.msmp_packed_layer_floats(25, 99) == -1


def test_argument_errors_are_reported_not_thrown(mp):
    L = mp.lib()
    rc = L.msmp_scatter_mean_f32(None, None, 10, None, None)
    assert rc == -1 and b'null pointer' in L.msmp_last_error()
    rc = L.msmp_mp_layer_f32(*([None] * 10), 1, 1, 1, 6, 100, 25, 2, None, None, 0, 1e-5, None, None, 0, None)
    assert rc == -1
    # every new entry point of this round validates its arguments the same way (no device work happens on these calls)
    assert L.msmp_node_tail_f32(*([None] * 5), 10, 1, 100, 2, None, None, 1, 1e-5, None, None) == -1
    assert L.msmp_build_tiles(None, None, 10, 20, 21, None, None, None, None, None, None) == -1
    assert L.msmp_edge_aggregate_tiled_f32(*([None] * 9), 10, 20, 25, 2, None, None, None) == -1
    assert L.msmp_pack_node_features_f32(None, None, None, 10, 25, 2, None, None) == -1 and L.msmp_node_feature_stride(25, 2) == 32 and L.msmp_node_feature_stride(50, 3) == 64
    assert L.msmp_mlp2_swish_f32(None, 10, 28, None, None, None) == -1
    assert L.msmp_prepare_nodes(None, 0, None, 0, 10, 25, 16.0, 4.0, 0, None, None, None, None, None, None, None, None, None) == -1
    assert L.msmp_adamw_f32(3, None, None, None, None, None, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, None) == -1
    assert L.msmp_lem_encoder_nodes_f32(*([None] * 5), 10, 25, 2, 0, 1.0, None, 1, None, None) == -1
    assert L.msmp_decoder2d_f32(None, None, 10, 25, None, None, None, None, 0.016, None, None) == -1
    assert L.msmp_packed_mlp2_floats(28) == 32 + 256 + (1 + 4) * 4096 and L.msmp_packed_mlp2_floats(129) == -1
    assert L.msmp_mlp2_input_stride(28) == 32 and L.msmp_mlp2_input_stride(59) == 64
    assert L.msmp_mp_layer_bwd_f32(*([None] * 11), 10, 20, 1, 25, 2, None, None, 1, 1e-5, None, None, None, None, 0, None) == -1
    assert L.msmp_mp_layer_bwd_workspace_bytes(0, 5, 25, 2, 1) == 0 and L.msmp_mp_layer_bwd_workspace_bytes(100, 588, 25, 2, 1) > 0
    # knobs: known keys are accepted, unknown ones rejected with a message
    for key in (b'split', b'edge_nb', b'tail', b'pair', b'lem', b'lem_nodes', b'tile'):
        assert L.msmp_tune(key, {b'split': 1, b'tail': 1, b'lem': 3, b'lem_nodes': 1, b'pair': 1, b'tile': 2}.get(key, 0)) == 0, key
    assert L.msmp_tune(b'no_such_knob', 1) != 0 and b'unknown key' in L.msmp_last_error()


@pytest.mark.parametrize('exp', ['E2', 'WE3', 'RPU', 'MSWG3'])
def test_graph_creator_mirror_tensors(mp, exp):
    """Rows G2, R1 on the host: with the edge_index handed in, the vectorised GraphCreator produces the
    reference's tensors bit for bit (x, y, pos, batch, parameter columns; create_next_graph)."""
    d = load(f'graph_{exp}.npz')
    pde_name, eqv, unstructured = EXPERIMENTS[exp]
    kw = dict(tmin=float(d['tmin']), tmax=float(d['tmax']), grid_size=[250, 100])
    pde = {'CE': lambda: mp.CE(L=16., **kw), 'WE': lambda: mp.WE(**kw),
           'AD': lambda: mp.AD(L=16., unstructured=unstructured, **kw)}[pde_name]()
    assert repr(pde) == pde_name and abs(pde.dt - float(d['dt'])) == 0 and pde.L == float(d['L'])
    gc = mp.GraphCreator(pde, neighbors=3, time_window=25, device='cpu')
    u = torch.tensor(d['u_super'].astype(np.float64))
    steps = d['steps'].tolist()
    x = torch.tensor(np.tile(d['x_grid'][None], (len(u), 1)))
    variables = {k[4:]: torch.tensor(v) for k, v in d.items() if k.startswith('var_')}
    data, labels = gc.create_data(u, steps)
    ref = graph_of(d)
    g = gc.create_graph(data, labels, x, variables, steps, edge_index=torch.tensor(ref.edge_index))
    for k in ('x', 'y', 'pos', 'batch', 'alpha', 'beta', 'gamma', 'bc_left', 'bc_right', 'c', 'a', 'b'):
        if hasattr(ref, k):
            assert np.array_equal(getattr(g, k).numpy(), getattr(ref, k)), k
    steps2 = d['steps2'].tolist()
    _, labels2 = gc.create_data(u, steps2)
    g2 = gc.create_next_graph(g, torch.tensor(d['pred'].astype(np.float64)), labels2, steps2)
    ref2 = graph_of(d, 'n_')
    for k in ('x', 'y', 'pos'):
        assert np.array_equal(getattr(g2, k).numpy(), getattr(ref2, k)), k


@pytest.mark.parametrize('kind', ['MP_PDE_Solver', 'MP_PDE_SolverGated', 'MP_PDE_Solver2D', 'MP_PDE_Solver2DGated'])
def test_state_dict_is_reference_compatible(mp, kind):
    """Checkpoint compatibility: the drop-in class has exactly the reference's state_dict keys and shapes."""
    d = load(f'solver_{kind}.npz')
    exp = str(d['experiment'])
    _, eqv, _ = EXPERIMENTS[exp]
    pde = mp.CE() if exp == 'E2' else mp.AD()
    model = getattr(mp, kind)(pde, time_window=25, eq_variables=eqv, hidden_layer=int(d['hidden_layer']))
    ref_sd = sd_of(d)
    sd = model.state_dict()
    assert set(sd) == set(ref_sd)
    for k, v in ref_sd.items():
        assert tuple(sd[k].shape) == v.shape, k
        assert sd[k].dtype == torch.float32
    model.load_state_dict({k: torch.tensor(v).double() for k, v in ref_sd.items()})   # float64 checkpoints load
    assert repr(model) == 'GNN' and model.eq_variables == eqv


@pytest.mark.parametrize('kind,comps', [('MP_PDE_SolverLEMLinGatedGLU', 1), ('MP_PDE_Solver2DLEMLinGatedGLU', 2)])
def test_glu_classes_have_the_reference_layout(mp, kind, comps):
    """The GLU classes cannot be instantiated from the reference here (their LEM needs the absent lem_cuda), so the
    state_dict layout is checked against the constructor arithmetic of experiments/models_gnn.py:1379-1462 /
    models_gnn2D.py:1198-1296: hidden width 164, GNN_LayerLin with time_window = comps * 25, LEM(2 + len(eq_variables) + comps, 164),
    two decoders Conv1d(comps, 8, 6, stride 2) -> Conv1d(8, comps, 15)."""
    eqv = {'beta': 0.2} if comps == 1 else {'a': 1.0, 'b': 1.0}
    nv, W, tw = len(eqv) + 1, 164, 25 * comps
    model = getattr(mp, kind)(mp.CE() if comps == 1 else mp.AD(), time_window=25, eq_variables=eqv, hidden_layer=2)
    sd = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    want = {}
    for stack in ('gnn_layers', 'gnn_layers_gate'):
        for i in range(2):
            pre = f'{stack}.{i}.'
            want.update({pre + 'message_net_1.0.weight': (W, 2 * W + tw + 1 + nv), pre + 'message_net_1.0.bias': (W,),
                         pre + 'message_net_2.0.weight': (W, W), pre + 'message_net_2.0.bias': (W,),
                         pre + 'update_net_1.0.weight': (W, 2 * W + nv), pre + 'update_net_1.0.bias': (W,),
                         pre + 'update_net_2.0.weight': (W, W), pre + 'update_net_2.0.bias': (W,)})
    ninp = 2 + len(eqv) + comps
    want.update({'embedding_lem.rnn.weights': (3 * W, ninp + W), 'embedding_lem.rnn.weights_lin_z': (W, ninp + W),
                 'embedding_lem.rnn.bias': (3 * W,), 'embedding_lem.rnn.bias_lin_z': (W,),
                 'lemoutput_mlp.0.weight': (W, W), 'lemoutput_mlp.0.bias': (W,), 'lemoutput_mlp.2.weight': (W, W), 'lemoutput_mlp.2.bias': (W,)})
    if comps == 2:
        want.update({'double_mlp.0.weight': (2 * W, W), 'double_mlp.0.bias': (2 * W,)})
    for dec in ('output_mlp_gate', 'output_mlp_diff'):
        want.update({dec + '.0.weight': (8, comps, 6), dec + '.0.bias': (8,), dec + '.2.weight': (comps, 8, 15), dec + '.2.bias': (comps,)})
    assert sd == want
    assert repr(model) == 'GNN' and mp.MODEL_NAMES['MSGMP-PDE' if comps == 1 else 'MSGMP-PDE2D'] is type(model)


def test_model_names_cover_the_references_getmodel(mp):
    """experiments/train.py:34-183: every name that constructs a message-passing GNN there constructs the mirror class here with the
    same keyword arguments ('GLEMGated2D' = RGATConv layers and the grid models BaseCNN / FNO / VNO are other model families)."""
    names_1d = ['MP-PDE', 'Gated', 'LEM', 'MSMP-PDE', 'MSSMP-PDE', 'MSGMP-PDE', 'SaveMSMP-PDE', 'LSTMGated', 'LSTM']
    names_2d = ['MP-PDE2D', 'Gated2D', 'MSMP-PDE2D', 'MSGMP-PDE2D', 'SaveMSMP-PDE2D', 'MSG2-PDE2D', 'LSTMGated2D', 'LEM2D', 'LSTM2D']
    assert set(mp.MODEL_NAMES) == set(names_1d + names_2d)
    for name in names_1d + names_2d:
        two_d = name.endswith('2D')
        model = mp.MODEL_NAMES[name](pde=mp.AD() if two_d else mp.CE(), time_window=25, eq_variables={'a': 1.0, 'b': 1.0} if two_d else {'beta': 0.2},
                                     hidden_layer=1)
        assert repr(model) == 'GNN', name
        if name.startswith('Save'):
            assert isinstance(model.embedding_lem, mp.LEMS) and hasattr(model.embedding_lem, 'reset_states')


def test_lem_launch_partition_covers_every_node(mp):
    """lem_partition (lem_kernel.hip): workgroups [0, full) take 96 nodes each from node 96 b, the ones behind them 32 nodes each
    from 96 full + 32 (b - full): for every node count and device size the workgroups must tile [0, n) without gap or overlap, and the
    one-tile round may only be chosen where it is modelled cheaper."""
    import ctypes
    L = ctypes.CDLL(mp.LIB_PATH)
    L.msmp_debug_lem_partition.argtypes = [ctypes.c_int64, ctypes.c_int, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    rng = np.random.default_rng(0)
    cases = [(n, cus) for cus in (1, 7, 128, 256, 304) for n in list(range(1, 400)) + [96 * cus + d for d in (-33, -1, 0, 1, 31, 32, 33, 95, 96, 97)]
             + [204800, 25600, 12800, 3200] + [int(x) for x in rng.integers(1, 3_000_000, 40)] if n > 0]
    mixed = 0
    for n, cus in cases:
        g, f = ctypes.c_int64(), ctypes.c_int64()
        assert L.msmp_debug_lem_partition(n, cus, ctypes.byref(g), ctypes.byref(f)) == 0
        grid, full = g.value, f.value
        assert 0 <= full <= grid and grid >= 1
        covered = 96 * full + 32 * (grid - full)
        assert covered >= n, (n, cus, grid, full)
        if full == grid:                                  # three-tile workgroups only: the last one may be ragged
            assert grid == (n + 95) // 96
        else:                                             # whole rounds of three-tile workgroups, then one-tile workgroups up to n
            assert full % cus == 0 and 96 * full < n and covered - n < 32 and (grid - full) >= 1
            rounds3, rounds_mixed = -(-((n + 95) // 96) // cus), full // cus + 0.5 * -(-(grid - full) // cus)
            assert rounds_mixed <= rounds3 - 0.3 + 1e-9
            mixed += 1
    assert mixed > 50


def test_parameter_counts_match_survey(mp):
    """SURVEY.md section 8 config table (measured on the reference)."""
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert n(mp.MP_PDE_Solver(mp.CE(), eq_variables={'beta': 0.2})) == 636409
    assert n(mp.MP_PDE_SolverGated(mp.CE(), eq_variables={'beta': 0.2})) == 1252345
    assert n(mp.MP_PDE_Solver2DGated(mp.AD(), eq_variables={'a': 1., 'b': 1.})) == 1330410
    lem = mp.MP_PDE_SolverLEMLinGated(mp.CE(), eq_variables={'beta': 0.2})
    assert n(lem) == 1231872 + 68096 + 33024 + 249
    assert set(k for k in lem.state_dict() if 'lem' in k) == {
        'embedding_lem.rnn.weights', 'embedding_lem.rnn.weights_lin_z', 'embedding_lem.rnn.bias',
        'embedding_lem.rnn.bias_lin_z', 'lemoutput_mlp.0.weight', 'lemoutput_mlp.0.bias',
        'lemoutput_mlp.2.weight', 'lemoutput_mlp.2.bias'}


def test_no_cpu_fallback(mp):
    """The product path refuses host tensors instead of silently computing elsewhere."""
    case = synthetic_case(mp, 'E2', bsz=2, seed=1, device='cpu')
    model = mp.MP_PDE_Solver(case.pde, time_window=25, eq_variables=case.eqv, hidden_layer=1)
    with torch.no_grad(), pytest.raises(mp.MsmpError):
        model(case.graph)
    with pytest.raises(mp.MsmpError):            # no structure / host tensors: refused, also under autograd
        layer = model.gnn_layers[0]
        mp.mp_layer(torch.zeros(2, 128), torch.zeros(2, 25), torch.zeros(2, 1), torch.zeros(2, 2), None, layer)


def test_lem_encoder_matches_oracle_cell(mp):
    """The PyTorch LEM restatement (LEMcuda.forward: the kernels' float64 reference in the GPU tests) and the oracle's cell agree
    (both follow the published cell; unpinned); the module itself has no CPU path: host tensors raise."""
    from oracle import msmp_oracle as O
    torch.manual_seed(0)
    lem = mp.LEM(4, 128)
    x = torch.randn(25, 50, 4)
    with pytest.raises(RuntimeError):
        lem(x)
    with torch.no_grad():
        y = lem.rnn(x).double().numpy()
    sd = {k: v.numpy().astype(np.float64) for k, v in lem.state_dict().items()}
    ref = O.lem_forward(x.numpy().astype(np.float64), sd['rnn.weights'], sd['rnn.weights_lin_z'], sd['rnn.bias'],
                        sd['rnn.bias_lin_z'], 1.0)
    assert np.abs(y - ref).max() < 5e-6


def test_shard_range_partitions():
    from msmp_pde_amd.dist import shard_range
    for b in (1, 7, 8, 2048, 2049):
        for w in (1, 2, 3, 4, 8):
            blocks = [shard_range(b, r, w) for r in range(w)]
            assert blocks[0][0] == 0 and blocks[-1][1] == b
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(w - 1))
            sizes = [g1 - g0 for g0, g1 in blocks]
            assert max(sizes) - min(sizes) <= 1


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import msmp_pde_amd as mp_
    from msmp_pde_amd import dist as D
    import torch.distributed as dist
    r, w, _ = D.init_from_env(backend='gloo')
    case = synthetic_case(mp_, 'E2', bsz=5, seed=2, device='cpu')
    sh = D.shard_graph(case.graph, r, w)
    D.barrier()
    n_nodes = D.reduce_scalar(sh.x.shape[0], 'sum')
    n_edges = D.reduce_scalar(sh.edge_index.shape[1], 'sum')
    t_max = D.reduce_scalar(1.0 + r, 'max')
    # every shard is self-contained: edges stay inside the shard and batch ids start at 0
    ok = int(sh.edge_index.min()) >= 0 and int(sh.edge_index.max()) < sh.x.shape[0] and int(sh.batch.min()) == 0
    same = torch.equal(sh.x, case.graph.x[r * 300: r * 300 + sh.x.shape[0]]) if r == 0 else True
    q.put((r, n_nodes, n_edges, t_max, ok and same, case.graph.x.shape[0], case.graph.edge_index.shape[1]))
    dist.destroy_process_group()


def test_sharding_two_ranks_gloo():
    """Row (e): N>1 path on CPU, world_size 2 over gloo: shards partition nodes and edges exactly; the
    max-over-ranks / sum-over-ranks reductions bench.py uses behave."""
    import torch.multiprocessing as tmp
    ctx = tmp.get_context('spawn')
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    for r, n_nodes, n_edges, t_max, ok, full_n, full_e in res:
        assert ok
        assert n_nodes == full_n and n_edges == full_e
        assert t_max == 2.0


def test_bench_launcher_starts_one_rank_per_gpu():
    """`python bench.py --gpus N` without a torchrun environment must start N ranks itself: the parent
    spawns N children with RANK / WORLD_SIZE / MASTER_* set, they join a gloo group and an all-reduce of 1 counts N; with a
    torchrun-style environment whose WORLD_SIZE disagrees with --gpus the run fails non-zero."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--launcher-selftest',
                        '--graphs', '2048'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    out = json.loads(line)
    assert out['ranks_seen'] == 2 and out['n_gpus'] == 2
    assert out['shards'] == [[0, 1024], [1024, 2048]]
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1', MASTER_PORT='29999')
    r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--launcher-selftest'], env=env2,
                        capture_output=True, text=True, timeout=300)
    assert r2.returncode != 0 and 'WORLD_SIZE=1' in r2.stderr


def test_bench_launcher_stops_the_other_ranks_when_one_fails():
    """a rank that dies leaves the others inside a collective until the backend's timeout.  The launcher polls
    its children: rank 1 exits with code 3 before joining the group, rank 0 (blocked in the rendezvous) is terminated, the
    launcher returns 3 within seconds and replays the failing rank's stderr."""
    import subprocess
    import time
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dist-backend', 'gloo', '--launcher-selftest',
                        '--selftest-fail-rank', '1'], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 3, (r.returncode, r.stderr[-1500:])
    assert time.time() - t0 < 120
    assert 'rank 1 exited with code 3' in r.stderr and '[rank 1] selftest: this rank fails on purpose' in r.stderr


def test_decoder_convolution_as_toeplitz_gemm_matches_conv1d():
    """solvers._conv1d_as_matmul (the AUTOGRAD decoder: Conv1d as one dense GEMM against the convolution's Toeplitz matrix, itself the
    weight times a fixed shift tensor) against torch.nn.functional.conv1d, forward and all three gradients, for every decoder geometry
    of the reference (models_gnn.py:210-224, models_gnn2D.py:79-88) in float64 on the CPU."""
    import torch
    from msmp_pde_amd.solvers import _conv1d_as_matmul
    torch.manual_seed(0)
    for cin, cout, k, s, lin in [(1, 8, 16, 3, 128), (8, 1, 14, 1, 38), (2, 8, 16, 3, 128), (8, 2, 14, 1, 38), (1, 8, 15, 4, 128), (8, 1, 10, 1, 29),
                                 (1, 8, 12, 2, 128), (8, 1, 10, 1, 59), (2, 8, 12, 2, 128), (8, 2, 10, 1, 59)]:
        x = torch.randn(5, cin, lin, dtype=torch.float64, requires_grad=True)
        w = torch.randn(cout, cin, k, dtype=torch.float64, requires_grad=True)
        b = torch.randn(cout, dtype=torch.float64, requires_grad=True)
        y, yr = _conv1d_as_matmul(x, w, b, s), torch.nn.functional.conv1d(x, w, b, stride=s)
        assert y.shape == yr.shape and (y - yr).abs().max().item() < 1e-12
        g = torch.randn_like(y)
        for ga, gr in zip(torch.autograd.grad(y, (x, w, b), g), torch.autograd.grad(yr, (x, w, b), g)):
            assert (ga - gr).abs().max().item() < 1e-11
