"""GPU tests of the training path: autograd through the HIP forward (recompute backward), and exact
graph-sharded data parallelism of the reference's sqrt-of-sum loss with the gradient all-reduce, on two
ranks (gloo, both ranks on the one GPU of the box; with the nccl backend the same code runs over RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import msmp_oracle as O
from oracle import msmp_oracle_torch as OT
from helpers import synthetic_case, fp32_floors, assert_parity, record_parity, err_stats

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TW = 25


@pytest.fixture(scope='module')
def mp():
    import msmp_pde_amd
    assert torch.cuda.is_available()
    return msmp_pde_amd


@pytest.mark.parametrize('kind,exp', [('MP_PDE_Solver', 'E2'), ('MP_PDE_SolverGated', 'E2'), ('MP_PDE_Solver2DGated', 'RPU'),
                                      ('MP_PDE_SolverLEMLinGated', 'E2'), ('MP_PDE_Solver2DLEMLinGated', 'RPU'),
                                      ('MP_PDE_SolverLEMLin', 'E2'), ('MP_PDE_Solver2DLEMLinG2', 'RPU'),
                                      ('MSSMP_PDE_Solver', 'E2'), ('MP_PDE_SolverGated', 'WE3'), ('MP_PDE_SolverLEMLinGated', 'WE3'),
                                      ('MP_PDE_Solver2DLEMLinGated', 'MSWG3'), ('MP_PDE_Solver2D', 'RPU'),
                                      ('MP_PDE_SolverLSTMLinGated', 'E2'), ('MP_PDE_Solver2DLSTMLin', 'RPU'),
                                      ('MP_PDE_SolverLEMLinGatedGLU', 'E2'), ('MP_PDE_Solver2DLEMLinGatedGLU', 'MSWG3')])
def test_gradients_match_float64_oracle(mp, kind, exp):
    """d loss / d parameters of the product (HIP forward, recompute backward, fp32) against torch autograd through
    the float64 oracle, for the reference's training loss sqrt(sum (pred - y)^2) (train_helper.py:126,138)."""
    torch.manual_seed(2)
    case = synthetic_case(mp, exp, bsz=3, seed=4)
    model = getattr(mp, kind)(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=2).cuda()
    graph = case.graph.to('cuda')
    pred = model(graph)
    loss = torch.sqrt(((pred - graph.y.to(pred.dtype)) ** 2).sum())
    loss.backward()

    sd64 = {k: v.detach().cpu().double().requires_grad_(True) for k, v in model.state_dict().items()}
    out = OT.solver_forward(kind, sd64, case.graph_np(), case.pde, TW, case.eqv, 2, as_numpy=False)
    y = torch.tensor(case.graph_np().y).double()
    ref_loss = torch.sqrt(((out - y) ** 2).sum())
    ref_loss.backward()
    assert abs(loss.item() - ref_loss.item()) < 1e-4 * ref_loss.item()
    # scale = largest gradient entry of the whole model: a bias that feeds an InstanceNorm directly
    # (GNN_LayerLin.update_net_2.bias) has an exactly-zero gradient, so per-tensor relative error is meaningless there
    scale = max(sd64[name].grad.abs().max().item() for name, _ in model.named_parameters())
    worst = 0.0
    lin_layers = 'Gated' in kind or kind.endswith('G2') or kind.startswith('MSSMP')
    for name, p in model.named_parameters():
        g, r = p.grad.double().cpu(), sd64[name].grad
        if lin_layers and name.endswith('update_net_2.0.bias'):
            # GNN_LayerLin feeds its InstanceNorm directly: this gradient is analytically zero (both sides are rounding
            # residue of sums of terms of size `scale`), so it gets an absolute bound instead of a relative one
            assert (g - r).abs().max().item() < 1e-4 * scale, (name, (g - r).abs().max().item(), scale)
            continue
        rel = (g - r).abs().max().item() / max(r.abs().max().item(), 1e-3 * scale)
        worst = max(worst, rel)
        assert rel < 2e-3, (name, rel)
    print(f'{kind}/{exp}: loss {loss.item():.6f} (oracle {ref_loss.item():.6f}); worst relative gradient error {worst:.2e}')


def _dp_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import msmp_pde_amd as mp_
    from msmp_pde_amd import dist as D, train as T
    D.init_from_env(backend='gloo')
    torch.manual_seed(5)                                    # same weights everywhere
    case = synthetic_case(mp_, 'E2', bsz=4, seed=6)
    model = mp_.MP_PDE_SolverGated(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=2).cuda()
    shard = D.shard_graph(case.graph, rank, world).to('cuda')
    loss = T.dp_loss_backward(model, shard)
    if rank == 0:
        torch.save({'loss': loss.item(), 'grads': {k: p.grad.cpu() for k, p in model.named_parameters()}}, out_path)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_data_parallel_gradients_are_exact(mp, tmp_path):
    """Two ranks, each with half of the graphs: after the S all-reduce, the scaled backward and the flat gradient
    all-reduce, every rank holds the single-device gradient of sqrt(sum over ALL nodes)."""
    import torch.multiprocessing as tmp
    from msmp_pde_amd import train as T
    out_path = str(tmp_path / 'dp.pt')
    ctx = tmp.get_context('spawn')
    port = 29700 + (os.getpid() % 200)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, out_path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    dp = torch.load(out_path, weights_only=True)

    torch.manual_seed(5)
    case = synthetic_case(mp, 'E2', bsz=4, seed=6)
    model = mp.MP_PDE_SolverGated(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=2).cuda()
    loss = T.dp_loss_backward(model, case.graph.to('cuda'))         # world size 1: plain single-device step
    assert abs(loss.item() - dp['loss']) < 1e-5 * abs(loss.item())
    scale = max(p.grad.abs().max().item() for p in model.parameters())
    for k, p in model.named_parameters():
        ref = p.grad.cpu()
        err = (dp['grads'][k] - ref).abs().max().item()
        # fp32 summation-order noise only (shard sums are added in another order): relative to the tensor's own largest
        # entry, plus an absolute floor for the gradients that are analytically zero (a bias feeding an InstanceNorm
        # directly: column sums of the norm's backward, pure rounding residue of terms of size `scale`)
        assert err < 1e-4 * ref.abs().max().item() + 1e-5 * scale, (k, err, ref.abs().max().item(), scale)


def _rccl_worker(port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import msmp_pde_amd as mp_
    from msmp_pde_amd import dist as D, train as T
    D.init_from_env(backend='nccl', force_group=True)       # a ONE-rank RCCL group: the collectives below really run on the backend
    assert torch.distributed.is_initialized() and torch.distributed.get_backend() == 'nccl' and T._world() == 1
    ranks_seen = int(round(D.reduce_scalar(1, 'sum')))      # bench.py's rank count, an all-reduce on the GPU
    torch.manual_seed(5)
    case = synthetic_case(mp_, 'E2', bsz=4, seed=6)
    model = mp_.MP_PDE_SolverGated(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=2).cuda()
    loss = T.dp_loss_backward(model, case.graph.to('cuda'))
    D.barrier()
    torch.save({'loss': loss.item(), 'ranks_seen': ranks_seen, 'grads': {k: p.grad.cpu() for k, p in model.named_parameters()}}, out_path)
    torch.distributed.destroy_process_group()


def test_rccl_path_runs_on_one_gpu(mp, tmp_path):
    """`backend='nccl'` (RCCL) executed for real -- a one-rank group on the box's single GPU runs bench.py's
    rank count and training's two all-reduces (the scalar S and the flat gradient buffer); the gradients are those of the
    ungrouped run, bit for bit (a one-rank SUM all-reduce is the identity).  In a child process: a process group per test
    process would otherwise outlive the test."""
    import torch.multiprocessing as tmp
    from msmp_pde_amd import train as T
    out_path = str(tmp_path / 'rccl.pt')
    ctx = tmp.get_context('spawn')
    p = ctx.Process(target=_rccl_worker, args=(29900 + (os.getpid() % 90), out_path))
    p.start()
    p.join(300)
    assert p.exitcode == 0
    got = torch.load(out_path, weights_only=True)
    assert got['ranks_seen'] == 1
    torch.manual_seed(5)
    case = synthetic_case(mp, 'E2', bsz=4, seed=6)
    model = mp.MP_PDE_SolverGated(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=2).cuda()
    loss = T.dp_loss_backward(model, case.graph.to('cuda'))         # no process group here
    assert loss.item() == got['loss']
    for k, p_ in model.named_parameters():
        assert torch.equal(p_.grad.cpu(), got['grads'][k]), k


def test_training_step_runs_and_decreases_loss(mp):
    """training_step (pushforward unrolling + AdamW) on the HIP path: the loss on a fixed batch goes down."""
    from msmp_pde_amd import train as T
    from msmp_pde_amd.synthetic import make_case
    torch.manual_seed(7)
    c = make_case('E2', 8, seed=9, device='cuda', dtype=torch.float32)
    model = mp.MP_PDE_SolverLEMLinGated(c.pde, time_window=TW, eq_variables=c.eqv, hidden_layer=2).cuda()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-3)
    losses = [T.training_step(model, c.creator, c.u_super, c.x, c.variables, [60] * 8, 1, opt).item() for _ in range(6)]
    print('losses', [round(l, 4) for l in losses])
    assert losses[-1] < losses[0]
    # the same loop with the optimisation step as one hipGraph launch (pushforward forwards eager, reading the replayed updates)
    opt2 = mp.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
    data, labels = c.creator.create_data(c.u_super, [60] * 8)
    step = T.CapturedTrainStep(model, opt2, c.creator.create_graph(data, labels, c.x, c.variables, [60] * 8))
    more = [T.training_step(model, c.creator, c.u_super, c.x, c.variables, [60] * 8, 1, opt2, captured=step).item() for _ in range(6)]
    print('captured', [round(l, 4) for l in more])
    assert more[-1] < more[0] < losses[0]


@pytest.mark.parametrize('ninp,n,t_len', [(3, 77, 25), (4, 300, 25), (6, 129, 50), (8, 40, 7), (1, 33, 3)])
def test_lem_training_kernels_match_float64_restatement(mp, ninp, n, t_len):
    """msmp_lem_train_fwd_f32 / msmp_lem_train_bwd_f32 (through LEM.forward_nodes) against the float64 PyTorch
    restatement of the cell (LEMcuda.forward): final state and the gradients of the four parameters (the reference's
    LEMFunction returns exactly these, experiments/models_gnn.py:296-302).  fp32 vs float64: 1e-5 on the state
    (the north-star tolerance), 1e-4 of the largest entry on the gradients (sums over N*T rows in fp32)."""
    import copy
    from msmp_pde_amd.lem import LEM
    torch.manual_seed(11 + ninp)
    lem = LEM(ninp, 128).cuda()
    ref = copy.deepcopy(lem).double()
    xin = torch.randn(n, t_len, ninp, device='cuda') * 0.7
    w_out = torch.randn(n, 128, device='cuda')
    y = lem.forward_nodes(xin)
    (y * w_out).sum().backward()
    y64 = ref.rnn(xin.double().permute(1, 0, 2).contiguous())
    (y64 * w_out.double()).sum().backward()
    err = (y.double() - y64).abs().max().item()
    assert err < 1e-5, err
    worst = 0.0
    for (k, p), q in zip(lem.named_parameters(), ref.parameters()):
        rel = (p.grad.double() - q.grad).abs().max().item() / q.grad.abs().max().item()
        worst = max(worst, rel)
        assert rel < 1e-4, (k, rel)
    print(f'LEM train ninp={ninp} N={n} T={t_len}: state err {err:.2e}, worst gradient error {worst:.2e}')
    # the reference layout [T, N, ninp] goes through the same kernels
    y2 = lem(xin.permute(1, 0, 2).contiguous())
    assert torch.equal(y2, y)


def _ragged(sizes):
    gp = torch.tensor([0] + list(np.cumsum(sizes)), dtype=torch.int32, device='cuda')
    batch = torch.repeat_interleave(torch.arange(len(sizes)), torch.tensor(sizes)).cuda()
    return gp, batch


def test_backward_glue_kernels_match_float64_autograd(mp):
    """The four glue kernels of the layer backward (train_kernels.hip) through the C-ABI, each against float64 autograd of
    the formula it differentiates; ragged graphs (1, 2, 3, 100, 130 nodes), nodes without in-edges, tw + 1 + nv odd."""
    from msmp_pde_amd._lib import check, ptr, current_stream
    from msmp_pde_amd import autograd as A
    L = mp.lib()
    g = torch.Generator(device='cpu').manual_seed(5)
    sizes = [1, 100, 3, 130, 2]
    gp, batch = _ragged(sizes)
    n, b, eps = sum(sizes), len(sizes), 1e-5
    r = lambda *s: torch.randn(*s, generator=g).cuda()

    # InstanceNorm backward
    x, gy = r(n, 128), r(n, 128)
    dx = torch.empty_like(x)
    check(L.msmp_instance_norm_bwd_f32(ptr(x), ptr(gy), ptr(gp), b, eps, ptr(dx), current_stream()), 'in_bwd')
    x64 = x.double().requires_grad_(True)
    A._instance_norm(x64, batch, b, eps).backward(gy.double())
    multi = (batch != 0) & (batch != 4)          # 1- and 2-node graphs: y = 0 / +-1, the derivative is ill-conditioned
    assert (dx.double() - x64.grad)[multi].abs().max().item() < 2e-5
    assert torch.isfinite(dx).all()

    # gated blend backward through both norms
    h, gate, main, go = r(n, 128), r(n, 128), r(n, 128), r(n, 128)
    dg, dm, dh = torch.empty_like(h), torch.empty_like(h), torch.empty_like(h)
    check(L.msmp_gate_blend_bwd_f32(ptr(go), ptr(h), ptr(gate), ptr(main), ptr(gp), b, eps, ptr(dg), ptr(dm), ptr(dh),
                                    current_stream()), 'blend_bwd')
    h64, g64, m64 = (t.double().requires_grad_(True) for t in (h, gate, main))
    tau = torch.sigmoid(A._instance_norm(g64, batch, b, eps))
    ((1 - tau) * h64 + tau * A._swish(A._instance_norm(m64, batch, b, eps))).backward(go.double())
    for got, ref in ((dg, g64.grad), (dm, m64.grad), (dh, h64.grad)):
        assert (got.double() - ref)[multi].abs().max().item() < 2e-5

    # edge concat (bit-exact) and mean backward + Swish'
    for tw, nv in ((25, 2), (25, 3), (50, 3)):
        e = 700
        tgt = torch.sort(torch.randint(0, n - 7, (e,), generator=g))[0].int().cuda()      # the last nodes have no in-edges
        col = torch.randint(0, n, (e,), generator=g).int().cuda()
        rowptr = torch.zeros(n + 1, dtype=torch.int32, device='cuda')
        rowptr[1:] = torch.cumsum(torch.bincount(tgt.long(), minlength=n), 0).int()
        u, pos, var = r(n, tw), r(n), r(n, nv)
        k = 256 + tw + 1 + nv
        ld = (k + 3) // 4 * 4
        out = torch.full((e, ld), float('nan'), device='cuda')
        check(L.msmp_edge_concat_f32(ptr(h), ptr(u), ptr(pos), ptr(var), ptr(tgt), ptr(col), e, tw, nv, ld, ptr(out),
                                     current_stream()), 'edge_concat')
        i, j = tgt.long(), col.long()
        ref = torch.cat((h[i], h[j], u[i] - u[j], (pos[i] - pos[j])[:, None], var[i]), 1)
        assert torch.equal(out[:, :k], ref)
        a2, dagg = r(e, 128), r(n, 128)
        got = torch.empty_like(a2)
        check(L.msmp_mean_bwd_dswish_f32(ptr(dagg), ptr(rowptr), ptr(tgt), ptr(a2), e, ptr(got), current_stream()), 'mean_bwd')
        a64 = a2.double().requires_grad_(True)
        A._seg_mean(A._swish(a64), i, n).backward(dagg.double())
        assert (got.double() - a64.grad).abs().max().item() < 1e-5


def test_explicit_backward_equals_autograd_recompute(mp):
    """The three backward paths of a layer -- msmp_mp_layer_bwd_f32 (2, the default), the same algorithm orchestrated from Python
    with library GEMMs (1), torch.autograd over the PyTorch restatement (0) -- all fp32 on the GPU, for the three layer forms
    (residual-Swish, Lin, gated pair)."""
    from msmp_pde_amd import autograd as A
    from msmp_pde_amd.synthetic import make_case
    c = make_case('E2', 5, seed=3, device='cuda', dtype=torch.float32)
    assert A.EXPLICIT_BACKWARD == 2
    for name in ('MP-PDE', 'Gated'):
        grads = {}
        for path in (2, 1, 0):
            A.EXPLICIT_BACKWARD = path
            try:
                torch.manual_seed(1)
                model = mp.MODEL_NAMES[name](c.pde, time_window=TW, eq_variables=c.eqv, hidden_layer=2).cuda()
                data, labels = c.creator.create_data(c.u_super, [60] * 5)
                graph = c.creator.create_graph(data, labels, c.x, c.variables, [60] * 5)
                torch.sqrt(((model(graph) - graph.y) ** 2).sum()).backward()
                grads[path] = {k: p.grad.clone() for k, p in model.named_parameters()}
            finally:
                A.EXPLICIT_BACKWARD = 2
        scale = max(v.abs().max().item() for v in grads[0].values())
        for path in (2, 1):
            for k in grads[0]:
                err = (grads[path][k] - grads[0][k]).abs().max().item()
                assert err < 1e-4 * grads[0][k].abs().max().item() + 1e-5 * scale, (name, path, k, err)


def test_grad_weights_kernel_shapes_and_strides(mp):
    """msmp_grad_weights_f32 through autograd.grad_weights: dW = A^T B and db = column sums for 8 pairs in one call, row
    counts that are no multiple of the 8-row step / 128-row split (1 ... 70 001), k2 from 1 to the maximum 319, row-strided
    views for both operands; against float64.  Also: repeatable bit for bit (fixed-order reduction) and the size limit."""
    from msmp_pde_amd.autograd import grad_weights
    g = torch.Generator(device='cpu').manual_seed(9)
    shapes = [(1, 1), (7, 128), (129, 136), (1000, 159), (1601, 160), (9408, 284), (70001, 311), (300, 319)]
    pairs, refs = [], []
    for i, (rows, k2) in enumerate(shapes):
        a_full = torch.randn(rows, 512, generator=g).cuda()
        b_full = torch.randn(rows, k2 + 5, generator=g).cuda()
        a = a_full[:, 128:256] if i % 2 else a_full[:, :128].contiguous()       # lda = 512 or 128
        b = b_full[:, :k2]                                                      # ldb = k2 + 5
        pairs.append((a, b))
        refs.append((a.double().t() @ b.double(), a.double().sum(0)))
    out = grad_weights(pairs)
    for i, (rw, rb) in enumerate(refs):
        dw, db = out[2 * i], out[2 * i + 1]
        assert dw.shape == rw.shape and db.shape == rb.shape
        tol = 2e-6 * (shapes[i][0] ** 0.5 + 4)                                  # fp32 sums of `rows` O(1) terms
        assert (dw.double() - rw).abs().max().item() < tol * 4, (shapes[i], (dw.double() - rw).abs().max().item())
        assert (db.double() - rb).abs().max().item() < tol * 4
    again = grad_weights(pairs)
    assert all(torch.equal(x, y) for x, y in zip(out, again))
    with pytest.raises(ValueError):
        grad_weights([(pairs[0][0], torch.randn(1, 320).cuda())])              # k2 > 319


@pytest.fixture(params=[0, 2], ids=['rocblas', 'own_gemm'])
def bwd_gemm(request, mp):
    """Both GEMM paths of msmp_mp_layer_bwd_f32 (tune "bwd_gemm": 0 = rocblas_sgemm + separate epilogues, 2 = rows_gemm_kernel; the
    default picks by size, which at test sizes would always be the library)."""
    from msmp_pde_amd._lib import check
    L = mp.lib()
    prev = L.msmp_tune_query(b'bwd_gemm')
    check(L.msmp_tune(b'bwd_gemm', request.param), 'tune')
    yield request.param
    check(L.msmp_tune(b'bwd_gemm', prev), 'tune')


@pytest.mark.parametrize('form', ['residual', 'lin', 'gated'])
@pytest.mark.parametrize('n_edges', [0, 900])
def test_layer_backward_entry_vs_float64_autograd(mp, form, n_edges, bwd_gemm):
    """msmp_mp_layer_bwd_f32 through the C-ABI on ragged graphs (1, 100, 3, 130, 2 nodes; nodes without in-edges; an edgeless
    batch): dL/dh and all parameter gradients against float64 torch.autograd over the restatement of the layer."""
    from msmp_pde_amd import autograd as A
    from msmp_pde_amd.graph import GraphStructure
    g = torch.Generator(device='cpu').manual_seed(21)
    sizes = [1, 100, 3, 130, 2]
    gp, batch = _ragged(sizes)
    n, tw, nv, eps = sum(sizes), 25, 3, 1e-5
    r = lambda *s: torch.randn(*s, generator=g).cuda()
    # edges inside graph 1 and graph 3 only, no self-loops needed for the test; the last nodes of each stay without in-edges
    src = torch.cat([torch.randint(1, 90, (n_edges // 2,), generator=g), torch.randint(104, 220, (n_edges - n_edges // 2,), generator=g)])
    dst = torch.cat([torch.randint(1, 90, (n_edges // 2,), generator=g), torch.randint(104, 220, (n_edges - n_edges // 2,), generator=g)])
    ei = torch.stack([src, dst]).cuda()
    gs = GraphStructure(ei, batch, n)
    h, u, pos, var, gout = r(n, 128), r(n, tw), r(n), r(n, nv), r(n, 128)
    k1, k3 = 256 + tw + 1 + nv, 256 + nv
    def params():
        return [r(128, k1) / k1 ** 0.5, r(128) * 0.1, r(128, 128) / 128 ** 0.5, r(128) * 0.1,
                r(128, k3) / k3 ** 0.5, r(128) * 0.1, r(128, 128) / 128 ** 0.5, r(128) * 0.1]
    ps = params() + (params() if form == 'gated' else [])
    dh, grads = A.layer_backward_native(gout, h, u, pos, var, gs, ps, form != 'residual', form == 'gated', eps)
    dh2, grads2 = A.layer_backward_native(gout, h, u, pos, var, gs, ps, form != 'residual', form == 'gated', eps)
    # no atomics anywhere (the source-side scatter walks the edges regrouped by source): bitwise reproducible
    assert torch.equal(dh, dh2) and all(torch.equal(a, b) for a, b in zip(grads, grads2))
    h64 = h.double().requires_grad_(True)
    p64 = [q.double().requires_grad_(True) for q in ps]
    s64, d64 = gs.col_long, gs.tgt_long
    args = (u.double(), pos.double(), var.double(), s64, d64, batch, len(sizes))
    if form == 'gated':
        tau = torch.sigmoid(A.layer_reference(h64, *args, p64[8:], True, eps))
        out = (1 - tau) * h64 + tau * A._swish(A.layer_reference(h64, *args, p64[:8], True, eps))
    else:
        out = A.layer_reference(h64, *args, p64, form == 'lin', eps)
    out.backward(gout.double())
    multi = ((batch != 0) & (batch != 4))[:, None]       # 1- and 2-node graphs: InstanceNorm's derivative is ill-conditioned
    scale = max(q.grad.abs().max().item() for q in p64)
    assert ((dh.double() - h64.grad) * multi).abs().max().item() < 2e-4 * max(h64.grad.abs().max().item(), 1.0)
    for i, (got, ref) in enumerate(zip(grads, p64)):
        assert got.shape == ref.shape
        assert (got.double() - ref.grad).abs().max().item() < 1e-4 * ref.grad.abs().max().item() + 2e-5 * scale, (form, n_edges, i)


@pytest.mark.parametrize('name', ['MSMP-PDE', 'MP-PDE'])
def test_packed_weights_follow_fused_optimizer_and_data_edits(mp, name):
    """The kernel-layout weight caches must follow every parameter update.  torch.optim.AdamW(fused=True) updates parameters
    WITHOUT bumping Tensor._version (found by a training soak: the forward kept using the packed weights of step 0), so the cache
    keys also carry an epoch that a process-wide optimizer-step hook advances; edits through `.data` need
    invalidate_packed_weights().  Checked against a fresh model loaded with the same state_dict, in train and eval mode."""
    from msmp_pde_amd import train as T
    from msmp_pde_amd.synthetic import make_case
    torch.manual_seed(1)
    c = make_case('E2', 4, seed=2, device='cuda', dtype=torch.float32)
    cls = mp.MODEL_NAMES[name]
    model = cls(c.pde, time_window=TW, eq_variables=c.eqv, hidden_layer=2).cuda().train()
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2, fused=True)
    steps = [60] * 4
    data, labels = c.creator.create_data(c.u_super, steps)
    graph = c.creator.create_graph(data, labels, c.x, c.variables, steps)

    def fresh_outputs():
        fresh = cls(c.pde, time_window=TW, eq_variables=c.eqv, hidden_layer=2).cuda()
        fresh.load_state_dict(model.state_dict())
        with torch.no_grad():
            return fresh.train()(graph), fresh.eval()(graph)

    with torch.no_grad():
        before = model(graph)
    for _ in range(3):
        T.training_step(model, c.creator, c.u_super, c.x, c.variables, steps, 1, opt)
    want_train, want_eval = fresh_outputs()
    with torch.no_grad():
        assert (model(graph) - before).abs().max().item() > 1e-3          # the weights did move
        assert torch.equal(model.train()(graph), want_train)
        assert torch.equal(model.eval()(graph), want_eval)
    with torch.no_grad():
        for p in model.parameters():
            p.data.mul_(1.01)                                               # invisible to Tensor._version
    mp.invalidate_packed_weights()
    want_train, _ = fresh_outputs()
    with torch.no_grad():
        assert torch.equal(model.train()(graph), want_train)


@pytest.mark.parametrize('ninp,n,t_len', [(4, 77, 2), (6, 200, 4)])
def test_lems_state_carry(mp, ninp, n, t_len):
    """LEMS (models_gnn.py:345-362) on the state-taking recurrence kernel: three consecutive calls with short sequences (so the
    carried (y, z) dominate the result) against the float64 restatement carrying the states explicitly, under no_grad (saved =
    NULL path) and with autograd (third call: parameter gradients with the carried states as constants); reset_states()."""
    import copy
    from msmp_pde_amd.lem import LEMS
    torch.manual_seed(3 + ninp)
    lem = LEMS(ninp, 128).cuda()
    ref = copy.deepcopy(lem).double()
    xs = [torch.randn(n, t_len, ninp, device='cuda') for _ in range(3)]
    states = None
    with torch.no_grad():
        for k in range(2):
            y = lem.forward_nodes(xs[k])
            y64, z64 = ref.rnn(xs[k].double().permute(1, 0, 2).contiguous(), states, return_state=True)
            states = (y64, z64)
            assert (y.double() - y64).abs().max().item() < 1e-5
            assert (lem.states[1].double() - z64).abs().max().item() < 1e-5
    y = lem.forward_nodes(xs[2])                                   # with autograd, from carried states
    w_out = torch.randn(n, 128, device='cuda')
    (y * w_out).sum().backward()
    y64 = ref.rnn(xs[2].double().permute(1, 0, 2).contiguous(), tuple(s.detach() for s in states))
    (y64 * w_out.double()).sum().backward()
    assert (y.double() - y64).abs().max().item() < 1e-5
    for (k, p), q in zip(lem.named_parameters(), ref.parameters()):
        assert (p.grad.double() - q.grad).abs().max().item() < 1e-4 * q.grad.abs().max().item(), k
    # the states really matter at this sequence length, and reset_states() drops them
    with torch.no_grad():
        lem.reset_states()
        y_reset = lem.forward_nodes(xs[2])
        y_zero = ref.rnn(xs[2].double().permute(1, 0, 2).contiguous())
    assert (y_reset.double() - y_zero).abs().max().item() < 1e-5
    assert (y_reset - y.detach()).abs().max().item() > 1e-2


@pytest.mark.parametrize('name,exp', [('MP-PDE', 'E2'), ('Gated2D', 'MSWG3')])
def test_output_mlp_called_the_reference_way_never_reaches_miopen(mp, name, exp):
    """`model.output_mlp(h[:, None])` under autograd (what reference-style code does, models_gnn.py:278) must
    not dispatch MIOpen's convolution (its backward-data kernel faulted on MI355X).  The decoder modules are nn.Conv1d subclasses
    with an unfold + einsum forward: same state_dict keys, same values as F.conv1d, gradients equal to float64 autograd, and the
    profiler sees no convolution op."""
    import torch.nn.functional as F
    torch.manual_seed(1)
    case = synthetic_case(mp, exp, bsz=2, seed=3)
    model = mp.MODEL_NAMES[name](case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=1).cuda()
    assert set(k for k in model.state_dict() if k.startswith('output_mlp')) == {'output_mlp.0.weight', 'output_mlp.0.bias', 'output_mlp.2.weight', 'output_mlp.2.bias'}
    comps = 2 if '2D' in name else 1
    x = torch.randn(50, comps, 128, device='cuda', requires_grad=True)
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
        out = model.output_mlp(x)
        out.square().sum().backward()
    ops = {e.key for e in prof.key_averages()}
    assert not any('convolution' in o or 'miopen' in o.lower() for o in ops), sorted(o for o in ops if 'conv' in o.lower())
    c1, c2 = model.output_mlp[0], model.output_mlp[2]
    x64 = x.detach().double().cpu().requires_grad_(True)
    mid = F.conv1d(x64, c1.weight.detach().double().cpu(), c1.bias.detach().double().cpu(), stride=c1.stride[0])
    ref = F.conv1d(mid * torch.sigmoid(mid), c2.weight.detach().double().cpu(), c2.bias.detach().double().cpu())
    ref.square().sum().backward()
    assert (out.detach().double().cpu() - ref.detach()).abs().max().item() < 1e-5
    assert (x.grad.double().cpu() - x64.grad).abs().max().item() < 1e-4 * x64.grad.abs().max().item()


def test_fused_adamw_matches_torch(mp):
    """msmp_pde_amd.optim.AdamW (one HIP launch per 48 tensors) against torch.optim.AdamW over several steps of a real model
    (108 tensors of very different sizes, incl. 1-element and odd-sized ones), with a learning-rate change in between (MultiStepLR,
    train.py:411); the packed-weight caches are invalidated by the step like with any optimizer."""
    from msmp_pde_amd import _lib
    torch.manual_seed(7)
    case = synthetic_case(mp, 'E2', bsz=2, seed=3)
    a = mp.MP_PDE_SolverLEMLinGated(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=2).cuda()
    b = mp.MP_PDE_SolverLEMLinGated(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=2).cuda()
    b.load_state_dict(a.state_dict())
    oa = mp.optim.AdamW(a.parameters(), lr=1e-3)
    ob = torch.optim.AdamW(b.parameters(), lr=1e-3)
    sa = torch.optim.lr_scheduler.MultiStepLR(oa, milestones=[2], gamma=0.4)
    sb = torch.optim.lr_scheduler.MultiStepLR(ob, milestones=[2], gamma=0.4)
    gen = torch.Generator(device='cuda').manual_seed(1)
    for it in range(4):
        epoch0 = _lib.PARAM_EPOCH[0]
        for pa, pb in zip(a.parameters(), b.parameters()):
            g = torch.randn(pa.shape, device='cuda', generator=gen) * (0.1 + it)
            pa.grad, pb.grad = g.clone(), g.clone()
        oa.step(); ob.step(); sa.step(); sb.step()
        assert _lib.PARAM_EPOCH[0] > epoch0
        for (name, pa), pb in zip(a.named_parameters(), b.parameters()):
            err = (pa - pb).abs().max().item()
            assert err <= 2e-7 * max(1.0, pb.abs().max().item()) + 1e-9, (it, name, err)
    st = oa.state_dict()
    assert set(st['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
    # torch's state layout for real: one `step` tensor per parameter, so that a state_dict round trip through
    # torch.save / torch.load into torch.optim.AdamW resumes with step + 1 per step (a shared tensor advanced it by the number of
    # parameters), and the continued trajectories of the two optimizers agree
    import io
    assert len({id(oa.state[p]['step']) for p in a.parameters()}) == len(list(a.parameters()))
    buf = io.BytesIO()
    torch.save(oa.state_dict(), buf)
    buf.seek(0)
    oc = torch.optim.AdamW(a.parameters(), lr=1e-3)
    oc.load_state_dict(torch.load(buf, weights_only=True))
    for pa, pb in zip(a.parameters(), b.parameters()):
        g = torch.randn(pa.shape, device='cuda', generator=gen)
        pa.grad, pb.grad = g.clone(), g.clone()
    oc.step(); ob.step()
    assert all(float(oc.state[p]['step']) == 5.0 for p in a.parameters())
    for (name, pa), pb in zip(a.named_parameters(), b.parameters()):
        err = (pa - pb).abs().max().item()
        assert err <= 4e-7 * max(1.0, pb.abs().max().item()) + 1e-9, (name, err)
    # a parameter that gets its first gradient later than its group peers, and a storage that was replaced (p.data = ...): handled
    w = [torch.nn.Parameter(torch.randn(5, 3, device='cuda')) for _ in range(2)]
    od = mp.optim.AdamW(w, lr=1e-2)
    w[0].grad = torch.ones_like(w[0])
    od.step()
    w[1].grad = torch.ones_like(w[1])
    w[0].data = w[0].data.clone()
    od.step()
    torch.cuda.synchronize()
    assert float(od.state[w[0]]['step']) == 2.0 and float(od.state[w[1]]['step']) == 1.0


def test_deterministic_reduction_kernels(mp):
    """msmp_colsum_f32 / msmp_sqerr_sum_f32 (the bias gradients and the loss of a captured training step) against float64, and
    bit-identical from run to run; the autograd wrappers route the encoder's Linear and the decoder's Conv1d bias gradients."""
    from msmp_pde_amd import reductions as R
    gen = torch.Generator(device='cuda').manual_seed(3)
    for rows, cols, group in ((1, 128, 1), (1600, 128, 1), (40000, 304, 38), (77, 25, 25), (5, 4096, 1)):
        x = torch.randn(rows, cols, device='cuda', generator=gen)
        got = R.colsum(x, group)
        ref = x.double().view(rows, cols // group, group).sum((0, 2))
        assert (got.double() - ref).abs().max().item() <= 2e-6 * (x.abs().double().view(rows, cols // group, group).sum((0, 2)).max().item() + 1e-30)
        assert torch.equal(got, R.colsum(x, group))
    a, b = torch.randn(1600, 25, device='cuda', generator=gen, requires_grad=True), torch.randn(1600, 25, device='cuda', generator=gen)
    s = R.sqerr_sum(a, b)
    ref = ((a.detach().double() - b.double()) ** 2).sum()
    assert abs(s.item() - ref.item()) <= 1e-6 * ref.item() and torch.equal(s, R.sqerr_sum(a, b))
    s.backward()
    assert torch.equal(a.grad, 2.0 * (a.detach() - b))
    lin = mp.solvers._lin(36, 128).cuda()
    x = torch.randn(300, 36, device='cuda', generator=gen)
    y = lin(x)
    assert torch.allclose(y, torch.nn.functional.linear(x, lin.weight, lin.bias), rtol=0, atol=1e-6)
    y.square().sum().backward()
    gb = (2.0 * y.detach().double()).sum(0)
    assert (lin.bias.grad.double() - gb).abs().max().item() <= 1e-5 * gb.abs().max().item()


def test_capturable_adamw_matches_the_host_step_count_form(mp):
    """optim.AdamW(capturable=True): step count and learning rate in device memory (msmp_adamw_capturable_f32) give the trajectory
    of the default form, a scheduler's learning-rate change included; state_dict carries the device count back per parameter."""
    torch.manual_seed(3)
    wa = [torch.nn.Parameter(torch.randn(n, device='cuda')) for n in (1, 5000, 12345)]
    wb = [torch.nn.Parameter(w.detach().clone()) for w in wa]
    oa, ob = mp.optim.AdamW(wa, lr=1e-2), mp.optim.AdamW(wb, lr=1e-2, capturable=True)
    sa = torch.optim.lr_scheduler.MultiStepLR(oa, milestones=[3], gamma=0.1)
    sb = torch.optim.lr_scheduler.MultiStepLR(ob, milestones=[3], gamma=0.1)
    gen = torch.Generator(device='cuda').manual_seed(1)
    for it in range(6):
        for pa, pb in zip(wa, wb):
            g = torch.randn(pa.shape, device='cuda', generator=gen)
            pa.grad, pb.grad = g.clone(), g.clone()
        oa.step(); ob.step(); sa.step(); sb.step()
        for pa, pb in zip(wa, wb):
            assert (pa - pb).abs().max().item() <= 1e-6 * max(1.0, pa.abs().max().item()), it
    sd = ob.state_dict()
    assert all(float(sd['state'][i]['step']) == 6.0 for i in range(3)) and '_msmp_dev' not in sd['param_groups'][0]


@pytest.mark.parametrize('name,exp', [('MSMP-PDE', 'E2'), ('MP-PDE', 'E2'), ('Gated', 'WE3'), ('MSMP-PDE2D', 'RPU')])
def test_captured_training_step_follows_the_eager_trajectory(mp, name, exp):
    """train.CapturedTrainStep (forward + loss + backward + AdamW of the reference's batch of 16 as ONE hipGraph launch) against the
    same iterations run eagerly: bit-identical losses and parameters over changing batches, with eager inference forwards of the
    model interleaved between the replays (the pattern under which a library reduction's memset node replayed out of order and
    returned a stale bias gradient in round 2: no library reduction is left in the step) -- and two captured runs agree bit for bit."""
    from msmp_pde_amd import train as T
    from msmp_pde_amd.synthetic import make_case
    bsz = 16
    c = make_case(exp, bsz, seed=9, device='cuda', dtype=torch.float32)

    def batches(k):
        out = []
        for i in range(k):
            steps = [40 + 7 * i + (j % 5) for j in range(bsz)]
            data, labels = c.creator.create_data(c.u_super, steps)
            out.append(c.creator.create_graph(data, labels, c.x, c.variables, steps))
        return out
    gs = batches(7)

    def run(captured):
        torch.manual_seed(11)
        model = mp.MODEL_NAMES[name](c.pde, time_window=TW, eq_variables=c.eqv, hidden_layer=3).cuda()
        opt = mp.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
        sched = torch.optim.lr_scheduler.MultiStepLR(opt, milestones=[6], gamma=0.5)
        losses, preds = [], []
        if captured:
            step = T.CapturedTrainStep(model, opt, gs[0], warmup=3)       # its warm-up iterations are undone: the capture does not train
        for i, g in enumerate(gs):
            if captured:
                losses.append(step(g))
            else:
                opt.zero_grad(set_to_none=True)
                losses.append(T.dp_loss_backward(model, g).detach().clone())
                opt.step()
            sched.step()
            with torch.no_grad():                      # eager work between the replays, reading the freshly updated weights
                preds.append(model(gs[(i + 1) % len(gs)]).clone())
        torch.cuda.synchronize()
        return losses, preds, [p.detach().clone() for p in model.parameters()]

    le, pe, we = run(False)
    lc, pc, wc = run(True)
    lc2, pc2, wc2 = run(True)
    print(name, 'losses', [round(float(l), 5) for l in lc])
    assert all(torch.isfinite(l) for l in lc) and float(lc[-1]) != float(lc[0])
    for i in range(len(le)):
        assert torch.equal(le[i], lc[i]), (i, float(le[i]), float(lc[i]))
        assert torch.equal(pe[i], pc[i]), i
    for a, b, b2 in zip(we, wc, wc2):
        assert torch.equal(a, b) and torch.equal(b, b2)
    assert all(torch.equal(a, b) for a, b in zip(lc, lc2))


@pytest.mark.parametrize('exp', ['E2', 'WE3'])
def test_parity_at_a_trained_operating_point(mp, exp):
    """every full-depth parity number so far is at UNTRAINED default-init weights, where the InstanceNorm stack is
    ill-conditioned and float32 itself misses north_star's 1e-5 on the MSMP-PDE classes.  Here the headline class
    (MP_PDE_SolverLEMLinGated, 6 gated pairs) is trained for 300 captured optimisation steps (train.CapturedTrainStep, the
    reference's batch of 16, experiments/train_helper.py:125-141; AdamW, lr 1e-4 as experiments/train.py) on the synthetic E2 / WE3
    trajectories of a fixed seed, and the HIP forward on the trained weights -- one forward on held-out samples and one rollout step
    on its own prediction (models_gnn.py:1365-1368, common/utils.py:431-471) -- is compared with the float64 oracle and with the
    float32 floor.  The bar asserted is the suite's (max(1e-5, 2 x floor)); whether the plain 1e-5 holds is RECORDED
    (profiles/parity_r04.json: 'meets_plain_1e-5'), not assumed."""
    from types import SimpleNamespace
    from msmp_pde_amd import train as T
    from msmp_pde_amd.synthetic import make_case
    kind = 'MP_PDE_SolverLEMLinGated'
    bsz = 16
    c = make_case(exp, bsz, seed=9, device='cuda', dtype=torch.float32)
    gs = []
    for i in range(8):
        steps = [40 + 7 * i + (j % 5) for j in range(bsz)]
        data, labels = c.creator.create_data(c.u_super, steps)
        gs.append(c.creator.create_graph(data, labels, c.x, c.variables, steps))
    torch.manual_seed(11)
    model = getattr(mp, kind)(c.pde, time_window=TW, eq_variables=c.eqv, hidden_layer=6).cuda()
    opt = mp.optim.AdamW(model.parameters(), lr=1e-4, capturable=True)
    step = T.CapturedTrainStep(model, opt, gs[0], warmup=3)
    losses = [float(step(gs[i % len(gs)])) for i in range(300)]
    torch.cuda.synchronize()
    assert all(np.isfinite(losses)) and np.mean(losses[-16:]) < 0.8 * np.mean(losses[:16]), (losses[:4], losses[-4:])
    model.eval()
    # held-out samples of the same distribution, the reference's evaluation step
    ce = make_case(exp, 8, seed=77, device='cuda', dtype=torch.float64)
    st = [75] * 8
    data, labels = ce.creator.create_data(ce.u_super, st)
    graph = ce.creator.create_graph(data, labels, ce.x, ce.variables, st)
    sd = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    to_np = lambda g: SimpleNamespace(**{k: v.detach().cpu().numpy() for k, v in g.__dict__.items() if torch.is_tensor(v)})
    results = {}
    with torch.no_grad():
        pred = model(graph)
    for tag in ('forward', 'rollout_step'):
        if tag == 'rollout_step':                   # the state update on the HIP prediction, then the next forward: the same input for both sides
            st = [100] * 8
            _, lab = ce.creator.create_data(ce.u_super, st)
            graph = ce.creator.create_next_graph(graph, pred, lab, st)
            with torch.no_grad():
                pred = model(graph)
        g = to_np(graph)
        ref = O.solver_forward(kind, sd, g, ce.pde, TW, ce.eqv, 6)
        floor = fp32_floors(kind, sd, g, ce.pde, TW, ce.eqv, 6)
        out = pred.double().cpu().numpy()
        err, rms = err_stats(out, ref)
        results[tag] = (err, rms)
        assert_parity('parity_at_a_trained_operating_point', f'{kind}/{exp}/trained300/{tag}', out, ref, floor)
        record_parity('trained_operating_point', f'{kind}/{exp}/{tag}', max_abs=err, rms=rms, out_scale=float(np.abs(ref).max()),
                      fp32_floor_max=max(err_stats(v, ref)[0] for v in floor.values()), meets_plain_1e_5=bool(err <= 1e-5),
                      loss_first16=float(np.mean(losses[:16])), loss_last16=float(np.mean(losses[-16:])), optimisation_steps=300)
    print(f'{exp} trained 300 steps (loss {np.mean(losses[:16]):.3f} -> {np.mean(losses[-16:]):.3f}): '
          + ', '.join(f'{k}: max {v[0]:.2e} rms {v[1]:.2e}' for k, v in results.items()))


def _dp_captured_worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK='0')
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import msmp_pde_amd as mp_
    from msmp_pde_amd import dist as D, train as T
    D.init_from_env(backend='gloo')
    case = synthetic_case(mp_, 'E2', bsz=4, seed=6)
    case2 = synthetic_case(mp_, 'E2', bsz=4, seed=8)
    shards = [D.shard_graph(c.graph, rank, world).to('cuda') for c in (case, case2, case)]

    def run(captured):
        torch.manual_seed(5)                                # same weights everywhere
        model = mp_.MP_PDE_SolverLEMLinGated(case.pde, time_window=TW, eq_variables=case.eqv, hidden_layer=2).cuda()
        opt = mp_.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
        step = T.CapturedTrainStep(model, opt, shards[0]) if captured else None
        losses = []
        for g in shards:
            if captured:
                losses.append(step(g))
            else:
                opt.zero_grad(set_to_none=True)
                losses.append(T.dp_loss_backward(model, g).detach().clone())
                opt.step()
        torch.cuda.synchronize()
        return [float(l) for l in losses], [p.detach().cpu().clone() for p in model.parameters()]

    le, we = run(False)
    lc, wc = run(True)
    if rank == 0:
        torch.save({'eager_losses': le, 'captured_losses': lc, 'same_params': all(torch.equal(a, b) for a, b in zip(we, wc)),
                    'max_param_diff': max((a - b).abs().max().item() for a, b in zip(we, wc))}, out_path)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_captured_training_step_under_data_parallelism(mp, tmp_path):
    """VERDICT r03 item 8a / missing #4: CapturedTrainStep with world size 2 (two ranks on the box's one GPU, gloo): forward + S_k,
    the scalar all-reduce, the scaled backward, the flat gradient all-reduce and AdamW as three graph replays with the two
    collectives between them -- the same order and arithmetic as the eager DP step, so losses and parameters of a three-step
    trajectory are the eager DP trajectory's, bit for bit.  Also: building the capture takes no optimisation step."""
    import torch.multiprocessing as tmp
    out_path = str(tmp_path / 'dpc.pt')
    ctx = tmp.get_context('spawn')
    port = 29300 + (os.getpid() % 200)
    procs = [ctx.Process(target=_dp_captured_worker, args=(r, 2, port, out_path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(400)
        assert p.exitcode == 0
    got = torch.load(out_path, weights_only=True)
    print('eager DP losses', got['eager_losses'], 'captured DP losses', got['captured_losses'], 'max param diff', got['max_param_diff'])
    assert got['eager_losses'] == got['captured_losses'] and got['same_params'], got


def test_captured_step_owns_what_it_points_at(mp):
    """(1) building a CapturedTrainStep leaves parameters, Adam moments and the step count where they were; (2) after
    optimizer.load_state_dict (which replaces the group dicts -- the device-side step / learning-rate words -- and the moment
    tensors) a replay would read and write freed memory: the step refuses and asks for a new capture; (3) a capturable group whose
    live parameters carry different step counts is an error, not a silently skipped update."""
    from msmp_pde_amd import train as T
    from msmp_pde_amd.synthetic import make_case
    torch.manual_seed(2)
    c = make_case('E2', 4, seed=9, device='cuda', dtype=torch.float32)
    data, labels = c.creator.create_data(c.u_super, [60] * 4)
    g = c.creator.create_graph(data, labels, c.x, c.variables, [60] * 4)
    model = mp.MP_PDE_SolverGated(c.pde, time_window=TW, eq_variables=c.eqv, hidden_layer=2).cuda()
    before = [p.detach().clone() for p in model.parameters()]
    opt = mp.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
    step = T.CapturedTrainStep(model, opt, g)
    assert all(torch.equal(a, b) for a, b in zip(before, model.parameters()))                      # (1)
    assert all(float(st['exp_avg'].abs().max()) == 0.0 and float(st['exp_avg_sq'].abs().max()) == 0.0 for st in opt.state.values())
    assert int(opt.param_groups[0]['_msmp_dev']['step'].item()) == 0
    l0 = float(step(g))
    assert int(opt.param_groups[0]['_msmp_dev']['step'].item()) == 1 and np.isfinite(l0)
    opt.load_state_dict(opt.state_dict())                                                          # (2)
    with pytest.raises(RuntimeError, match='capture again'):
        step(g)
    step2 = T.CapturedTrainStep(model, opt, g)            # a fresh capture on the loaded state works and continues the count
    assert float(step2(g)) < l0 and int(opt.param_groups[0]['_msmp_dev']['step'].item()) == 2
    # (3)
    opt3 = mp.optim.AdamW(model.parameters(), lr=1e-3, capturable=True)
    opt3.zero_grad(set_to_none=True)
    T.dp_loss_backward(model, g)
    ps = list(model.parameters())
    for i, p in enumerate(ps):
        opt3.state[p] = {'step': torch.tensor(5.0 if i else 0.0), 'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p)}
    with pytest.raises(RuntimeError, match='step count'):
        opt3.step()
