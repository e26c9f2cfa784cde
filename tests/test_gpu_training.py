"""GPU tests of the training path: autograd through the HIP forward (recompute backward), and exact
graph-sharded data parallelism of the reference's sqrt-of-sum loss with the gradient all-reduce, on two
ranks (gloo, both ranks on the one GPU of the box; with the nccl backend the same code runs over RCCL)."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import msmp_oracle_torch as OT
from helpers import synthetic_case

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TW = 25


@pytest.fixture(scope='module')
def mp():
    import msmp_pde_amd
    assert torch.cuda.is_available()
    return msmp_pde_amd


@pytest.mark.parametrize('kind,exp', [('MP_PDE_Solver', 'E2'), ('MP_PDE_SolverGated', 'E2'), ('MP_PDE_Solver2DGated', 'RPU'),
                                      ('MP_PDE_SolverLEMLinGated', 'E2'), ('MP_PDE_Solver2DLEMLinGated', 'RPU')])
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
    for name, p in model.named_parameters():
        g, r = p.grad.double().cpu(), sd64[name].grad
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
        rel = (dp['grads'][k] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-2 * scale)
        assert rel < 1e-4, (k, rel)     # fp32 summation-order noise only (shard sums are added in another order)


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
