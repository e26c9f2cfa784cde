"""Training step of the reference under graph-sharded data parallelism.

Reference (single device): experiments/train_helper.py:91-141 -- pushforward unrolling under no_grad,
then `loss = sqrt(MSELoss(reduction='sum')(pred, graph.y)); loss.backward(); optimizer.step()`.
The loss is sqrt(S) with S = sum over ALL nodes of the batch, which is not a mean of per-shard losses, so
exact data parallelism is:  (1) all-reduce(SUM) the scalar S_k of every rank;  (2) backward S_k scaled by
1/(2 sqrt(S));  (3) all-reduce(SUM) the gradients (one flat ~5 MB buffer: 1.25-1.34 M fp32 parameters),
RCCL over xGMI when the backend is nccl.  Parameters are replicated; every rank must draw the same
`unrolled_graphs` and its own slice of `random_steps`."""
import torch
import torch.distributed as dist


def _world():
    """Number of ranks; 0 when there is no process group at all (an initialised ONE-rank group still runs its collectives:
    dist.init_from_env(force_group=True) exercises the RCCL path on a single GPU)."""
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 0


def allreduce_gradients(params):
    """SUM-all-reduce the gradients of `params` as one flat buffer (a single collective per step)."""
    params = [p for p in params if p.grad is not None]
    if _world() == 0 or not params:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


def dp_loss_backward(model, graph):
    """Forward + backward of the reference's loss on this rank's shard; gradients are left all-reduced
    (identical on every rank and equal to the single-device gradient).  Returns the global loss sqrt(S)."""
    pred = model(graph)
    s_local = ((pred - graph.y.to(pred.dtype)) ** 2).sum()
    s_total = s_local.detach().clone()
    if _world() > 0:
        dist.all_reduce(s_total, op=dist.ReduceOp.SUM)
    loss = torch.sqrt(s_total)
    (s_local / (2.0 * loss)).backward()
    allreduce_gradients(list(model.parameters()))
    return loss


def training_step(model, creator, u_super, x, variables, random_steps, unrolled_graphs, optimizer):
    """One iteration of training_loop (experiments/train_helper.py:91-141) on this rank's samples."""
    optimizer.zero_grad()
    steps = list(random_steps)
    data, labels = creator.create_data(u_super, steps)
    graph = creator.create_graph(data, labels, x, variables, steps)
    with torch.no_grad():                                   # pushforward trick, :106-112
        for _ in range(unrolled_graphs):
            steps = [s + creator.tw for s in steps]
            _, labels = creator.create_data(u_super, steps)
            pred = model(graph)
            graph = creator.create_next_graph(graph, pred, labels, steps)
    loss = dp_loss_backward(model, graph)
    optimizer.step()
    lem = getattr(model, 'embedding_lem', None)            # the Save variants start every sample from fresh LEM states (:144-145)
    if hasattr(lem, 'reset_states'):
        lem.reset_states()
    return loss
