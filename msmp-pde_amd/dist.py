"""Data parallelism over graphs.  Graphs of a batch are independent
(radius/knn never connect different `batch` ids, InstanceNorm is per graph, the decoder is per
node), so rank r evaluates the contiguous block of graphs shard_range(B, r, W); the rollout needs
no collective.  One process per GPU; torch.distributed (backend "nccl" = RCCL) is only used by
the callers for barriers and for gathering scalars."""
import torch

from .graph import Data


def shard_range(n_graphs, rank, world):
    """Contiguous, balanced block [g0, g1) of graphs for `rank` (first n_graphs % world ranks get one more)."""
    q, r = divmod(n_graphs, world)
    g0 = rank * q + min(rank, r)
    return g0, g0 + q + (1 if rank < r else 0)


def shard_graph(graph, rank, world):
    """The sub-batch of `graph` owned by `rank`: node rows sliced, edge_index filtered and rebased,
    batch ids rebased to start at 0.  Exact: per-graph results do not depend on the other graphs."""
    batch = graph.batch
    n_graphs = int(batch[-1].item()) + 1
    g0, g1 = shard_range(n_graphs, rank, world)
    node_mask = (batch >= g0) & (batch < g1)
    idx = node_mask.nonzero().view(-1)
    n0 = int(idx[0].item()) if idx.numel() else 0
    n1 = n0 + idx.numel()
    ei = graph.edge_index
    emask = (ei[1] >= n0) & (ei[1] < n1)
    out = Data(x=graph.x[n0:n1], edge_index=ei[:, emask] - n0)
    n = batch.shape[0]
    for k, v in graph.__dict__.items():
        if k in ('x', 'edge_index') or k.startswith('_'):
            continue
        if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == n:
            v = v[n0:n1]
            if k == 'batch':
                v = v - g0
        setattr(out, k, v)
    return out


def init_from_env(backend=None, force_group=False):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun), one process per GPU.
    Returns (rank, world, local_rank).  Single-process runs (WORLD_SIZE unset or 1) skip torch.distributed unless `force_group`
    (or MSMP_FORCE_DIST=1 in the environment) asks for a ONE-rank group: every collective of the data-parallel path then really
    goes through the backend (nccl = RCCL), which is how the RCCL code path is exercised on a single-GPU box."""
    import os
    import torch.distributed as dist
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    force_group = force_group or os.environ.get('MSMP_FORCE_DIST') == '1'
    if (world > 1 or force_group) and not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if torch.cuda.is_available():
            torch.cuda.set_device(local % torch.cuda.device_count())
        if world == 1:
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            os.environ.setdefault('MASTER_PORT', str(29400 + os.getpid() % 500))
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def reduce_scalar(value, op='max', device=None):
    """All-reduce one float over the ranks (max of the per-rank step time; sum of per-rank unit counts)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    if device is None:
        device = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX if op == 'max' else dist.ReduceOp.SUM)
    return float(t.item())
