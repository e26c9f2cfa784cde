"""Stand-in for torch_scatter.scatter (imported by the reference; not reached on the in-scope path)."""
import torch


def scatter(src, index, dim=0, out=None, dim_size=None, reduce='sum'):
    assert dim == 0
    n = int(index.max()) + 1 if dim_size is None else dim_size
    res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device).index_add_(0, index, src)
    if reduce in ('sum', 'add'):
        return res
    if reduce == 'mean':
        cnt = torch.zeros(n, dtype=src.dtype, device=src.device).index_add_(
            0, index, torch.ones(index.numel(), dtype=src.dtype, device=src.device)).clamp_(min=1)
        return res / cnt.view((-1,) + (1,) * (src.dim() - 1))
    raise NotImplementedError(reduce)
