"""Stand-ins for torch_cluster.radius_graph / knn_graph (documented semantics, brute force).

radius_graph(x, r, batch, loop=False, max_num_neighbors=32): pair (j -> i) kept iff same batch,
  i != j and squared distance < r*r (strict), at most max_num_neighbors per target i (lowest j
  first); returns stack([source j, target i]) grouped by ascending target, sources ascending.
knn_graph(x, k, batch, loop=False): for each target i its k nearest same-batch nodes j != i,
  ascending distance, ties -> lower index; stack([source j, target i]) grouped by ascending target.
"""
import torch


def _as2d(x):
    return x.view(-1, 1) if x.dim() == 1 else x


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow='source_to_target', num_workers=1):
    assert flow == 'source_to_target'
    x = _as2d(x)
    n = x.size(0)
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long)
    src, dst = [], []
    r2 = r * r
    for b in torch.unique(batch).tolist():
        idx = (batch == b).nonzero().view(-1)
        xb = x[idx]
        d2 = ((xb[:, None, :] - xb[None, :, :]) ** 2).sum(-1)
        m = d2 < r2
        if not loop:
            m = m & ~torch.eye(len(idx), dtype=torch.bool)
        for ti in range(len(idx)):
            js = m[ti].nonzero().view(-1)[:max_num_neighbors]
            src.append(idx[js])
            dst.append(idx[ti].repeat(len(js)))
    return torch.stack([torch.cat(src), torch.cat(dst)], 0)


def knn_graph(x, k, batch=None, loop=False, flow='source_to_target', cosine=False, num_workers=1):
    assert flow == 'source_to_target' and not cosine
    x = _as2d(x)
    n = x.size(0)
    if batch is None:
        batch = torch.zeros(n, dtype=torch.long)
    src, dst = [], []
    for b in torch.unique(batch).tolist():
        idx = (batch == b).nonzero().view(-1)
        xb = x[idx]
        d2 = ((xb[:, None, :] - xb[None, :, :]) ** 2).sum(-1)
        if not loop:
            d2 = d2 + torch.diag(torch.full((len(idx),), float('inf'), dtype=d2.dtype))
        order = torch.sort(d2, dim=1, stable=True).indices[:, :k]
        for ti in range(len(idx)):
            js = order[ti]
            js = js[torch.isfinite(d2[ti, js])]
            src.append(idx[js])
            dst.append(idx[ti].repeat(len(js)))
    return torch.stack([torch.cat(src), torch.cat(dst)], 0)
