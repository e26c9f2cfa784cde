"""Stand-ins for torch_geometric.nn.{MessagePassing, InstanceNorm} (documented semantics).

MessagePassing.propagate, flow='source_to_target':
  *_j <- tensor[edge_index[0]]  (source),  *_i <- tensor[edge_index[1]]  (target)
  aggregate 'mean' over edge_index[1] into N rows: sum / clamp(count, 1)
  update(aggr_out, **kwargs restricted to update()'s signature)
InstanceNorm(C) (affine=False, track_running_stats=False): per graph, per channel
  (x - mean) / sqrt(biased_var + eps), eps = 1e-5.
"""
import inspect
import torch
from torch import nn


class MessagePassing(nn.Module):
    def __init__(self, aggr='add', flow='source_to_target', node_dim=-2):
        super().__init__()
        assert flow == 'source_to_target'
        assert node_dim in (-2, 0)
        self.aggr = aggr
        self._msg_params = [p for p in inspect.signature(self.message).parameters]
        self._upd_params = [p for p in inspect.signature(self.update).parameters][1:]

    def propagate(self, edge_index, size=None, **kwargs):
        src, dst = edge_index[0], edge_index[1]
        n = None
        for v in kwargs.values():
            if torch.is_tensor(v):
                n = v.size(0)
                break
        margs = {}
        for name in self._msg_params:
            if name.endswith('_i'):
                margs[name] = kwargs[name[:-2]].index_select(0, dst)
            elif name.endswith('_j'):
                margs[name] = kwargs[name[:-2]].index_select(0, src)
            else:
                margs[name] = kwargs[name]
        msg = self.message(**margs)
        out = torch.zeros((n,) + tuple(msg.shape[1:]), dtype=msg.dtype, device=msg.device)
        out.index_add_(0, dst, msg)
        if self.aggr == 'mean':
            cnt = torch.zeros(n, dtype=msg.dtype, device=msg.device)
            cnt.index_add_(0, dst, torch.ones(dst.numel(), dtype=msg.dtype, device=msg.device))
            out = out / cnt.clamp(min=1).unsqueeze(-1)
        elif self.aggr not in ('add', 'sum'):
            raise NotImplementedError(self.aggr)
        uargs = {name: kwargs[name] for name in self._upd_params}
        return self.update(out, **uargs)

    def message(self, x_j):
        return x_j

    def update(self, inputs):
        return inputs


class InstanceNorm(nn.Module):
    def __init__(self, in_channels, eps=1e-5, momentum=0.1, affine=False, track_running_stats=False):
        super().__init__()
        assert not affine and not track_running_stats
        self.eps = eps

    def forward(self, x, batch=None):
        if batch is None:
            batch = torch.zeros(x.size(0), dtype=torch.long, device=x.device)
        b = int(batch.max()) + 1
        ones = torch.ones(x.size(0), dtype=x.dtype, device=x.device)
        cnt = torch.zeros(b, dtype=x.dtype, device=x.device).index_add_(0, batch, ones).clamp_(min=1).view(-1, 1)
        mean = torch.zeros(b, x.size(1), dtype=x.dtype, device=x.device).index_add_(0, batch, x) / cnt
        xc = x - mean.index_select(0, batch)
        var = torch.zeros(b, x.size(1), dtype=x.dtype, device=x.device).index_add_(0, batch, xc * xc) / cnt
        return xc / (var + self.eps).sqrt().index_select(0, batch)


def _absent(name):
    class _Absent(nn.Module):
        def __init__(self, *a, **k):
            raise NotImplementedError(f'{name}: not provided by the stand-in (out of scope)')
    _Absent.__name__ = name
    return _Absent


def global_mean_pool(*a, **k):
    raise NotImplementedError


def avg_pool_x(*a, **k):
    raise NotImplementedError


BatchNorm = _absent('BatchNorm')
GCNConv = _absent('GCNConv')
GATConv = _absent('GATConv')
SAGEConv = _absent('SAGEConv')
TransformerConv = _absent('TransformerConv')
RGATConv = _absent('RGATConv')
