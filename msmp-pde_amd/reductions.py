"""Deterministic reductions for the PyTorch-side pieces of a training iteration (msmp_colsum_f32, msmp_sqerr_sum_f32).

A training iteration of the reference (experiments/train_helper.py:91-141) leaves three kinds of reductions outside the layer /
LEM kernels: the bias gradients of the encoder's and decoder's Linear / Conv1d modules (column sums of dL/dy) and the loss
`MSELoss(reduction='sum')`.  As library reductions they zero a semaphore buffer with hipMemsetAsync, which becomes a memset node
in a captured training step (train.CapturedTrainStep) and was seen to replay out of order; these autograd functions run them on
the library's own two-launch kernels instead (fixed summation order: two runs give the same bits)."""
import torch

from ._lib import lib, check, ptr, current_stream


def _ws(cols, device):
    return torch.empty(lib().msmp_reduce_workspace_bytes(cols), dtype=torch.uint8, device=device)


def colsum(x2d, group=1):
    """[rows, cols] float32 CUDA -> [cols // group]: sums over the rows and over groups of `group` consecutive columns."""
    x2d = x2d.contiguous()
    rows, cols = x2d.shape
    out = torch.empty(cols // group, dtype=torch.float32, device=x2d.device)
    ws = _ws(cols, x2d.device)
    check(lib().msmp_colsum_f32(ptr(x2d), rows, cols, group, ptr(out), ptr(ws), ws.numel(), current_stream()), 'msmp_colsum_f32')
    return out


def _hip_ok(*ts):
    return all(t.is_cuda and t.dtype == torch.float32 for t in ts)


class _BiasAdd(torch.autograd.Function):
    """y[r, j, g] = x[r, j, g] + bias[j]  (x [rows, len(bias) * group]); the bias gradient is msmp_colsum_f32."""

    @staticmethod
    def forward(ctx, x, bias, group):
        ctx.group = group
        rows = x.shape[0]
        return (x.view(rows, bias.shape[0], group) + bias[None, :, None]).view(x.shape)

    @staticmethod
    def backward(ctx, g):
        return g, colsum(g.reshape(g.shape[0], -1), ctx.group), None


def bias_add(x2d, bias, group=1):
    """x2d [rows, len(bias) * group] + bias (each entry over `group` consecutive columns); deterministic bias gradient on the GPU."""
    if _hip_ok(x2d, bias) and torch.is_grad_enabled() and bias.requires_grad:
        return _BiasAdd.apply(x2d.contiguous(), bias, group)
    return (x2d.view(x2d.shape[0], bias.shape[0], group) + bias[None, :, None]).view(x2d.shape)


class _SqErrSum(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, target):
        ctx.save_for_backward(pred, target)
        out = torch.empty(1, dtype=torch.float32, device=pred.device)
        ws = _ws(1, pred.device)
        check(lib().msmp_sqerr_sum_f32(ptr(pred), ptr(target), pred.numel(), ptr(out), ptr(ws), ws.numel(), current_stream()),
              'msmp_sqerr_sum_f32')
        return out.view(())

    @staticmethod
    def backward(ctx, g):
        pred, target = ctx.saved_tensors
        return (pred - target) * (2.0 * g), None


def sqerr_sum(pred, target):
    """sum((pred - target)^2) = MSELoss(reduction='sum') (experiments/train_helper.py:125), differentiable in pred."""
    target = target.to(pred.dtype)
    if _hip_ok(pred, target):
        return _SqErrSum.apply(pred.contiguous(), target.contiguous())
    return ((pred - target) ** 2).sum()
