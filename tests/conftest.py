"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path.

`-m "not gpu"`: oracle vs golden vectors, host logic, C-ABI symbol checks, gloo multi-process tests.
`-m gpu`      : parity tests proper; they call the HIP kernels through the C-ABI on cuda:0.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


# ---------------------------------------------------------------------------------------------------------------------
# Guard bands for the GPU tests.  GPU AddressSanitizer is not available on the test pool, so every `-m gpu` test runs with
# torch.empty / torch.empty_like (what the host layer and the tests allocate kernel outputs and workspaces with) replaced by
# versions that put 1 KB of sentinel values on both sides of the tensor; at the end of the test every band must be intact.
# A kernel that stores past the end of a ragged last tile (or before the start) of ANY output fails the test that ran it.
# MSMP_NO_GUARD=1 switches it off.
# ---------------------------------------------------------------------------------------------------------------------
_GUARD = 256                      # elements on each side (>= 256 bytes: keeps the 256-byte alignment of the tensor)
_GUARD_MAX_ELEMS = 1 << 28        # larger tensors are passed through (the full-size tests)


@pytest.fixture(autouse=True)
def _guard_bands(request):
    if 'gpu' not in request.keywords or os.environ.get('MSMP_NO_GUARD'):
        yield
        return
    import math
    import torch
    if not torch.cuda.is_available():
        yield
        return
    sentinel = {torch.float32: 12345.678, torch.int32: 0x5A5A5A5A, torch.uint8: 0x5A, torch.int64: 0x5A5A5A5A5A5A}
    real_empty, real_empty_like = torch.empty, torch.empty_like
    tracked = []

    def guarded(shape, dtype, device):
        n = math.prod(shape)
        base = real_empty(n + 2 * _GUARD, dtype=dtype, device=device)
        base[:_GUARD] = sentinel[dtype]
        base[_GUARD + n:] = sentinel[dtype]
        tracked.append((base, n))
        return base[_GUARD:_GUARD + n].view(shape)

    def eligible(shape, dtype, device):
        return (device is not None and torch.device(device).type == 'cuda' and dtype in sentinel
                and all(isinstance(s, int) for s in shape) and 0 < math.prod(shape) <= _GUARD_MAX_ELEMS)

    def empty(*size, **kw):
        if set(kw) <= {'dtype', 'device'}:
            shape = size[0] if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
            try:
                shape = tuple(int(s) for s in shape)
            except (TypeError, ValueError):
                return real_empty(*size, **kw)
            dtype = kw.get('dtype') or torch.get_default_dtype()
            if eligible(shape, dtype, kw.get('device')):
                return guarded(shape, dtype, kw['device'])
        return real_empty(*size, **kw)

    def empty_like(x, **kw):
        if not kw and x.is_contiguous() and eligible(tuple(x.shape), x.dtype, x.device):
            return guarded(tuple(x.shape), x.dtype, x.device)
        return real_empty_like(x, **kw)

    torch.empty, torch.empty_like = empty, empty_like
    try:
        yield
    finally:
        torch.empty, torch.empty_like = real_empty, real_empty_like
    torch.cuda.synchronize()
    bad = None
    for base, n in tracked:
        s = sentinel[base.dtype]
        cnt = (base[:_GUARD] != s).sum() + (base[_GUARD + n:] != s).sum()
        bad = cnt if bad is None else bad + cnt
    n_bad = int(bad.item()) if bad is not None else 0
    assert n_bad == 0, f'{n_bad} guard-band elements around kernel outputs / workspaces were overwritten (out-of-bounds stores)'


@pytest.fixture(autouse=True)
def _range_status_stays_clear(request):
    """Every GPU test must leave the sticky range status of the fp16-split path (msmp_last_status) at 0: no test input may
    saturate or overflow silently.  Tests that trip it on purpose clear it themselves (last_status(reset=True))."""
    if 'gpu' not in request.keywords:
        yield
        return
    import torch
    if not torch.cuda.is_available():
        yield
        return
    import msmp_pde_amd as mp
    mp.last_status(reset=True)
    yield
    torch.cuda.synchronize()
    flags = mp.last_status(reset=True)
    assert flags == 0, f'range status {flags} left behind: a kernel saturated a node row or met a non-finite value'
