"""Stand-in for the absent lem_cuda extension: import succeeds, calls raise (its source is not in the reference)."""


def forward(*a, **k):
    raise NotImplementedError('lem_cuda is absent from the reference tree')


def backward(*a, **k):
    raise NotImplementedError('lem_cuda is absent from the reference tree')
