"""Deterministic parameter values for the full-depth golden fixtures.

A depth-6 state_dict of a gated solver is 1.25 M parameters (5 MB per fixture).  Instead of committing the values, the
generator (gen_golden.py, which runs the REFERENCE's classes) and the tests (which run the oracle / the HIP path) both
fill the state_dict from this function: names and shapes come from the model's own state_dict (reference and drop-in
classes have identical ones, tests/test_host_cpu.py), values from numpy's PCG64 stream, which is stable across numpy
versions and platforms.  Distribution = torch's default nn.Linear / nn.Conv1d initialisation (uniform in
+-1/sqrt(fan_in)), rounded to float32 so the float64 reference and the float32 kernels see identical numbers.
"""
import numpy as np


def seeded_state_dict(shapes, seed):
    """shapes: ordered {name: shape} (a state_dict's names and shapes).  Returns {name: float32 ndarray}."""
    rng = np.random.default_rng(seed)
    fan = {}
    for name, shape in shapes.items():          # a bias shares the fan-in of the weight that precedes it
        if name.endswith('weight') and len(shape) >= 2:
            fan[name[:-len('weight')]] = int(np.prod(shape[1:]))
    out = {}
    for name, shape in shapes.items():
        prefix = name[:name.rfind('.') + 1]
        fan_in = fan.get(prefix, None)
        if fan_in is None:
            fan_in = int(shape[-1]) if len(shape) else 1
        bound = 1.0 / np.sqrt(max(fan_in, 1))
        out[name] = rng.uniform(-bound, bound, size=tuple(shape)).astype(np.float32)
    return out
