"""Diagnostic: how much of the full-depth error of the fp16-split path is the dropped lo*lo product?
Runs the per-layer error profile of scripts/diag_error.py for the library MSMP_LIB_PATH points at (default build, or the
`MSMP_LOLO=1|2 python msmp-pde_amd/build.py` diagnostic builds) and prints one JSON line per (class, experiment):
per-layer max / rms error of the hidden state against the float64 oracle for the HIP path and for a float32 evaluation of
the oracle, and the same for the network output.  scripts/diag_lolo.sh runs it for the three builds.
    python scripts/diag_lolo.py [kind exp]..."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
import msmp_pde_amd as mp
from helpers import layer_error_profile

TW = 25
if os.environ.get('MSMP_DIAG_SPLIT') is not None:
    mp.lib().msmp_tune(b'split', int(os.environ['MSMP_DIAG_SPLIT']))


def profile(kind, exp):
    rec = layer_error_profile(mp, kind, exp)
    rec['lib'] = os.path.basename(os.environ.get('MSMP_LIB_PATH', 'libmsmp_pde.so')) + ('' if os.environ.get('MSMP_DIAG_SPLIT') is None else ' split=' + os.environ['MSMP_DIAG_SPLIT'])
    return rec


if __name__ == '__main__':
    args = sys.argv[1:] or ['MP_PDE_SolverLEMLinGated', 'E2', 'MP_PDE_SolverLEMLinGated', 'WE3']
    for kind, exp in zip(args[0::2], args[1::2]):
        r = profile(kind, exp)
        print(json.dumps(r), flush=True)
        o = r['out']
        print(f"# {r['lib']} {kind}/{exp}: out max {o['hip'][0]:.3e} ({o['hip'][0] / o['f32'][0]:.2f} x f32) rms {o['hip'][1]:.3e} "
              f"({o['hip'][1] / o['f32'][1]:.2f} x f32); per-layer rms ratio hip/f32: "
              + ' '.join(f"{l['hip'][1] / l['f32'][1]:.2f}" for l in r['layers'])
              + ' | fresh rms ' + ' '.join(f"{l['hip_fresh'][1]:.1e}" for l in r['layers']), flush=True)
