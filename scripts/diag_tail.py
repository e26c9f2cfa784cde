"""Where do two builds of the node tail differ?  python scripts/diag_tail.py libA.so libB.so   (one process per library, same seeded input:
one gated launch over 3 graphs of 100 nodes; prints max |a - b| per block of 32 rows and per channel residue mod 4 = channel tile T)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if sys.argv[1] == '--child':
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import numpy as np, torch
    import msmp_pde_amd as mp
    from msmp_pde_amd._lib import check, ptr, current_stream
    import test_gpu_kernels as T
    L = mp.lib(); rng = np.random.default_rng(7); sizes = [100, 100, 100]; n = 300; nv = 2
    h, _, _, var = T.layer_inputs(rng, n, 25, nv)
    aggs = [rng.standard_normal((n, 128)).astype(np.float32) for _ in range(2)]
    blobs = [T.pack(mp, T.rand_layer_sd(rng, 25, nv, scale=2.0), 25, nv) for _ in range(2)]
    gptr = T.dev(np.array([0, 100, 200, 300], dtype=np.int32))
    dh, dvar, dagg = T.dev(h), T.dev(var), [T.dev(a) for a in aggs]
    out = torch.full((n, 128), float('nan'), device='cuda')
    check(L.msmp_node_tail_f32(ptr(dh), ptr(dagg[0]), ptr(dagg[1]), ptr(dvar), ptr(gptr), n, 3, 100, nv, ptr(blobs[0]), ptr(blobs[1]), 1, 1e-5, ptr(out),
                               current_stream()), 'tail')
    torch.cuda.synchronize(); torch.save(out.cpu(), sys.argv[2]); sys.exit(0)
import torch
outs = []
for lib in sys.argv[1:3]:
    f = f'/tmp/diag_tail_{lib}.pt'
    subprocess.run([sys.executable, __file__, '--child', f], env=dict(os.environ, MSMP_LIB_PATH=os.path.join(ROOT, 'msmp-pde_amd', lib)), check=True)
    outs.append(torch.load(f))
d = (outs[0] - outs[1]).abs()
print('max diff', d.max().item(), 'nan', torch.isnan(outs[1]).sum().item())
for g in range(3):
    print('graph', g, 'row blocks of 32:', [f'{d[100 * g + 32 * b:100 * g + min(32 * b + 32, 100)].max().item():.1e}' for b in range(4)])
print('channel tile T (channel mod 4):', [f'{d[:, t::4].max().item():.1e}' for t in range(4)])
print('channel blocks of 32:', [f'{d[:, 32 * b:32 * b + 32].max().item():.1e}' for b in range(4)])
print('rows of graph 0 with diff > 1e-4:', (d[:100].max(dim=1).values > 1e-4).nonzero().flatten().tolist()[:40])
