"""Do two (or more) builds of the library give the same bits?  Run on the GPU box:
    python scripts/ab_identity.py libmsmp_pde.so libmsmp_pde_v3.so [--graphs 64]
Each library runs in its own process (MSMP_LIB_PATH) on the same seeded workloads (E2 / WE3 / RPU / MSWG3, MSMP-PDE classes: two
rollout steps) and prints a SHA-256 of the predictions; this parent compares them and, where they differ, prints the largest
difference (so a numerics-preserving but not bit-identical change is told apart from a wrong one)."""
import hashlib, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [('E2', 'MSMP-PDE'), ('WE3', 'MSMP-PDE'), ('RPU', 'MSMP-PDE2D'), ('MSWG3', 'MSMP-PDE2D'), ('E2', 'MP-PDE')]


def child(graphs, out):
    sys.path.insert(0, ROOT)
    import argparse, torch, bench
    import msmp_pde_amd as mp
    dev = torch.device('cuda:0')
    res = {}
    for exp, model in CASES:
        a = bench.parse(['--experiment', exp, '--model', model, '--graphs', str(graphs)])
        w = bench.Workload(a, mp, dev, graphs, seed=1)
        with torch.no_grad():
            w.first(); w.step(); w.step()
        torch.cuda.synchronize()
        p = w.pred.float().cpu().contiguous()
        res[f'{exp}/{model}'] = hashlib.sha256(p.numpy().tobytes()).hexdigest()
        torch.save(p, f'{out}.{exp}.{model}.pt')
    json.dump(res, open(out, 'w'))


if __name__ == '__main__':
    if sys.argv[1] == '--child':
        child(int(sys.argv[2]), sys.argv[3])
        sys.exit(0)
    libs = [a for a in sys.argv[1:] if a.endswith('.so')]
    graphs = int(sys.argv[sys.argv.index('--graphs') + 1]) if '--graphs' in sys.argv else 64
    outs = []
    for i, l in enumerate(libs):
        out = f'/tmp/ab_identity_{i}.json'
        env = dict(os.environ, MSMP_LIB_PATH=os.path.join(ROOT, 'msmp-pde_amd', l))
        r = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(graphs), out], env=env, capture_output=True, text=True)
        if r.returncode:
            print(l, 'FAILED', r.stderr[-1500:]); sys.exit(1)
        outs.append(json.load(open(out)))
    import torch
    ok = True
    for k in outs[0]:
        for i in range(1, len(libs)):
            same = outs[i][k] == outs[0][k]
            msg = 'bit-identical'
            if not same:
                exp, model = k.split('/')
                a = torch.load(f'/tmp/ab_identity_0.json.{exp}.{model}.pt'); b = torch.load(f'/tmp/ab_identity_{i}.json.{exp}.{model}.pt')
                msg = f'DIFFERENT: max |a - b| = {(a - b).abs().max().item():.3e} (max |a| = {a.abs().max().item():.3e}), nan {int(torch.isnan(b).sum())}'
                ok = False
            print(f'{k:22s} {libs[0]} vs {libs[i]}: {msg}', flush=True)
    sys.exit(0 if ok else 2)
