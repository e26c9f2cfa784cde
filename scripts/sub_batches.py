"""The 2048-graph rollout step as S sub-batches stepping concurrently on S streams (eager launches, per-stream workspaces):
does overlapping kernels of different sub-batches (tails of one kernel's grid beside the head of the next) pay?
    python scripts/sub_batches.py [graphs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import msmp_pde_amd as mp
args = bench.parse(['--no-cpu-baseline', '--no-extras'])
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device('cuda:0')

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    whole = bench.Workload(args, mp, dev, n_graphs, seed=1)
    whole.first()
    timed(whole.step, 150)                     # pre-heat
    res = {1: timed(whole.step, 60)}
    for S in (2, 3, 4):
        parts = [bench.Workload(args, mp, dev, n_graphs // S, seed=10 * S + i) for i in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)]
        for p, s in zip(parts, streams):
            p.model = whole.model
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                p.first()
        def step():
            for p, s in zip(parts, streams):
                with torch.cuda.stream(s):
                    p.step()
        timed(step, 20)
        res[S] = timed(step, 60)
        res[-S] = timed(lambda: [p.step() for p in parts], 30)       # the same sub-batches one after the other on one stream
    t1 = timed(whole.step, 60)
with torch.no_grad():
    # S sub-batches, each a hipGraph replay of its forward on its own stream (the host issues ~6 launches per sub-batch and step instead of ~25)
    for S in (1, 2, 3, 4, 6):
        parts = [bench.Workload(args, mp, dev, n_graphs // S, seed=50 + 7 * S + i) for i in range(S)]
        streams = [torch.cuda.Stream() for _ in range(S)]
        caps = []
        for p, s in zip(parts, streams):
            p.model = whole.model
            p.first()
            caps.append(p.model.capture(p.graph))
        torch.cuda.synchronize()
        def step_cap():
            for p, cap, s in zip(parts, caps, streams):
                with torch.cuda.stream(s):
                    step = 75 + 25 * (p.i % 7)
                    src = p.x0 if p.i % 7 == 0 else p.pred
                    p.i += 1
                    same = [step] * p.bsz
                    _, lab = p.case.creator.create_data(p.case.u_super, same)
                    g = p.case.creator.create_next_graph(p.graph, src, lab, same)
                    p.pred = cap(g)
        for s in streams: s.wait_stream(torch.cuda.current_stream())
        timed(step_cap, 20)
        t_cap = timed(step_cap, 100)
        print(f'{n_graphs} graphs: {S} sub-batches on {S} streams, captured forwards: {t_cap:.3f} ms per step', flush=True)
        del caps, parts
with torch.no_grad():
    whole.model.sub_batches = 2
    timed(whole.step, 10)
    t_mode = timed(whole.step, 60)
    whole.model.sub_batches = 1
print(f'{n_graphs} graphs: Solver.sub_batches = 2 (the product mode, one Workload): {t_mode:.3f} ms per step')
print(f'{n_graphs} graphs, ms per rollout step: one batch {res[1]:.3f} (again at the end: {t1:.3f}); '
      + '; '.join(f'{S} sub-batches on {S} streams {res[S]:.3f} (one stream: {res[-S]:.3f})' for S in (2, 3, 4)))
