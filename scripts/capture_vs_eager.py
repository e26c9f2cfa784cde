"""Rollout step time, eager forward vs Solver.capture() replay (hipGraph), per batch size:  python scripts/capture_vs_eager.py [graphs ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import msmp_pde_amd as mp
args = bench.parse(['--no-cpu-baseline', '--no-extras'])
for n_graphs in [int(a) for a in sys.argv[1:]] or [32, 256, 2048]:
    wl = bench.Workload(args, mp, torch.device('cuda:0'), n_graphs, seed=1)
    with torch.no_grad():
        wl.first()
        for _ in range(5): wl.step()
        def run(n, fwd):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n):
                step = 75 + 25 * (wl.i % 7); wl.i += 1
                same = [step] * wl.bsz
                _, lab = wl.case.creator.create_data(wl.case.u_super, same)
                g = wl.case.creator.create_next_graph(wl.graph, wl.pred, lab, same)
                wl.pred = fwd(g)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        n = 400 if n_graphs <= 256 else 100
        run(20, wl.model)
        eager = run(n, wl.model)
        cap = wl.model.capture(wl.graph)
        run(20, cap)
        replay = run(n, cap)
        # host time of one eager step issue (no sync inside): how close the CPU is to being the bound
        t0 = time.perf_counter(); 
        for _ in range(50):
            step = 75; same = [step] * wl.bsz
            _, lab = wl.case.creator.create_data(wl.case.u_super, same)
            g = wl.case.creator.create_next_graph(wl.graph, wl.pred, lab, same)
            wl.pred = wl.model(g)
        issue = (time.perf_counter() - t0) / 50 * 1e3
        torch.cuda.synchronize()
        print(f'{n_graphs:5d} graphs: eager {eager:.4f} ms per step, captured forward {replay:.4f} ms, host issue time of an eager step {issue:.4f} ms', flush=True)
