import os, sys, time
sys.path.insert(0, '/root/repo')
import torch, bench
import msmp_pde_amd as mp
args = bench.parse(['--no-cpu-baseline', '--no-extras', '--model', 'MP-PDE'])
for n_graphs in (32, 128):
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
        run(20, wl.model); eager = run(400, wl.model)
        cap = wl.model.capture(wl.graph); run(20, cap); replay = run(400, cap)
        t0 = time.perf_counter()
        for _ in range(100):
            same = [75] * wl.bsz
            _, lab = wl.case.creator.create_data(wl.case.u_super, same)
            g = wl.case.creator.create_next_graph(wl.graph, wl.pred, lab, same)
        torch.cuda.synchronize(); upd = (time.perf_counter() - t0) / 100 * 1e3
        print(f'MP-PDE {n_graphs} graphs: eager {eager:.4f} ms/step, captured forward {replay:.4f}, state update alone {upd:.4f}', flush=True)
