"""Does running the batch as two half-batches on two streams (hipGraph replays of the forward, private workspaces) overlap
kernels with different bottlenecks?  python scripts/two_stream_halves.py [graphs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import msmp_pde_amd as mp
args = bench.parse(['--no-cpu-baseline', '--no-extras'])
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = torch.device('cuda:0')
with torch.no_grad():
    whole = bench.Workload(args, mp, dev, n_graphs, seed=1)
    whole.first()
    for _ in range(5): whole.step()
    halves = [bench.Workload(args, mp, dev, n_graphs // 2, seed=2 + i) for i in range(2)]
    for h in halves:
        h.model = whole.model
        h.first()
        for _ in range(3): h.step()
    def timed(fn, n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3
    n = 60
    # pre-heat
    timed(whole.step, 100)
    t_whole = timed(whole.step, n)
    t_seq = timed(lambda: [h.step() for h in halves], n)
    caps = [h.model.capture(h.graph) for h in halves]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def step_conc():
        for h, cap, s in zip(halves, caps, streams):
            with torch.cuda.stream(s):
                step = 75 + 25 * (h.i % 7); h.i += 1
                same = [step] * h.bsz
                _, lab = h.case.creator.create_data(h.case.u_super, same)
                g = h.case.creator.create_next_graph(h.graph, h.pred, lab, same)
                h.pred = cap(g)
    for s in streams: s.wait_stream(torch.cuda.current_stream())
    timed(step_conc, 10)
    t_conc = timed(step_conc, n)
    print(f'{n_graphs} graphs: one batch {t_whole:.4f} ms per step; two halves one after the other {t_seq:.4f} ms; two halves on two streams (graph replays) {t_conc:.4f} ms')
