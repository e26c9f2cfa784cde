"""Power cap evidence: the same rollout step with an idle gap (one spinning thread: torch.cuda._sleep) of G x the step time enqueued behind
every step.  If the package power cap is what sets the clock, the kernels of the duty-cycled run are FASTER (the energy not drawn during
the gap is available to them).  Run under rocprofv3 --kernel-trace --stats once per gap:
    rocprofv3 --kernel-trace --stats -d out -- python3 scripts/duty_cycle.py <gap fraction>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import msmp_pde_amd as mp
gap = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
a = bench.parse(['--graphs', '2048'])
w = bench.Workload(a, mp, torch.device('cuda:0'), 2048, seed=1)
with torch.no_grad():
    w.first()
    for _ in range(3): w.step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): w.step()
    torch.cuda.synchronize(); step_s = (time.perf_counter() - t0) / 20
    cycles = int(gap * step_s * 2.28e9)         # _sleep spins on the shader clock counter (measured: 645 k cycles = 283 us)
    t_end = time.perf_counter() + 6.0           # >= 6 s so that the power manager settles
    n = 0
    while time.perf_counter() < t_end:
        for _ in range(10):
            w.step()
            if cycles: torch.cuda._sleep(cycles)
        torch.cuda.synchronize(); n += 10
print(f'gap {gap}: plain step {step_s * 1e3:.3f} ms, {n} duty-cycled steps')
