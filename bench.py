#!/usr/bin/env python3
"""bench.py -- rollout-steps/sec of the MSMP-PDE message-passing rollout step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--graphs 2048] [--model MSMP-PDE]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): E2 (Burgers-type CE, nx=100, time_window=25, radius graph n=3,
588 edges/graph), model MSMP-PDE = MP_PDE_SolverLEMLinGated (6 gated layer pairs), 2048 graphs per GPU,
random-init weights, synthetic trajectories resident in HBM.  One "step" = one rollout step of the
reference's unrolled evaluation (experiments/train_helper.py:255-261): create_next_graph (state update)
+ model(graph) under no_grad, on the rank's whole batch.  Multi-GPU: graphs are independent, so every
rank runs its own 2048-graph batch with no data-path collective (weak scaling); value = total rollout
steps of all ranks / max-over-ranks time.

Extra objects on the JSON line:
  roofline      dominant kernel (edge-message MLP): algorithmic FLOPs of the dense formulation
                (2*E*K_msg*H + 2*E*H*H per launch) / mean launch time measured with
                HIP events on the launch stream inside the timed region, vs the fp32 MFMA peak.
  scatter_hbm   row L2 standalone: achieved HBM GB/s of the CSR segmented-mean kernel on a message tensor of the workload's size
                (outside the timed region; the default layer path fuses the mean into the message kernel).
  cpu_baseline  the CPU oracle (torch-CPU float64 edition, kind "port") timed on this host on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_FP16_MFMA_TFLOPS = 2500.0    # same guide, "Peak BF16/FP16 MFMA" (dense)
H = 128


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--graphs', type=int, default=2048, help='graphs per GPU')
    ap.add_argument('--model', default='MSMP-PDE', help='MSMP-PDE | Gated | MP-PDE')
    ap.add_argument('--experiment', default='E2')
    ap.add_argument('--neighbors', type=int, default=3, help='n of the graph builder (radius n*dx / knn k); 8, 16 = the MSWG3 edge-count stress of SURVEY 8(d)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-sample-graphs', type=int, default=128)
    ap.add_argument('--cpu-sample-steps', type=int, default=3)
    ap.add_argument('--dist-backend', default=None, help='torch.distributed backend (default nccl = RCCL); gloo lets two ranks share one GPU for testing')
    ap.add_argument('--tune', action='append', default=[], metavar='KEY=VALUE', help='msmp_tune override for kernel A/B runs (e.g. lem=1)')
    ap.add_argument('--fp32-mfma', action='store_true', help='use the fp32-MFMA kernels instead of the fp16-split matrix path')
    ap.add_argument('--time-all-kernels', action='store_true', help='event-time every kernel family, not only the dominant one')
    return ap.parse_args()


def cpu_baseline(args, kind, eqv):
    """The oracle (torch-CPU float64 edition, `kind: port`: the reference itself is PyTorch on the CPU)
    on a bounded sample of the same workload: `cpu_sample_graphs` graphs x `cpu_sample_steps` rollout
    steps, scaled to the 2048-graph step."""
    import numpy as np
    import torch
    from threadpoolctl import threadpool_limits
    import msmp_pde_amd as mp
    from msmp_pde_amd.synthetic import make_case
    from oracle import msmp_oracle as O, msmp_oracle_torch as OT
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    cores = min(avail, 16)          # the GPU box's CPU share for one GPU; BLAS threads actually used
    b = args.cpu_sample_graphs
    torch.manual_seed(0)
    case = make_case(args.experiment, b, seed=0, device='cuda', dtype=torch.float64, neighbors=args.neighbors)
    model = mp.MODEL_NAMES[args.model](case.pde, time_window=25, eq_variables=eqv, hidden_layer=6)
    sd = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    steps = [50] * b
    data, labels = case.creator.create_data(case.u_super, steps)
    g = case.creator.create_graph(data, labels, case.x, case.variables, steps)
    from types import SimpleNamespace
    gn = SimpleNamespace(**{k: v.detach().cpu().numpy() for k, v in g.__dict__.items() if torch.is_tensor(v)})
    traj = case.u_super.cpu().numpy()
    pde_name = repr(case.pde)
    times = []
    torch.set_num_threads(cores)
    with threadpool_limits(limits=cores):
        pred = OT.solver_forward(kind, sd, gn, case.pde, 25, eqv, 6)
        step = 50
        for _ in range(args.cpu_sample_steps):
            t0 = time.perf_counter()
            step += 25
            _, lab = O.create_data(traj, [step] * b, 25)
            gn = O.create_next_graph(pde_name, case.pde, 25, gn, pred, lab, [step] * b)
            pred = OT.solver_forward(kind, sd, gn, case.pde, 25, eqv, 6)
            times.append(time.perf_counter() - t0)
    t_step = float(np.median(times)) * (args.graphs / b)      # scaled to the full batch
    return {'value': 1.0 / t_step, 'unit': 'rollout-steps/s', 'cores': cores, 'kind': 'port',
            'sample': f'torch-CPU float64 oracle, {b} graphs x {args.cpu_sample_steps} rollout steps after 1 warm-up, '
                      f'median step time scaled x{args.graphs / b:g} to {args.graphs} graphs'}


def main():
    args = parse()
    import torch
    import msmp_pde_amd as mp
    from msmp_pde_amd import dist as D, _lib
    from msmp_pde_amd.synthetic import make_case, EXPERIMENTS

    rank, world, local = D.init_from_env(args.dist_backend)
    local = local % max(torch.cuda.device_count(), 1)      # (testing) more ranks than GPUs: share devices
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (no CPU fallback)'
    assert world == args.gpus or world == 1, f'--gpus {args.gpus} but WORLD_SIZE={world}'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    L = mp.lib()
    split_path = not args.fp32_mfma
    L.msmp_tune(b'split', int(split_path))
    for kv in args.tune:
        k, v = kv.split('=')
        assert L.msmp_tune(k.encode(), int(v)) == 0, kv

    exp = args.experiment
    eqv = dict(EXPERIMENTS[exp])
    cls = mp.MODEL_NAMES[args.model]
    kind = cls.__name__
    torch.manual_seed(0)                       # same weights on every rank
    case = make_case(exp, args.graphs, seed=1000 + rank, device=dev, dtype=torch.float32, neighbors=args.neighbors)
    model = cls(case.pde, time_window=25, eq_variables=eqv, hidden_layer=6).to(dev).eval()
    bsz = args.graphs
    steps0 = [50] * bsz
    data, labels = case.creator.create_data(case.u_super, steps0)
    graph = case.creator.create_graph(data, labels, case.x, case.variables, steps0)
    n_nodes, n_edges = graph.x.shape[0], graph.edge_index.shape[1]

    def rollout_step(i, pred):
        step = 75 + 25 * (i % 7)               # the reference unrolls steps 75..225 (train_helper.py:255)
        same = [step] * bsz
        _, lab = case.creator.create_data(case.u_super, same)
        g = case.creator.create_next_graph(graph, pred, lab, same)
        return model(g)

    with torch.no_grad():
        pred = model(graph)
        for i in range(args.warmup):
            pred = rollout_step(i, pred)
        mask = 0b1111111 if args.time_all_kernels else (1 << _lib.K_EDGE_MLP) | (1 << _lib.K_NODE_PROJ)
        L.msmp_timing_reset()
        L.msmp_timing_enable(mask)
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            pred = rollout_step(args.warmup + i, pred)
        torch.cuda.synchronize()
        D.barrier()
        elapsed = time.perf_counter() - t0
        L.msmp_timing_enable(0)
    finite = bool(torch.isfinite(pred).all().item())
    elapsed = D.reduce_scalar(elapsed, 'max')
    total_steps = D.reduce_scalar(args.steps, 'sum')

    if rank != 0:
        return
    nv = len(eqv) + 1
    k_msg = model.gnn_layers[0].message_net_1[0].in_features        # 2H + Tw + 1 + nv (Tw = 2*tw for the *2D classes)
    # Row L1 (message MLP) in the reference's dense formulation: 2*E*K_msg*H + 2*E*H*H per layer.
    flop_l1_dense = 2.0 * n_edges * k_msg * H + 2.0 * n_edges * H * H
    # What the dominant kernel executes in the factorised form: message_net_2 only (message_net_1 became the
    # per-node projections of node_proj_kernel: 2 * N * 2 * (H + 32*tail_chunks) * H).
    flop_edge_exec = 2.0 * n_edges * H * H
    n_launch, ms_total = _lib.timing_read(_lib.K_EDGE_MLP)
    n_proj, ms_proj = _lib.timing_read(_lib.K_NODE_PROJ)
    t_launch = ms_total / max(n_launch, 1) * 1e-3
    t_proj = ms_proj / max(n_proj, 1) * 1e-3
    factorised = n_proj > 0
    flop_exec = flop_edge_exec if factorised else flop_l1_dense
    achieved = flop_exec / t_launch / 1e12 if n_launch else None
    alg_tflops = flop_l1_dense / (t_launch + t_proj) / 1e12 if n_launch else None
    traffic = None
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')     # HBM bytes per launch from the rocprofv3 PMC passes
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            traffic = (tj.get('edge_mlp_kernel_occ4') or tj.get('edge_mlp_kernel_occ2') or tj.get('edge_mlp_kernel') or {}).get('hbm_bytes_per_launch')
        except Exception:
            traffic = None
    # the split path's ceiling is the f16 matrix pipe doing 3 MFMAs per fp32 K-step; the fp32-MFMA path's is the fp32 peak
    peak_eq = PEAK_FP16_MFMA_TFLOPS / 3.0 if split_path else PEAK_FP32_MFMA_TFLOPS
    out = {
        'metric': 'rollout-steps/sec (whole node), E2 nx=100 tw=25',
        'value': total_steps / elapsed, 'unit': 'rollout-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'{exp} {args.model} ({kind}), {bsz} graphs/GPU x nx=100, time_window=25, '
                               f'{"6 gated layer pairs" if model.GATED else "6 layers"}, '
                               f'{"radius graph n=" if exp in ("E2", "MSWG3") else "knn graph k="}{args.neighbors}', 'graphs_per_gpu': bsz, 'nodes': n_nodes,
                   'edges': n_edges, 'parallelism': f'dp{world} (graph-sharded, no collective in the rollout)',
                   'graph_steps_per_s': total_steps * bsz / elapsed, 'output_finite': finite},
        # `achieved` counts the fp32 GEMM FLOPs the dominant kernel computes (conservative: the factorised form
        # removed 69 % of row L1's dense FLOPs).  The kernel evaluates them on the fp16 matrix pipe (2-way fp16 split
        # of both operands, 3 MFMAs per K=16 step, fp32-class accuracy), so `peak` is that pipe's dense peak / 3 and
        # `frac` equals the literal f16-MFMA utilisation (`matrix_pipe`); `vs_fp32_mfma_peak` prices the same FLOPs
        # against what an fp32-MFMA implementation could reach at best (it exceeds 1).  `algorithmic` prices row L1 (node_proj + edge kernels) at SURVEY 8d's dense figure.
        'roofline': {'bound': 'mfma', 'kernel': 'edge_mlp_kernel (message_net_2 + Swish + per-target mean'
                     + (', factorised message_net_1' if factorised else ', dense message_net_1')
                     + ('; fp32 GEMM on the fp16 matrix pipe via 2-way fp16 split)' if split_path else '; fp32 MFMA)'),
                     'achieved': achieved, 'peak': peak_eq,
                     'unit': 'TFLOP/s', 'frac': (achieved / peak_eq) if achieved else None,
                     'peak_note': ('fp32-equivalent peak of the 3-MFMA fp16 split = dense f16 MFMA peak / 3' if split_path
                                   else 'dense fp32 MFMA peak'),
                     'vs_fp32_mfma_peak': (achieved / PEAK_FP32_MFMA_TFLOPS) if achieved else None,
                     'traffic': traffic, 'launches': n_launch, 'avg_launch_ms': t_launch * 1e3,
                     'executed_gflop_per_launch': flop_exec / 1e9,
                     'matrix_pipe': ({'dtype': 'f16 (3 MFMAs per fp32 K=16 step)', 'executed_tflops': 3 * achieved,
                                      'peak_tflops': PEAK_FP16_MFMA_TFLOPS, 'frac': 3 * achieved / PEAK_FP16_MFMA_TFLOPS}
                                     if split_path and achieved else None),
                     'algorithmic': {'row': 'L1+L2 message MLP + mean = node_proj_kernel + edge_mlp_kernel per layer',
                                     'gflop_per_layer': flop_l1_dense / 1e9, 'ms_per_layer': (t_launch + t_proj) * 1e3,
                                     'tflops': alg_tflops, 'frac': alg_tflops / PEAK_FP32_MFMA_TFLOPS if alg_tflops else None},
                     'share_of_step': (ms_total / args.steps) / (elapsed / args.steps * 1e3) if n_launch else None},
    }
    if args.time_all_kernels:
        names = {_lib.K_SCATTER_MEAN: 'scatter_mean', _lib.K_NODE_UPDATE: 'node_update', _lib.K_NORM: 'norm_blend',
                 _lib.K_LEM: 'lem_encoder', _lib.K_NODE_PROJ: 'node_project', _lib.K_DECODER: 'decoder'}
        out['kernels_ms_per_step'] = {'edge_mlp': ms_total / args.steps}
        for k, nm in names.items():
            n_k, ms_k = _lib.timing_read(k)
            out['kernels_ms_per_step'][nm] = ms_k / args.steps
        n_sc, ms_sc = _lib.timing_read(_lib.K_SCATTER_MEAN)
        if n_sc:
            sc_bytes = n_edges * H * 4 + n_edges * 0 + (n_nodes + 1) * 4 + n_nodes * H * 4
            out['scatter_hbm'] = {'achieved_GBps': sc_bytes / (ms_sc / n_sc * 1e-3) / 1e9, 'peak_GBps': 8000.0,
                                  'algorithmic_bytes_per_launch': sc_bytes}
    if rank == 0 and 'scatter_hbm' not in out:
        # row L2 standalone ("HBM GB/s on the scatter"): the default layer path fuses the mean into the message kernel,
        # so the CSR segmented-mean kernel is timed here on a message tensor of the workload's size, outside the timed region
        from msmp_pde_amd.graph import structure_of
        gs = structure_of(graph)
        msg = torch.randn(n_edges, H, device=dev)
        agg = torch.empty(n_nodes, H, device=dev)
        L = _lib.lib()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for i in range(23):
            if i == 3:
                ev0.record()
            _lib.check(L.msmp_scatter_mean_f32(_lib.ptr(msg), _lib.ptr(gs.rowptr), n_nodes, _lib.ptr(agg), _lib.current_stream()), 'scatter')
        ev1.record()
        torch.cuda.synchronize()
        sc_ms = ev0.elapsed_time(ev1) / 20
        sc_bytes = n_edges * H * 4 + (n_nodes + 1) * 4 + n_nodes * H * 4
        out['scatter_hbm'] = {'kernel': 'scatter_mean_kernel (standalone row L2; fused into the message kernel on the default path)',
                              'achieved_GBps': sc_bytes / (sc_ms * 1e-3) / 1e9, 'peak_GBps': 8000.0,
                              'frac': sc_bytes / (sc_ms * 1e-3) / 1e9 / 8000.0, 'algorithmic_bytes_per_launch': sc_bytes,
                              'avg_launch_ms': sc_ms}
        del msg, agg
    if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
        out['cpu_baseline'] = cpu_baseline(args, kind, eqv)
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
