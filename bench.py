#!/usr/bin/env python3
"""bench.py -- rollout-steps/sec of the MSMP-PDE message-passing rollout step on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--graphs 2048] [--model MSMP-PDE] [--scaling strong|weak]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): E2 (Burgers-type CE, nx=100, time_window=25, radius graph n=3,
588 edges/graph), model MSMP-PDE = MP_PDE_SolverLEMLinGated (6 gated layer pairs), a 2048-graph batch,
random-init weights, synthetic trajectories resident in HBM.  One "step" = one rollout step of the
reference's unrolled evaluation (experiments/train_helper.py:255-261): create_next_graph (state update)
+ model(graph) under no_grad, on the whole batch.

Multi-GPU: graphs are independent, so the batch is sharded by graph with no
data-path collective.  `--gpus N` without a torchrun environment starts N child processes itself (one per GPU, before
this process touches the GPU; a process that has initialised the GPU is never exec'd).  The headline at N > 1 is the
metric's own form, STRONG scaling: the 2048-graph batch split over the ranks (msmp_pde_amd.dist.shard_range), value =
steps / max-over-ranks time; the same run also measures WEAK scaling (2048 graphs on every rank, object `weak`).
`--scaling weak` makes the weak number the headline instead.

Extra objects on the JSON line:
  roofline      dominant kernel (the edge-message kernel): see DESIGN.md section 6.
  scatter_hbm   row L2 standalone: achieved HBM GB/s of the CSR segmented-mean kernel at the workload's size.
  cpu_baseline  the CPU oracle (torch-CPU float64 edition, kind "port") timed on this host on a bounded sample.
  config0       BASELINE.json configs[0] (E2 MP-PDE, 32 graphs): the HIP path and the CPU port side by side.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_FP16_MFMA_TFLOPS = 2500.0    # same guide, "Peak BF16/FP16 MFMA" (dense)
PEAK_HBM_GBPS = 8000.0            # same guide, HBM3E peak (spec); ~6300 achievable
H = 128


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--graphs', type=int, default=2048, help='graphs of the batch (strong: in total; weak: per GPU)')
    ap.add_argument('--scaling', choices=('strong', 'weak'), default='strong', help='headline mode at N > 1 (SURVEY 8d: the metric is the 2048-graph batch split over the GPUs)')
    ap.add_argument('--model', default='MSMP-PDE', help='MSMP-PDE | Gated | MP-PDE | ...')
    ap.add_argument('--experiment', default='E2')
    ap.add_argument('--neighbors', type=int, default=3, help='n of the graph builder (radius n*dx / knn k); 8, 16 = the MSWG3 edge-count stress of SURVEY 8(d)')
    ap.add_argument('--preheat-s', type=float, default=1.0, help='untimed steps run for this long before the timed window (sustained clocks, not boost)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip scatter_hbm / config0 / the second scaling mode (profiling runs)')
    ap.add_argument('--cpu-sample-graphs', type=int, default=128)
    ap.add_argument('--cpu-sample-steps', type=int, default=3)
    ap.add_argument('--dist-backend', default=None, help='torch.distributed backend (default nccl = RCCL); gloo lets ranks share one GPU for testing')
    ap.add_argument('--sub-batches', type=int, default=0, help='step the rank\'s batch as this many independent sub-batches of whole graphs on as many streams (0 = auto: 2 for shards of 64..1024 graphs, else 1)')
    ap.add_argument('--tune', action='append', default=[], metavar='KEY=VALUE', help='msmp_tune override for kernel A/B runs (e.g. lem=1)')
    ap.add_argument('--fp32-mfma', action='store_true', help='use the fp32-MFMA kernels instead of the fp16-split matrix path')
    ap.add_argument('--time-all-kernels', action='store_true', help='event-time every kernel family, not only the dominant one')
    ap.add_argument('--launcher-selftest', action='store_true', help='(tests) ranks only join the process group and count themselves; needs no GPU')
    ap.add_argument('--selftest-fail-rank', type=int, default=-1, help='(tests) this rank exits with code 3 before joining the group: the launcher must stop the others')
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# N > 1 without torchrun: this process only spawns (it never touches the GPU)
# ---------------------------------------------------------------------------------------------------------------------
def launch_children(args, argv, poll_s=0.2, grace_s=10.0):
    """One child per rank, started before anything touches the GPU here.  The children are POLLED: when one exits non-zero the
    others would sit in a collective until the backend's timeout, so they are terminated (then killed) at once and the launcher
    exits with the failing rank's code.  Rank 0 inherits stdout (the JSON line); every rank's stderr goes to a temporary file that
    is replayed with a `[rank r]` prefix when that rank failed (ranks != 0 have their stdout there too)."""
    import tempfile
    import time
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs, logs = [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR='127.0.0.1',
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
        log = tempfile.TemporaryFile(mode='w+')
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=None if r == 0 else log, stderr=log))
    rc, failed = 0, None
    alive = set(range(args.gpus))
    while alive and failed is None:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0:
                rc, failed = code, r
                break
        if alive and failed is None:
            time.sleep(poll_s)
    if failed is not None:                      # stop the survivors: SIGTERM, then SIGKILL after a grace period (exact PIDs we started)
        for r in alive:
            procs[r].terminate()
        deadline = time.time() + grace_s
        for r in alive:
            try:
                procs[r].wait(timeout=max(0.0, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
                procs[r].wait()
        print(f'bench.py launcher: rank {failed} exited with code {rc}; the other {len(alive)} rank(s) were stopped', file=sys.stderr)
    for r, log in enumerate(logs):
        log.seek(0)
        text = log.read()
        log.close()
        if text and (failed is not None or os.environ.get('MSMP_BENCH_VERBOSE')):
            for line in text.splitlines()[-40:]:
                print(f'[rank {r}] {line}', file=sys.stderr)
    return rc if rc >= 0 else 128 - rc          # a signal-terminated child reports as 128 + signal


# ---------------------------------------------------------------------------------------------------------------------
def source_hash():
    """Hash of the kernel sources: stamps profiles/traffic.json so a stale PMC figure is never reported for other code."""
    h = hashlib.sha256()
    base = os.path.join(ROOT, 'msmp-pde_amd', 'csrc')
    for f in sorted(os.listdir(base)):
        if f.endswith(('.hip', '.h')):
            h.update(f.encode())
            h.update(open(os.path.join(base, f), 'rb').read())
    return h.hexdigest()[:16]


def cpu_baseline(model_name, experiment, graphs_total, sample_graphs, sample_steps, neighbors=3):
    """The oracle (torch-CPU float64 edition, `kind: port`: the reference itself is PyTorch on the CPU) on a bounded
    sample of the workload: `sample_graphs` graphs x `sample_steps` rollout steps, scaled to `graphs_total`."""
    import numpy as np
    import torch
    from threadpoolctl import threadpool_limits
    import msmp_pde_amd as mp
    from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
    from oracle import msmp_oracle as O, msmp_oracle_torch as OT
    from types import SimpleNamespace
    eqv = dict(EXPERIMENTS[experiment])
    kind = mp.MODEL_NAMES[model_name].__name__
    avail = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
    cores = min(avail, 16)          # the GPU box's CPU share for one GPU; BLAS threads actually used
    b = sample_graphs
    torch.manual_seed(0)
    case = make_case(experiment, b, seed=0, device='cuda', dtype=torch.float64, neighbors=neighbors)
    model = mp.MODEL_NAMES[model_name](case.pde, time_window=25, eq_variables=eqv, hidden_layer=6)
    sd = {k: v.detach().numpy().astype(np.float64) for k, v in model.state_dict().items()}
    steps = [50] * b
    data, labels = case.creator.create_data(case.u_super, steps)
    g = case.creator.create_graph(data, labels, case.x, case.variables, steps)
    gn = SimpleNamespace(**{k: v.detach().cpu().numpy() for k, v in g.__dict__.items() if torch.is_tensor(v)})
    traj = case.u_super.cpu().numpy()
    pde_name = repr(case.pde)
    times = []
    torch.set_num_threads(cores)
    with threadpool_limits(limits=cores):
        pred = OT.solver_forward(kind, sd, gn, case.pde, 25, eqv, 6)
        step = 50
        for _ in range(sample_steps):
            t0 = time.perf_counter()
            step += 25
            _, lab = O.create_data(traj, [step] * b, 25)
            gn = O.create_next_graph(pde_name, case.pde, 25, gn, pred, lab, [step] * b)
            pred = OT.solver_forward(kind, sd, gn, case.pde, 25, eqv, 6)
            times.append(time.perf_counter() - t0)
    t_step = float(np.median(times)) * (graphs_total / b)      # scaled to the full batch
    scaled = f', median step time scaled x{graphs_total / b:g} to {graphs_total} graphs' if graphs_total != b else ', median step time'
    return {'value': 1.0 / t_step, 'unit': 'rollout-steps/s', 'cores': cores, 'kind': 'port',
            'sample': f'torch-CPU float64 oracle ({kind}), {b} graphs x {sample_steps} rollout steps after 1 warm-up{scaled}'}


class Workload:
    """One rank's share of the batch, resident in HBM, and its rollout step."""

    def __init__(self, args, mp, dev, n_graphs, seed):
        import torch
        from msmp_pde_amd.synthetic import make_case, EXPERIMENTS
        self.eqv = dict(EXPERIMENTS[args.experiment])
        cls = mp.MODEL_NAMES[args.model]
        torch.manual_seed(0)                       # same weights on every rank
        self.case = make_case(args.experiment, n_graphs, seed=seed, device=dev, dtype=torch.float32, neighbors=args.neighbors)
        self.model = cls(self.case.pde, time_window=25, eq_variables=self.eqv, hidden_layer=6).to(dev).eval()
        self.bsz = n_graphs
        steps0 = [50] * n_graphs
        data, labels = self.case.creator.create_data(self.case.u_super, steps0)
        self.graph = self.case.creator.create_graph(data, labels, self.case.x, self.case.variables, steps0)
        self.n_nodes, self.n_edges = self.graph.x.shape[0], self.graph.edge_index.shape[1]
        self.x0 = self.graph.x.clone()              # the ground-truth window every unrolled trajectory starts from
        self.pred = None
        self.i = 0

    def first(self):
        self.pred = self.model(self.graph)

    def step(self):
        step = 75 + 25 * (self.i % 7)               # the reference unrolls steps 75..225 (train_helper.py:255)
        # ... and starts every unrolled trajectory from ground truth (train_helper.py:243-261): after the 7 steps of one trajectory the
        # next one restarts from the true window.  Same kernels and bytes as a step that feeds the prediction back (the state
        # update assigns whichever tensor it is given); what it avoids is an UNTRAINED network fed its own output for hundreds of
        # steps (WE3: out = u + cumsum(dt) diff with cumsum(dt) up to 10 grows by ~10 per step and leaves the fp16-split path's
        # |x| <= 255 after ~25 steps, at which point Solver.forward switches the model to the exact-fp32 kernels: round 4's
        # range policy; round 3 saturated silently there).
        src = self.x0 if self.i % 7 == 0 else self.pred
        self.i += 1
        same = [step] * self.bsz
        _, lab = self.case.creator.create_data(self.case.u_super, same)
        g = self.case.creator.create_next_graph(self.graph, src, lab, same)
        self.pred = self.model(g)


class SplitWorkload:
    """The rank's batch as S independent sub-batches of whole graphs, each a Workload stepping on a stream of its own (state
    update and forward): the ~20 dependent launches of a step become S chains whose kernels overlap (scripts/sub_batches.py:
    1.06 -> 0.92 ms per step at 256 graphs, the 8-GPU strong-scaling shard; 3.41 -> 3.21 at 1024).  Same model, same work."""

    def __init__(self, args, mp, dev, n_graphs, seed, parts):
        import torch
        self.torch = torch
        sizes = [n_graphs // parts + (1 if i < n_graphs % parts else 0) for i in range(parts)]
        self.parts = [Workload(args, mp, dev, n, seed=seed + 7919 * i) for i, n in enumerate(sizes) if n > 0]
        for p in self.parts[1:]:
            p.model = self.parts[0].model
        self.streams = [torch.cuda.Stream(device=dev) for _ in self.parts]
        self.parts[0].model.warm_caches(dev)      # the shared model's packed blobs are built on THIS stream, which every part's stream waits for
        mp.lib().msmp_tune(b'lem_share', len(self.parts))       # the LEM launches of the sub-batches share the CUs: each plans its rounds for its share
        for s in self.streams:
            s.wait_stream(torch.cuda.current_stream(dev))
        self.model, self.graph, self.bsz = self.parts[0].model, self.parts[0].graph, n_graphs
        self.n_nodes, self.n_edges = sum(p.n_nodes for p in self.parts), sum(p.n_edges for p in self.parts)

    def first(self):
        for p, s in zip(self.parts, self.streams):
            with self.torch.cuda.stream(s):
                p.first()

    def step(self):
        for p, s in zip(self.parts, self.streams):
            with self.torch.cuda.stream(s):
                p.step()

    @property
    def pred(self):
        self.torch.cuda.synchronize()
        return self.torch.cat([p.pred for p in self.parts], 0)


def timed_run(wl, D, torch, steps, warmup, preheat_s, before_timed=None, after_timed=None):
    """W untimed warm-up steps, `preheat_s` seconds of further untimed steps (so that the timed window runs at the sustained,
    power-limited clock and not at boost), then EXACTLY `steps` steps between barrier + synchronize brackets."""
    with torch.no_grad():
        wl.first()
        for _ in range(warmup):
            wl.step()
        torch.cuda.synchronize()
        t_heat, n_heat = time.perf_counter(), 0
        while time.perf_counter() - t_heat < preheat_s:
            for _ in range(4):
                wl.step()
            torch.cuda.synchronize()
            n_heat += 4
        if before_timed:
            before_timed()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            wl.step()
        torch.cuda.synchronize()
        D.barrier()
        elapsed = time.perf_counter() - t0
        if after_timed:
            after_timed()
    return D.reduce_scalar(elapsed, 'max'), n_heat


def run_rank(args):
    import torch
    import msmp_pde_amd as mp
    from msmp_pde_amd import dist as D, _lib

    if args.launcher_selftest and int(os.environ.get('RANK', '0')) == args.selftest_fail_rank:
        print('selftest: this rank fails on purpose', file=sys.stderr)
        return 3
    rank, world, local = D.init_from_env(args.dist_backend)
    if world != args.gpus:
        print(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)
        return 2
    if args.launcher_selftest:
        seen = int(round(D.reduce_scalar(1, 'sum')))
        blocks = [D.shard_range(args.graphs, r, world) for r in range(world)]
        if rank == 0:
            print(json.dumps({'ranks_seen': seen, 'n_gpus': world, 'shards': blocks}), flush=True)
        return 0 if seen == args.gpus else 2
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (no CPU fallback)'
    local = local % max(torch.cuda.device_count(), 1)      # (testing) more ranks than GPUs: share devices
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    L = mp.lib()
    split_path = not args.fp32_mfma
    L.msmp_tune(b'split', int(split_path))
    for kv in args.tune:
        k, v = kv.split('=')
        assert L.msmp_tune(k.encode(), int(v)) == 0, kv
    ranks_seen = int(round(D.reduce_scalar(1, 'sum')))       # an all-reduce of 1 over RCCL: every rank is really there
    if ranks_seen != args.gpus:
        print(f'bench.py: {ranks_seen} ranks answered, --gpus {args.gpus}', file=sys.stderr)
        return 2

    modes = ['strong', 'weak'] if args.scaling == 'strong' else ['weak', 'strong']
    if world == 1 or args.no_extras:
        modes = modes[:1]
    results = {}
    timing = {}
    for mode in modes:
        if mode == 'strong':
            g0, g1 = D.shard_range(args.graphs, rank, world)
            n_graphs = g1 - g0
        else:
            n_graphs = args.graphs
        # sub-batches on streams: asked for, or (auto) for shards of at most 1024 graphs, where a step is bound by the latency of its
        # dependent launches; the 2048-graph line stays one batch so that the dominant kernel's launches are timed undisturbed
        parts = args.sub_batches if args.sub_batches > 0 else (2 if n_graphs <= 1024 and n_graphs >= 64 else 1)
        wl = SplitWorkload(args, mp, dev, n_graphs, 1000 + rank, parts) if parts > 1 else Workload(args, mp, dev, n_graphs, seed=1000 + rank)
        head = mode == modes[0]

        def start_events():
            mask = 0b1111111 if args.time_all_kernels else (1 << _lib.K_EDGE_MLP) | (1 << _lib.K_NODE_PROJ)
            L.msmp_timing_reset()
            L.msmp_timing_enable(mask)

        elapsed, n_heat = timed_run(wl, D, torch, args.steps if head else max(args.steps // 2, 5), args.warmup, args.preheat_s,
                                    start_events if head else None, (lambda: L.msmp_timing_enable(0)) if head else None)
        k_steps = args.steps if head else max(args.steps // 2, 5)
        finite = bool(torch.isfinite(wl.pred).all().item())
        graphs_all = int(round(D.reduce_scalar(n_graphs, 'sum')))
        # strong: a step advances the ONE shared batch, so steps/s = K / t.  weak: every rank advances its own batch: N K / t.
        value = k_steps / elapsed if mode == 'strong' else world * k_steps / elapsed
        results[mode] = {'value': value, 'ms_per_step': elapsed / k_steps * 1e3, 'graphs_total': graphs_all, 'graphs_this_rank': n_graphs,
                         'steps': k_steps, 'graph_steps_per_s': graphs_all * k_steps / elapsed, 'output_finite': finite, 'preheat_steps': n_heat,
                         'range_exact': bool(getattr(wl.model, '_range_exact', False))}
        if head:
            timing = {'edge': _lib.timing_read(_lib.K_EDGE_MLP), 'proj': _lib.timing_read(_lib.K_NODE_PROJ),
                      'n_nodes': wl.n_nodes // parts, 'n_edges': wl.n_edges // parts, 'sub_batches': parts, 'model': wl.model, 'graph': wl.graph, 'steps': k_steps,
                      'all': ({k: _lib.timing_read(k) for k in range(7)} if args.time_all_kernels else None)}
            head_wl = wl
        else:
            del wl
            torch.cuda.empty_cache()

    if rank != 0:
        return 0
    head = results[modes[0]]
    model, graph = timing['model'], timing['graph']
    n_nodes, n_edges = timing['n_nodes'], timing['n_edges']
    kind = type(model).__name__
    exp = args.experiment
    k_msg = model.gnn_layers[0].message_net_1[0].in_features        # 2H + Tw + 1 + nv (Tw = 2*tw for the *2D classes)
    # Row L1 (message MLP) in the reference's dense formulation: 2*E*K_msg*H + 2*E*H*H per layer.
    flop_l1_dense = 2.0 * n_edges * k_msg * H + 2.0 * n_edges * H * H
    n_launch, ms_total = timing['edge']
    n_proj, ms_proj = timing['proj']
    t_launch = ms_total / max(n_launch, 1) * 1e-3
    t_proj = ms_proj / max(n_proj, 1) * 1e-3
    # Which form ran: `folded` = node tiles staged in LDS, P / Q projected inside the message kernel;
    # `factorised` = separate projection kernel; else the literal per-edge GEMM.
    from msmp_pde_amd.graph import structure_of
    tiles = structure_of(graph).tiles()
    folded = tiles is not None and n_proj == 0 and L.msmp_tune_query(b'tile') == 2 and split_path
    factorised = folded or n_proj > 0
    k_tail = 32 * ((k_msg - 2 * H + 31) // 32)                      # [u | pos | vars] columns, padded to whole chunks
    flop_gemm2 = 2.0 * n_edges * H * H                              # message_net_2: the per-edge GEMM that remains
    flop_proj_alg = 2.0 * n_nodes * 2 * (H + k_tail) * H            # P and Q of every node once (what the projection kernel executes)
    # useful fp32 FLOPs of the dominant kernel per launch (no halo recomputation, no padding)
    flop_exec = (flop_gemm2 + flop_proj_alg) if folded else (flop_gemm2 if factorised else flop_l1_dense)
    # what the matrix pipe really executes in the folded kernel: one 32-slot node block per tile (halo nodes recomputed, slots padded)
    flop_pipe = (flop_gemm2 + 2.0 * tiles[0].n_tiles * 32 * 2 * (H + k_tail) * H) if folded else flop_exec
    achieved = flop_exec / t_launch / 1e12 if n_launch else None
    alg_tflops = flop_l1_dense / (t_launch + t_proj) / 1e12 if n_launch else None
    # HBM traffic of the dominant kernel: PMC passes of scripts/profile_gpu.sh, valid only for the sources and workload they were taken on
    traffic, traffic_note = None, 'no profiles/traffic.json'
    tpath = os.path.join(ROOT, 'profiles', 'traffic.json')
    if os.path.exists(tpath):
        try:
            tj = json.load(open(tpath))
            stamp = tj.get('_stamp', {})
            want = {'source_hash': source_hash(), 'workload': f'{exp}/{args.model}/{head["graphs_this_rank"]}/n{args.neighbors}', 'split': int(split_path)}
            sym = stamp.get('dominant_kernel')
            if all(stamp.get(k) == v for k, v in want.items()) and sym in tj:
                traffic = tj[sym].get('hbm_bytes_per_launch')
                traffic_note = f'rocprofv3 --pmc FETCH_SIZE (x2, gfx950 correction) + WRITE_SIZE, kernel {sym}, profile {stamp.get("tag")}'
            else:
                traffic_note = f'profiles/traffic.json was taken on other sources / workload ({stamp}); not reported'
        except Exception as exc:                        # noqa: BLE001
            traffic_note = f'profiles/traffic.json unreadable: {exc}'
    peak_eq = PEAK_FP16_MFMA_TFLOPS / 3.0 if split_path else PEAK_FP32_MFMA_TFLOPS
    # algorithmic HBM bytes of the fused message + mean launch (SURVEY 8d "fused layer" accounting for this kernel's share):
    # read the node rows it consumes once, the CSR, write the aggregate
    alg_bytes = ((2 * n_nodes * H * 4) if (factorised and not folded) else n_nodes * (H + k_msg - 2 * H) * 4) + n_edges * 4 + (n_nodes + 1) * 4 + n_nodes * H * 4
    frac_mfma = (achieved / peak_eq) if achieved else None
    frac_hbm = (alg_bytes / t_launch / 1e9 / PEAK_HBM_GBPS) if n_launch else None
    out = {
        'metric': 'rollout-steps/sec (whole node), E2 nx=100 tw=25',
        'value': head['value'], 'unit': 'rollout-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': head['ms_per_step'], 'higher_is_better': True,
        'scaling': modes[0], 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic', 'ranks_seen': ranks_seen,
        'config': {'workload': f'{exp} {args.model} ({kind}), {head["graphs_total"]} graphs x nx=100, time_window=25, '
                               f'{"6 gated layer pairs" if getattr(model, "GATED", False) else "6 layers"}, '
                               f'{"radius graph n=" if exp in ("E2", "MSWG3") else "knn graph k="}{args.neighbors}',
                   'graphs_total': head['graphs_total'], 'graphs_per_gpu': head['graphs_this_rank'], 'nodes_per_gpu': n_nodes, 'edges_per_gpu': n_edges,
                   'parallelism': f'dp{world} (graph-sharded, no collective in the rollout)',
                   'sub_batches': timing.get('sub_batches', 1),
                   'graph_steps_per_s': head['graph_steps_per_s'],
                   'edge_steps_per_s': head['graph_steps_per_s'] * n_edges / max(head['graphs_this_rank'], 1),
                   'output_finite': head['output_finite'],
                   # True would mean the range policy moved the model to the exact-fp32 kernels inside the run (data left |x| <= 255): not the path this line is about
                   'range_policy_switched_to_exact_fp32': head['range_exact'],
                   'preheat': f'{head["preheat_steps"]} untimed steps (>= {args.preheat_s:g} s) after the {args.warmup} warm-up steps'},
        # `achieved` counts the USEFUL fp32 GEMM FLOPs the dominant kernel computes (message_net_2 per edge + the per-node projections
        # once per node; the factorised form removed 69 % of row L1's dense FLOPs).  They execute on the fp16 matrix pipe (2-way fp16 split of both operands, 3 MFMAs per K=16 step, fp32-class
        # accuracy), so `peak` is that pipe's dense peak / 3 and `frac` equals the literal f16-MFMA utilisation (`matrix_pipe`).
        # `bound` names the nearer of the two rooflines; `bound_detail` says what the phase profile shows actually limits it.
        'roofline': {'bound': 'mfma' if (frac_mfma or 0) >= (frac_hbm or 0) else 'hbm',
                     'bound_detail': 'neither roof: see DESIGN.md section 4 (phase profile: node-row gather latency, LDS port and activation VALU; matrix pipe and HBM both under 0.4)',
                     'kernel': 'edge message kernel (message_net_2 + Swish + per-target mean'
                     + (', node tiles staged in LDS with the per-node projections of message_net_1 folded in' if folded
                        else ', factorised message_net_1 (separate projection kernel)' if factorised else ', dense message_net_1')
                     + ('; fp32 GEMM on the fp16 matrix pipe via 2-way fp16 split)' if split_path else '; fp32 MFMA)'),
                     'achieved': achieved, 'peak': peak_eq,
                     'unit': 'TFLOP/s', 'frac': frac_mfma,
                     'peak_note': ('fp32-equivalent peak of the 3-MFMA fp16 split = dense f16 MFMA peak / 3' if split_path
                                   else 'dense fp32 MFMA peak'),
                     'vs_fp32_mfma_peak': (achieved / PEAK_FP32_MFMA_TFLOPS) if achieved else None,
                     'hbm': {'algorithmic_bytes_per_launch': alg_bytes, 'achieved_GBps': alg_bytes / t_launch / 1e9 if n_launch else None,
                             'peak_GBps': PEAK_HBM_GBPS, 'frac': frac_hbm},
                     'traffic': traffic, 'traffic_note': traffic_note, 'launches': n_launch, 'avg_launch_ms': t_launch * 1e3,
                     'executed_gflop_per_launch': flop_exec / 1e9,
                     'matrix_pipe': ({'dtype': 'f16 (3 MFMAs per fp32 K=16 step)', 'executed_tflops': 3 * flop_pipe / t_launch / 1e12,
                                      'peak_tflops': PEAK_FP16_MFMA_TFLOPS, 'frac': 3 * flop_pipe / t_launch / 1e12 / PEAK_FP16_MFMA_TFLOPS,
                                      'note': 'everything the pipe executes, incl. halo recomputation and padded node slots of the folded projections'}
                                     if split_path and achieved else None),
                     'algorithmic': {'row': 'L1+L2 message MLP + mean = node projection + message kernels per layer',
                                     'gflop_per_layer': flop_l1_dense / 1e9, 'ms_per_layer': (t_launch + t_proj) * 1e3,
                                     'tflops': alg_tflops, 'frac': alg_tflops / PEAK_FP32_MFMA_TFLOPS if alg_tflops else None},
                     'share_of_step': (ms_total / timing['steps']) / head['ms_per_step'] if n_launch else None},
    }
    if len(modes) > 1:
        other = results[modes[1]]
        out[modes[1]] = {'value': other['value'], 'unit': 'rollout-steps/s', 'ms_per_step': other['ms_per_step'], 'steps': other['steps'],
                         'graphs_total': other['graphs_total'], 'graphs_per_gpu': other['graphs_this_rank'],
                         'graph_steps_per_s': other['graph_steps_per_s'],
                         'note': ('every rank advances its own 2048-graph batch; value = N x steps / max time' if modes[1] == 'weak'
                                  else 'the 2048-graph batch split over the ranks; value = steps / max time')}
    if args.time_all_kernels and timing['all']:
        names = {_lib.K_EDGE_MLP: 'edge_mlp', _lib.K_SCATTER_MEAN: 'scatter_mean', _lib.K_NODE_UPDATE: 'node_update', _lib.K_NORM: 'norm_blend',
                 _lib.K_LEM: 'lem_encoder', _lib.K_NODE_PROJ: 'node_project', _lib.K_DECODER: 'decoder'}
        out['kernels_ms_per_step'] = {nm: timing['all'][k][1] / timing['steps'] for k, nm in names.items()}
        out['kernels_launches_per_step'] = {nm: timing['all'][k][0] / timing['steps'] for k, nm in names.items()}
    if not args.no_extras:
        # row L2 standalone ("HBM GB/s on the scatter"): the default layer path fuses the mean into the message kernel,
        # so the CSR segmented-mean kernel is timed here on a message tensor of the workload's size, outside the timed region
        from msmp_pde_amd.graph import structure_of
        gs = structure_of(graph)
        msg = torch.randn(n_edges, H, device=dev)
        agg = torch.empty(n_nodes, H, device=dev)
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
                              'achieved_GBps': sc_bytes / (sc_ms * 1e-3) / 1e9, 'peak_GBps': PEAK_HBM_GBPS,
                              'frac': sc_bytes / (sc_ms * 1e-3) / 1e9 / PEAK_HBM_GBPS, 'algorithmic_bytes_per_launch': sc_bytes,
                              'avg_launch_ms': sc_ms}
        del msg, agg
        # SURVEY 8d "GPU timing": (ii) the forward alone and the graph construction (once per rollout, outside the timed region), both
        # timed here after the timed region on the workload's own batch
        wl_main = head_wl.parts[0] if isinstance(head_wl, SplitWorkload) else head_wl       # (sub-batches: the figures below are one sub-batch's)
        with torch.no_grad():
            for i in range(13):
                if i == 3:
                    torch.cuda.synchronize()
                    t_f = time.perf_counter()
                wl_main.model(wl_main.graph)
            torch.cuda.synchronize()
            fwd_ms = (time.perf_counter() - t_f) / 10 * 1e3
            steps0 = [50] * wl_main.bsz
            t_g = []
            for i in range(3):
                torch.cuda.synchronize()
                t0g = time.perf_counter()
                data_g, labels_g = wl_main.case.creator.create_data(wl_main.case.u_super, steps0)
                g_new = wl_main.case.creator.create_graph(data_g, labels_g, wl_main.case.x, wl_main.case.variables, steps0)
                gs_new = structure_of(g_new)
                gs_new.tiles()
                torch.cuda.synchronize()
                t_g.append((time.perf_counter() - t0g) * 1e3)
        out['breakdown'] = {'forward_only_ms': fwd_ms, 'state_update_ms': head['ms_per_step'] - fwd_ms,
                            'graph_construction_ms': min(t_g),
                            'note': 'forward_only = model(graph) alone; state_update = create_data labels + create_next_graph (the rest of a step); '
                                    'graph_construction = create_data + create_graph (radius / knn graph on the device) + CSR + node tiles, once per rollout, '
                                    'not part of a step'}
    if world == 1 and not args.no_extras:
        # BASELINE.json configs[0]: E2 MP-PDE on 32 graphs (the reference's own CPU-runnable case), HIP path and CPU port
        a0 = parse(['--model', 'MP-PDE', '--experiment', 'E2', '--graphs', '32'])
        wl0 = Workload(a0, mp, dev, 32, seed=1000)
        el0, _ = timed_run(wl0, D, torch, 50, 5, 0.2)
        out['config0'] = {'workload': 'E2 MP-PDE (MP_PDE_Solver), 32 graphs (BASELINE.json configs[0])',
                          'value': 50 / el0, 'unit': 'rollout-steps/s', 'ms_per_step': el0 / 50 * 1e3}
        if not args.no_cpu_baseline:
            out['config0']['cpu_baseline'] = cpu_baseline('MP-PDE', 'E2', 32, 32, 5)
        del wl0
    if not args.no_cpu_baseline and world == 1:      # rank 0 at N = 1 only
        out['cpu_baseline'] = cpu_baseline(args.model, exp, args.graphs, args.cpu_sample_graphs, args.cpu_sample_steps, args.neighbors)
    print(json.dumps(out), flush=True)
    return 0


def main():
    argv = sys.argv[1:]
    args = parse(argv)
    if args.gpus > 1 and not os.environ.get('WORLD_SIZE'):
        sys.exit(launch_children(args, argv))         # before any GPU call in this process
    rc = run_rank(args)
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            dist.destroy_process_group()
    except Exception:                                  # noqa: BLE001
        pass
    sys.exit(rc)


if __name__ == '__main__':
    main()
