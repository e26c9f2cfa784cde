"""Training step of the reference under graph-sharded data parallelism.

Reference (single device): experiments/train_helper.py:91-141 -- pushforward unrolling under no_grad,
then `loss = sqrt(MSELoss(reduction='sum')(pred, graph.y)); loss.backward(); optimizer.step()`.
The loss is sqrt(S) with S = sum over ALL nodes of the batch, which is not a mean of per-shard losses, so
exact data parallelism is:  (1) all-reduce(SUM) the scalar S_k of every rank;  (2) backward S_k scaled by
1/(2 sqrt(S));  (3) all-reduce(SUM) the gradients (one flat ~5 MB buffer: 1.25-1.34 M fp32 parameters),
RCCL over xGMI when the backend is nccl.  Parameters are replicated; every rank must draw the same
`unrolled_graphs` and its own slice of `random_steps`."""
import torch
import torch.distributed as dist

from .reductions import sqerr_sum


def _world():
    """Number of ranks; 0 when there is no process group at all (an initialised ONE-rank group still runs its collectives:
    dist.init_from_env(force_group=True) exercises the RCCL path on a single GPU)."""
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 0


def allreduce_gradients(params):
    """SUM-all-reduce the gradients of `params` as one flat buffer (a single collective per step)."""
    params = [p for p in params if p.grad is not None]
    if _world() == 0 or not params:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in params])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    off = 0
    for p in params:
        n = p.grad.numel()
        p.grad.copy_(flat[off:off + n].view_as(p.grad))
        off += n


def dp_loss_backward(model, graph):
    """Forward + backward of the reference's loss on this rank's shard; gradients are left all-reduced
    (identical on every rank and equal to the single-device gradient).  Returns the global loss sqrt(S)."""
    pred = model(graph)
    s_local = sqerr_sum(pred, graph.y)                      # MSELoss(reduction='sum'), deterministic two-launch kernel on the GPU
    s_total = s_local.detach().clone()
    if _world() > 0:
        dist.all_reduce(s_total, op=dist.ReduceOp.SUM)
    loss = torch.sqrt(s_total)
    (s_local / (2.0 * loss)).backward()
    allreduce_gradients(list(model.parameters()))
    return loss


def training_step(model, creator, u_super, x, variables, random_steps, unrolled_graphs, optimizer, captured=None):
    """One iteration of training_loop (experiments/train_helper.py:91-141) on this rank's samples.
    captured: a CapturedTrainStep built on a batch of the same structure (same grid, batch size and neighbourhood): forward + loss +
    backward + optimizer update are then ONE hipGraph launch; the pushforward unrolling stays eager (inference kernels)."""
    if captured is None:
        optimizer.zero_grad()
    steps = list(random_steps)
    data, labels = creator.create_data(u_super, steps)
    graph = creator.create_graph(data, labels, x, variables, steps)
    with torch.no_grad():                                   # pushforward trick, :106-112
        for _ in range(unrolled_graphs):
            steps = [s + creator.tw for s in steps]
            _, labels = creator.create_data(u_super, steps)
            pred = model(graph)
            graph = creator.create_next_graph(graph, pred, labels, steps)
    if captured is not None:
        loss = captured(graph)
    else:
        loss = dp_loss_backward(model, graph)
        optimizer.step()
    lem = getattr(model, 'embedding_lem', None)            # the Save variants start every sample from fresh LEM states (:144-145)
    if hasattr(lem, 'reset_states'):
        lem.reset_states()
    return loss


class CapturedTrainStep:
    """hipGraph of ONE optimisation step on a fixed batch structure: forward, loss, backward and the optimizer update
    (`dp_loss_backward` + `optimizer.step()`, experiments/train_helper.py:125-141) replayed as graph launches.

    The reference's batch of 16 graphs is ~500 kernel launches of a few microseconds each: eager, the iteration is bound by
    the host issuing them.  What makes the capture safe here:
      * every kernel of the step is stream-ordered and takes no host read-back: layers / LEM forward and backward are the library's
        kernels, the bias gradients of the encoder / decoder and the loss sum are its deterministic reductions (reductions.py: no
        library reduction, hence no memset node in the graph), AdamW runs with the step count and learning rate in device memory
        (optim.AdamW(capturable=True));
      * everything the captured launches point at is owned here: static copies of the batch tensors, private layer / backward
        workspaces, the gradients and the packed weight blobs (allocated from the graph's pool during capture), and references to
        the optimizer's moment tensors and device-side step / learning-rate words (checked at every call: see __call__);
      * the weights are re-packed INSIDE the graph (the packed-blob caches are invalidated right before the capture, so every
        replay packs the parameters it is about to use) and the caches are invalidated again after each replay, so an eager
        forward in between (validation) packs the updated parameters.
    The batch STRUCTURE (edge_index, batch vector, node count) is that of `graph` at construction; `__call__(graph)` copies the
    floating-point node tensors (x, y, pos, equation-parameter columns) of another batch of the same structure and replays.

    Building the capture does NOT train: the `warmup` eager iterations it needs (optimizer state, workspaces, structure caches) run
    on the construction batch, and parameters, Adam moments and the step count are put back afterwards, in place -- the
    first call is the first optimisation step, as in the reference's loop.

    Data parallel (world size > 1): the step is THREE graphs with the two collectives of `dp_loss_backward`
    between them, in the same order and arithmetic as the eager path (so the captured trajectory equals the eager DP trajectory):
        graph F: forward + this rank's S_k;   all-reduce(SUM) of the scalar S;
        graph B: loss = sqrt(S), backward of S_k / (2 loss);   ONE flat all-reduce(SUM) of the gradients (RCCL over xGMI);
        graph U: the AdamW update.
    Without a process group (or with a one-rank group) it is ONE graph."""

    def __init__(self, model, optimizer, graph, warmup=3):
        import copy
        from . import autograd as _ag
        from .layers import _Workspace
        from ._lib import invalidate_packed_weights
        if not all(g.get('capturable') for g in optimizer.param_groups):
            raise RuntimeError('CapturedTrainStep needs msmp_pde_amd.optim.AdamW(..., capturable=True)')
        self.model, self.opt = model, optimizer
        self.data = copy.copy(graph)
        self._float_keys = [k for k in graph.keys() if torch.is_tensor(getattr(graph, k)) and getattr(graph, k).is_floating_point()]
        for k in self._float_keys:
            setattr(self.data, k, getattr(graph, k).clone())
        dev = self.data.x.device
        self._invalidate = invalidate_packed_weights
        self._dp = _world() > 1
        params = list(model.parameters())
        old_bwd = _ag._bwd_ws.pop(dev, None)              # the backward's grow-only scratch: a private one for the graph
        try:
            with _Workspace.private(dev) as ws:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                # what the warm-up iterations change: put back afterwards (in place: the captured launches keep pointing at it)
                p0 = [p.detach().clone() for p in params]
                had_state = {id(p): bool(optimizer.state.get(p)) for p in params}
                s0 = {id(p): {k: v.detach().clone() for k, v in optimizer.state[p].items() if torch.is_tensor(v)} for p in params if had_state[id(p)]}
                def _count(g):          # the group's step count before the warm-up: the device word, or (no step taken yet on this optimizer object) the loaded state's
                    if g.get('_msmp_dev'):
                        return int(g['_msmp_dev']['step'].item())
                    ts = [int(optimizer.state[p]['step'].item()) for p in g['params'] if optimizer.state.get(p)]
                    return ts[0] if ts else 0
                d0 = [_count(g) for g in optimizer.param_groups]
                with torch.cuda.stream(side):             # eager warm-up: optimizer state, workspaces, structure caches
                    for _ in range(warmup):
                        optimizer.zero_grad(set_to_none=True)
                        dp_loss_backward(model, self.data)
                        optimizer.step()
                torch.cuda.current_stream().wait_stream(side)
                torch.cuda.synchronize()
                with torch.no_grad():
                    for p, q in zip(params, p0):
                        p.copy_(q)
                    for p in params:
                        st = optimizer.state.get(p)
                        if not st:
                            continue
                        for k, v in st.items():
                            if torch.is_tensor(v):
                                v.copy_(s0[id(p)][k]) if had_state[id(p)] else v.zero_()
                    for g, d in zip(optimizer.param_groups, d0):
                        if g.get('_msmp_dev'):
                            g['_msmp_dev']['step'].fill_(d)
                optimizer.zero_grad(set_to_none=True)
                invalidate_packed_weights()               # the pack kernels must be part of the graph
                if not self._dp:
                    self.graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self.graph):
                        self.loss = dp_loss_backward(model, self.data)
                        optimizer.step()
                    self._graphs = [self.graph]
                else:
                    gf, gb, gu = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    cap = torch.cuda.Stream()             # ONE capture stream: autograd runs a node's backward on its forward's stream
                    with torch.cuda.graph(gf, stream=cap):
                        pred = model(self.data)
                        self._s_local = sqerr_sum(pred, self.data.y)
                    self._s_total = self._s_local.detach().clone()
                    with torch.cuda.graph(gb, pool=gf.pool(), stream=cap):
                        self.loss = torch.sqrt(self._s_total)
                        (self._s_local / (2.0 * self.loss)).backward()
                    with torch.cuda.graph(gu, pool=gf.pool(), stream=cap):
                        optimizer.step()
                    self._graphs = [gf, gb, gu]
                torch.cuda.synchronize()
            self._ws = ws.buffers()
            self._bwd_ws = _ag._bwd_ws.pop(dev, None)     # pinned: the captured backward points into it
        finally:
            if old_bwd is not None:
                _ag._bwd_ws[dev] = old_bwd
        self._blobs = [getattr(m, a) for m in model.modules() for a in ('_packed', '_embed_blob') if getattr(m, a, None) is not None]
        self._params = [(p, p.data_ptr()) for p in params]
        # the optimizer-side memory the graph points at: moments, device step count and learning rate (optimizer.load_state_dict
        # replaces the group dicts and the state tensors; a replay would then read and write freed memory)
        self._opt_dev = [g.get('_msmp_dev') for g in optimizer.param_groups]      # by POSITION: load_state_dict installs new group dicts
        self._opt_state = [(p, optimizer.state[p]['exp_avg'], optimizer.state[p]['exp_avg_sq']) for p in params if optimizer.state.get(p)]
        invalidate_packed_weights()                       # the capture-time hook saw no replay: eager users re-pack

    def _check_pointers(self):
        for p, ptr0 in self._params:
            if p.data_ptr() != ptr0:
                raise RuntimeError('CapturedTrainStep: a parameter was re-allocated (model.to(), load_state_dict with assign): capture again')
        groups = self.opt.param_groups
        for i, dev in enumerate(self._opt_dev):
            if i >= len(groups) or groups[i].get('_msmp_dev') is not dev:
                raise RuntimeError('CapturedTrainStep: the optimizer\'s device-side step / learning-rate state was replaced '
                                   '(optimizer.load_state_dict): capture again')
        for p, m, v in self._opt_state:
            st = self.opt.state.get(p)
            if not st or st['exp_avg'] is not m or st['exp_avg_sq'] is not v:
                raise RuntimeError('CapturedTrainStep: the optimizer state tensors were replaced (optimizer.load_state_dict): capture again')

    def __call__(self, graph):
        self._check_pointers()
        for k in self._float_keys:
            getattr(self.data, k).copy_(getattr(graph, k))
        self.opt.sync_lr()                                # a scheduler may have changed param_groups[i]["lr"]
        if not self._dp:
            self.graph.replay()
        else:
            gf, gb, gu = self._graphs
            gf.replay()
            self._s_total.copy_(self._s_local.detach())
            dist.all_reduce(self._s_total, op=dist.ReduceOp.SUM)
            gb.replay()
            allreduce_gradients([p for p, _ in self._params])
            gu.replay()
        self._invalidate()
        return self.loss.clone()
