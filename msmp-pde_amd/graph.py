"""Graph side of the hot path: the `Data` container, device-side edge_index builders, the
CSR-by-target structure the kernels consume, and a vectorised mirror of the reference's GraphCreator.

Reference: common/utils.py:267-471 (GraphCreator), torch_cluster.radius_graph / knn_graph call sites
common/utils.py:368,377,380, torch_geometric.data.Data (:382).  SURVEY.md section 8a rows G1, G2, R1.
"""
import math

import torch

from . import _lib
from ._lib import lib, check, ptr, current_stream


import os as _os
_NO_SLICE = _os.environ.get('MSMP_NO_WINDOW_SLICE') == '1'       # A/B switch of create_data's zero-copy window (scripts only)


class Data(object):
    """Attribute bag standing in for torch_geometric.data.Data (common/utils.py:382-385):
    x [N,Tw], y, pos [N,2] (t, x), batch [N] int64, edge_index [2,E] int64, plus the per-experiment
    [N,1] parameter columns.  `.to(device)` moves every tensor attribute, like PyG's."""

    def __init__(self, x=None, edge_index=None, **kwargs):
        self.x = x
        self.edge_index = edge_index
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith('_')]

    def to(self, device):
        for k in list(self.__dict__):
            v = self.__dict__[k]
            if torch.is_tensor(v):
                self.__dict__[k] = v.to(device)
            elif isinstance(v, GraphStructure):
                self.__dict__[k] = None if torch.device(device) != v.rowptr.device else v
        return self


class GraphStructure(object):
    """What the kernels need of a batch of graphs, built once and reused over the rollout (the
    reference reuses edge_index and batch in create_next_graph, common/utils.py:431-471):
    CSR by target (rowptr [N+1], col [E] = source, tgt [E]) and graph_ptr [B+1], all int32 on device."""

    def __init__(self, edge_index, batch, n_nodes):
        dev = edge_index.device
        if dev.type != 'cuda':
            raise _lib.MsmpError('GraphStructure needs device tensors (HIP path only, no CPU fallback)')
        L = lib()
        self.n_nodes = int(n_nodes)
        self.n_edges = int(edge_index.shape[1])
        ei = edge_index.contiguous()
        if ei.dtype != torch.int64:
            ei = ei.long()
        counts = torch.bincount(batch, minlength=int(batch[-1].item()) + 1 if batch.numel() else 1)
        self.n_graphs = int(counts.numel())
        self.graph_ptr = torch.zeros(self.n_graphs + 1, dtype=torch.int32, device=dev)
        self.graph_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        self.max_graph_nodes = int(counts.max().item()) if counts.numel() else 0
        self._tgt_long = self._col_long = self._by_source = None
        self.rowptr = torch.empty(self.n_nodes + 1, dtype=torch.int32, device=dev)
        self.col = torch.empty(max(self.n_edges, 1), dtype=torch.int32, device=dev)
        self.tgt = torch.empty(max(self.n_edges, 1), dtype=torch.int32, device=dev)
        ws_bytes = L.msmp_build_csr_workspace_bytes(self.n_edges, self.n_nodes)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        check(L.msmp_build_csr(ptr(ei), self.n_edges, self.n_nodes, ptr(self.rowptr), ptr(self.col), ptr(self.tgt),
                               ptr(ws), ws_bytes, current_stream()), 'msmp_build_csr')
        # largest in-degree decides whether the fused message+mean kernel applies (one D2H read per structure)
        self.max_in_degree = int((self.rowptr[1:] - self.rowptr[:-1]).max().item()) if self.n_edges else 0
        self._key = (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version,
                     batch.data_ptr(), batch._version)
        self._tiles = False        # not built yet (None = the structure does not fit node tiles)
        self._period = False

    def period(self):
        """(nodes, edges) of one graph when the batch is n_graphs copies of ONE pattern -- what the reference's GraphCreator builds:
        every sample of a dataset lives on the same grid (common/utils.py:364-380) -- else None.  Two device comparisons and one
        read-back per structure."""
        if self._period is not False:
            return self._period
        self._period = None
        b = self.n_graphs
        if b >= 2 and self.n_edges and self.n_nodes % b == 0 and self.n_edges % b == 0 and self.max_graph_nodes * b == self.n_nodes:
            nx, eg = self.n_nodes // b, self.n_edges // b
            dev = self.rowptr.device
            k = torch.arange(b, dtype=torch.int32, device=dev)[:, None]
            same = ((self.col[:self.n_edges].view(b, eg) - k * nx == self.col[:eg]).all()
                    & (self.rowptr[:-1].view(b, nx) - k * eg == self.rowptr[:nx]).all())
            if bool(same):
                self._period = (nx, eg)
        return self._period

    def tiles(self):
        """Node tiles of the LDS-staged message kernel (msmp_tiles_t; include/msmp_pde.h), built once per structure; None when
        the graph does not tile (in-degrees above 32: a target's in-edges must fit the 32 lanes of one wave; or a tile of target
        nodes would touch more than MSMP_TILE_NCAP distinct nodes unless it became too small to pay off): the callers then take
        the gather kernels.  group_nodes: as many consecutive targets as fit a wave's 32 edge lanes at the largest in-degree (at
        most 8: four groups share the 32 node slots), then shrunk while the tiles' node lists do not fit (banded graphs have a
        near-constant halo: a few retries, one device read-back each).
        A batch of identical graphs (`period()`) whose size the tile divides gets a PERIODIC descriptor: the tiles of ONE graph
        (a few KB that stay in the scalar / L2 caches instead of 5 MB of per-tile metadata streamed from HBM per launch), no tile
        straddles two graphs, and a graph's result does not depend on its position in the batch (bitwise graph-order equivariance
        and sharding).  Where the tile does not divide the graph (knn configs: 28 | 22 nodes per tile, 100 per graph) the tiles
        straddle graph boundaries by default -- cutting them costs 11-20 % more tiles -- unless msmp_tune("tile_align", 1)."""
        if self._tiles is not False:
            return self._tiles
        self._tiles = None
        if self.n_edges == 0 or self.max_in_degree <= 0 or self.max_in_degree > _lib.MSMP_TILE_GROUP_EDGES:
            return None
        L = lib()
        dev = self.rowptr.device
        gn0 = min(_lib.MSMP_TILE_NCAP // 4, _lib.MSMP_TILE_GROUP_EDGES // self.max_in_degree)
        gn = gn0
        per = self.period()
        align = bool(L.msmp_tune_query(b'tile_align'))
        for _attempt in range(4):
            if gn < max(1, gn0 // 2):           # tiles this small waste the 128-lane block: not worth it
                return None
            tn = 4 * gn
            periodic = per is not None and (per[0] % tn == 0 or align)
            n_src, e_src = (per if periodic else (self.n_nodes, self.n_edges))
            n_meta = (n_src + tn - 1) // tn          # tiles the descriptor arrays describe
            n_tiles = n_meta * (self.n_graphs if periodic else 1)
            if periodic and n_tiles * n_meta >= 1 << 32:
                periodic, n_src, e_src = False, self.n_nodes, self.n_edges
                n_meta = n_tiles = (n_src + tn - 1) // tn
            tile_node = torch.empty(n_meta * _lib.MSMP_TILE_NCAP, dtype=torch.int32, device=dev)
            tile_count = torch.empty(n_meta, dtype=torch.int32, device=dev)
            tile_halo = torch.empty(n_meta * 4, dtype=torch.int32, device=dev)
            edge_slot = torch.empty(n_meta * _lib.MSMP_TILE_EDGES, dtype=torch.int32, device=dev)
            stats = torch.empty(3, dtype=torch.int32, device=dev)
            # a periodic descriptor is built from the first graph's CSR: rowptr[:nx + 1] / col[:eg] are that graph's, in its own node ids
            check(L.msmp_build_tiles(ptr(self.rowptr), ptr(self.col), n_src, e_src, gn, ptr(tile_node), ptr(tile_count),
                                     ptr(tile_halo), ptr(edge_slot), ptr(stats), current_stream()), 'msmp_build_tiles')
            max_nodes, max_edges, n_listed = (int(v) for v in stats.tolist())
            if max_nodes <= _lib.MSMP_TILE_NCAP and max_edges <= _lib.MSMP_TILE_GROUP_EDGES:
                desc = _lib.MsmpTiles(tn, gn, n_tiles, ptr(tile_node), ptr(tile_count), ptr(tile_halo), ptr(edge_slot), int(n_listed > 0),
                                      n_meta if periodic else 0, n_src if periodic else 0)
                self._tiles = (desc, tile_node, tile_count, edge_slot, tile_halo)      # the tensors keep the descriptor's memory alive
                return self._tiles
            gn -= max(1, -(-(max_nodes - _lib.MSMP_TILE_NCAP) // 4)) if max_nodes > _lib.MSMP_TILE_NCAP else 1
        return None

    @property
    def tgt_long(self):
        """int64 copies of the edge endpoints (torch index ops of the training backward), made once."""
        if self._tgt_long is None:
            self._tgt_long, self._col_long = self.tgt[:self.n_edges].long(), self.col[:self.n_edges].long()
        return self._tgt_long

    @property
    def col_long(self):
        self.tgt_long
        return self._col_long

    def by_source(self):
        """(perm, rowptr): the edges regrouped by SOURCE node (stable), for means over a node's out-edges
        (msmp_scatter_mean_f32 on msg[perm]); made once."""
        if self._by_source is None:
            src = self.col_long
            perm = torch.sort(src, stable=True)[1]
            rowptr = torch.zeros(self.n_nodes + 1, dtype=torch.int32, device=src.device)
            rowptr[1:] = torch.cumsum(torch.bincount(src, minlength=self.n_nodes), 0).to(torch.int32)
            self._by_source = (perm, rowptr)
        return self._by_source

    def by_source32(self):
        """(perm int32, rowptr int32) of by_source() for the kernels (msmp_mp_layer_bwd_f32's deterministic scatter); made once."""
        if getattr(self, '_by_source32', None) is None:
            perm, rowptr = self.by_source()
            self._by_source32 = (perm.to(torch.int32).contiguous(), rowptr)
        return self._by_source32

    def matches(self, edge_index, batch):
        return self._key == (edge_index.data_ptr(), tuple(edge_index.shape), edge_index._version,
                             batch.data_ptr(), batch._version)


def structure_of(data):
    """GraphStructure of a Data object, cached on it (rebuilt if edge_index / batch changed)."""
    gs = getattr(data, '_msmp_structure', None)
    if gs is None or not gs.matches(data.edge_index, data.batch):
        gs = GraphStructure(data.edge_index, data.batch, data.x.shape[0])
        data._msmp_structure = gs
    return gs


def _graph_ptr_from_sizes(sizes, device):
    gp = torch.zeros(len(sizes) + 1, dtype=torch.int32)
    gp[1:] = torch.cumsum(torch.as_tensor(sizes, dtype=torch.int64), 0).to(torch.int32)
    return gp.to(device)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, sizes=None):
    """Device-side torch_cluster.radius_graph (call site common/utils.py:368).  x: [N] or [N,dim]
    float64 on the GPU; `batch` sorted ascending (or `sizes` = nodes per graph).  Returns edge_index
    [2,E] int64 in canonical order: ascending target, then ascending source.  Bit-exact contract:
    pair kept iff (x_i-x_j)^2 summed over dims in float64 < r*r."""
    assert not loop
    x2 = x.reshape(x.shape[0], -1).to(torch.float64).contiguous()
    dev = x2.device
    n = x2.shape[0]
    if sizes is None:
        sizes = torch.bincount(batch).tolist() if batch is not None else [n]
    gp = _graph_ptr_from_sizes(sizes, dev)
    L = lib()
    st = current_stream()
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    check(L.msmp_radius_graph_count_f64(ptr(x2), x2.shape[1], ptr(gp), len(sizes), n, float(r), int(max_num_neighbors),
                                        ptr(rowptr), st), 'msmp_radius_graph_count_f64')
    e = int(rowptr[-1].item())
    ei = torch.empty((2, e), dtype=torch.int64, device=dev)
    check(L.msmp_radius_graph_fill_f64(ptr(x2), x2.shape[1], ptr(gp), len(sizes), n, float(r), int(max_num_neighbors),
                                       ptr(rowptr), e, ptr(ei), st), 'msmp_radius_graph_fill_f64')
    return ei


def knn_graph(x, k, batch=None, loop=False, sizes=None):
    """Device-side torch_cluster.knn_graph (call sites common/utils.py:377,380): per target its k nearest
    same-graph nodes, ascending float64 squared distance, ties -> lower index."""
    assert not loop
    x2 = x.reshape(x.shape[0], -1).to(torch.float64).contiguous()
    dev = x2.device
    n = x2.shape[0]
    if sizes is None:
        sizes = torch.bincount(batch).tolist() if batch is not None else [n]
    gp = _graph_ptr_from_sizes(sizes, dev)
    e = sum(s * min(k, s - 1) for s in sizes)
    L = lib()
    rowptr = torch.empty(n + 1, dtype=torch.int32, device=dev)
    ei = torch.empty((2, e), dtype=torch.int64, device=dev)
    check(L.msmp_knn_graph_f64(ptr(x2), x2.shape[1], ptr(gp), len(sizes), n, int(k), e, ptr(rowptr), ptr(ei),
                               current_stream()), 'msmp_knn_graph_f64')
    return ei


def torch_time_axis(pde):
    """t = torch.linspace(tmin, tmax, nt) in float64 on the CPU, exactly the values the reference
    builds at common/utils.py:340,456 (ATen's CPU linspace; computed there so pos[:,0] is bit-identical)."""
    return torch.linspace(pde.tmin, pde.tmax, pde.grid_size[0], dtype=torch.float64)


class GraphCreator(object):
    """Vectorised mirror of common/utils.py:267-471.  Same constructor and methods; the per-node and
    per-sample Python `torch.cat` loops (utils.py:349-362, 390-426, 459-467) become index arithmetic, the
    tensors are assembled directly on `device`, and edge_index comes from the device builders above
    (or is passed in / cached: the grid is the same for every batch)."""

    def __init__(self, pde, neighbors=2, time_window=5, t_resolution=250, x_resolution=100, device=None):
        assert isinstance(neighbors, int) and isinstance(time_window, int)
        self.pde = pde
        self.n = neighbors
        self.tw = time_window
        self.t_res = t_resolution
        self.x_res = x_resolution
        self.device = device
        self._edge_cache = {}
        self._small = {}      # device-resident index helpers (step lists, window offsets, the time axis), see _cached

    def _cached(self, key, make):
        """Small tensors that repeat from one rollout step to the next (the reference rebuilds them on the host every call):
        keeping them on the device removes a handful of H2D copies and tiny kernels per step -- a 32-graph step is ~60
        dependent launches, so each one counts there."""
        v = self._small.get(key)
        if v is None:
            if len(self._small) > 256:
                self._small.clear()
            v = self._small[key] = make()
        return v

    def _steps_on(self, steps, device):
        if torch.is_tensor(steps):
            return steps.to(device=device, dtype=torch.long)
        steps = tuple(int(s) for s in steps)
        return self._cached(('steps', steps, str(device)), lambda: torch.tensor(steps, dtype=torch.long, device=device))

    def _time_axis_on(self, device):
        p = self.pde
        return self._cached(('t', float(p.tmin), float(p.tmax), int(p.grid_size[0]), str(device)),
                            lambda: torch_time_axis(p).to(device))

    def to(self, device):
        self.device = device
        return self

    # -- common/utils.py:300-317
    def create_data(self, datapoints, steps):
        dev, tw = datapoints.device, self.tw
        if not torch.is_tensor(steps) and len(steps) > 0 and all(int(s) == int(steps[0]) for s in steps) and not _NO_SLICE:
            # every sample at the same step (the rollout loops of the reference, train_helper.py:255-273 `same_steps`): the window is a
            # plain slice of the trajectory tensor -- views, no gather kernel (a 41-MB indexed copy per rollout step at 2048 graphs)
            s0 = int(steps[0])
            if tw <= s0 <= datapoints.shape[1] - tw:
                return datapoints[:, s0 - tw:s0], datapoints[:, s0:s0 + tw]
        steps_t = self._steps_on(steps, dev)
        b = self._cached(('b', datapoints.shape[0], str(dev)), lambda: torch.arange(datapoints.shape[0], device=dev)[:, None])
        win = self._cached(('win', tw, str(dev)), lambda: torch.arange(-tw, tw, device=dev)[None, :])
        block = datapoints[b, steps_t[:, None] + win]         # one gather: [step - tw, step) is the data, [step, step + tw) the labels
        return block[:, :tw], block[:, tw:]

    def _flatten(self, block):
        """[B,tw,nx] -> [B*nx, tw]; AD: [B,tw,2,nx] -> [B*nx, 2*tw] component-major (utils.py:350-357)."""
        if f'{self.pde}' == 'AD':
            b, tw, c, nx = block.shape
            return block.permute(0, 3, 2, 1).reshape(b * nx, c * tw)
        b, tw, nx = block.shape
        return block.permute(0, 2, 1).reshape(b * nx, tw)

    def build_edge_index(self, x0, bsz, device):
        """Row G1.  x0: the [nx] float64 grid of one sample (the reference uses x[0] for all, utils.py:359,366)."""
        name = f'{self.pde}'
        nx = x0.shape[0]
        x0c = x0.detach().to('cpu', torch.float64).contiguous()
        key = (name, self.n, bsz, nx, str(device), bool(getattr(self.pde, 'untructured_grid', False)),
               hash(x0c.numpy().tobytes()))
        if key in self._edge_cache:
            return self._edge_cache[key]
        sizes = [nx] * bsz
        if name in ('CE', 'KF', 'KS', 'AD'):
            if name == 'AD' and getattr(self.pde, 'untructured_grid', False):
                # utils.py:343-346: periodic embedding, evaluated on the CPU in float64 like the reference
                xx = 2 * math.pi * x0c / (torch.max(x0c) - 1e-3)
                x_per = torch.stack([torch.cos(xx), torch.sin(xx)], 1)
                ei = knn_graph(x_per.repeat(bsz, 1).to(device), self.n, sizes=sizes)
            else:
                dx = float(x0c[1] - x0c[0])
                ei = radius_graph(x0c.repeat(bsz).to(device), self.n * dx + 0.0001, sizes=sizes)
        elif name == 'WE':
            ei = knn_graph(x0c.repeat(bsz).to(device), self.n, sizes=sizes)
        else:
            raise ValueError(f'unknown pde {name}')
        self._edge_cache = {key: ei}
        return ei

    # -- common/utils.py:320-428
    def create_graph(self, data, labels, x, variables, steps, edge_index=None):
        device = self.device if self.device is not None else data.device
        nt, nx = self.pde.grid_size[0], self.pde.grid_size[1]
        bsz = data.shape[0]
        t = torch_time_axis(self.pde)
        steps_t = torch.as_tensor(steps, dtype=torch.long)
        u = self._flatten(data.to(device))
        y = self._flatten(labels.to(device))
        x0 = x[0]
        x_pos = x0.to(device).repeat(bsz)
        t_pos = t[steps_t].to(device).repeat_interleave(nx)
        batch = torch.arange(bsz, device=device).repeat_interleave(nx)
        if edge_index is None:
            edge_index = self.build_edge_index(x0, bsz, device)
        graph = Data(x=u, edge_index=edge_index)
        graph.y = y
        graph.pos = torch.cat((t_pos[:, None].to(x_pos.dtype), x_pos[:, None]), 1)
        graph.batch = batch

        def col(name, sign=1.0):
            v = torch.as_tensor(variables[name]).to(device)
            return (sign * v)[batch][:, None]

        name = f'{self.pde}'
        if name == 'CE':           # utils.py:388-397; beta is stored negated (:392)
            graph.alpha, graph.beta, graph.gamma = col('alpha'), col('beta', -1.0), col('gamma')
        elif name == 'KF':
            graph.r, graph.D = col('r'), col('D')
        elif name == 'WE':
            graph.bc_left, graph.bc_right, graph.c = col('bc_left'), col('bc_right'), col('c')
        elif name == 'AD':
            graph.a, graph.b = col('a'), col('b')
        return graph

    # -- common/utils.py:431-471
    def create_next_graph(self, graph, pred, labels, steps):
        keep = 2 * self.tw if f'{self.pde}' == 'AD' else self.tw
        if pred.shape[1] == keep and graph.x.shape[1] == keep:
            graph.x = pred.to(graph.x.dtype)         # cat(x, pred)[:, keep:] == pred
        else:
            graph.x = torch.cat((graph.x, pred.to(graph.x.dtype)), 1)[:, keep:]
        nx = self.pde.grid_size[1]
        device = graph.x.device
        graph.y = self._flatten(labels.to(device))
        t_now = self._time_axis_on(device)[self._steps_on(steps, device)]          # float64 values of the CPU linspace
        if graph.pos.is_contiguous():
            graph.pos.view(-1, nx, graph.pos.shape[1])[:, :, 0] = t_now[:, None].to(graph.pos.dtype)
        else:
            graph.pos[:, 0] = t_now.repeat_interleave(nx).to(graph.pos.dtype)
        return graph
