"""Minimal PDE descriptors for the hot path.  The solver classes and GraphCreator read only
`L`, `tmin`, `tmax`, `dt`, `grid_size`, `repr()` and (AD) `untructured_grid` from the reference's PDE
objects (equations/PDEs.py:37-98, 150-200, 246-300), at call time.  The
reference's own objects can be passed instead: duck typing."""


class _PDE(object):
    name = 'PDE'

    def __init__(self, tmin=0.0, tmax=1.0, grid_size=(250, 100), L=16.0):
        self.tmin = tmin
        self.tmax = tmax
        self.grid_size = list(grid_size)
        self.L = L
        self.dt = self.tmax / (self.grid_size[0] - 1)

    def __repr__(self):
        return self.name


class CE(_PDE):
    """Combined equation (E1-E3); equations/PDEs.py:37-98.  L = 16."""
    name = 'CE'


class WE(_PDE):
    """Wave equation (WE1-3); equations/PDEs.py:150-200.  L = |xmax - xmin| = 16."""
    name = 'WE'

    def __init__(self, tmin=0.0, tmax=20.0, grid_size=(250, 100), xmin=-8.0, xmax=8.0):
        super().__init__(tmin, tmax, grid_size, abs(xmax - xmin))
        self.xmin, self.xmax = xmin, xmax


class AD(_PDE):
    """Linear advection system, two components (RP, RPU, MSWG, MSWG3); equations/PDEs.py:246-300."""
    name = 'AD'

    def __init__(self, tmin=0.0, tmax=0.5, grid_size=(250, 100), L=16.0, unstructured=False):
        super().__init__(tmin, tmax, grid_size, L)
        self.untructured_grid = unstructured    # (sic) the reference's attribute name, equations/PDEs.py:296
