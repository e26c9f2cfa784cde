"""msmp-pde_amd: the MI355X-native message-passing rollout path of MSMP-PDE.

Host-side mirror of the reference's module interface over the C-ABI library libmsmp_pde.so
(include/msmp_pde.h).  See DESIGN.md.  Import as `msmp_pde_amd` (alias module at the repo root).
"""
from ._lib import (lib, MsmpError, LIB_PATH, invalidate_packed_weights, last_status, MsmpRangeWarning,         # noqa: F401
                   MSMP_STATUS_INPUT_RANGE, MSMP_STATUS_NODE_SATURATED, MSMP_STATUS_NONFINITE)
from .pde import CE, WE, AD                                                   # noqa: F401
from .graph import Data, GraphCreator, GraphStructure, structure_of, radius_graph, knn_graph   # noqa: F401
from .layers import Swish, GNN_Layer, GNN_LayerLin, mp_layer                  # noqa: F401
from .lem import LEM, LEMS                                                          # noqa: F401
from . import optim                                                                 # noqa: F401  (optim.AdamW: fused HIP step)
from .solvers import (MP_PDE_Solver, MP_PDE_SolverGated, MP_PDE_SolverLEMLinGated, MP_PDE_Solver2D,   # noqa: F401
                      MP_PDE_Solver2DGated, MP_PDE_Solver2DLEMLinGated, MP_PDE_SolverLEMLin, MP_PDE_Solver2DLEMLin,
                      MP_PDE_Solver2DLEMLinG2, MSSMP_PDE_Solver, MSSMP_PDE_Solver_sub,
                      MP_PDE_SolverLEMLinGatedSave, MP_PDE_SolverLEMLinGatedGLU, MP_PDE_Solver2DLEMLinGatedGLU, MP_PDE_SolverLSTMLin, MP_PDE_SolverLSTMLinGated, MP_PDE_Solver2DLSTMLin, MP_PDE_Solver2DLSTMLinGated,
                      MODEL_NAMES)
