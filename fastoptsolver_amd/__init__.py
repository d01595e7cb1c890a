"""fastoptsolver_amd — MI355X-native inner loops (FISTA / ISTA / FISTA-Δ / L-BFGS) behind the call signatures
of ElBaldo1/FastOptSolver.  Hand-written HIP for gfx950 through a C ABI (include/fos.h); no CPU fallback."""
from ._core import Problem, prepare                                   # noqa: F401
from ._lib import FosError                                            # noqa: F401
from .iterative_solvers import (estimate_lipschitz, fista, fista_delta, fista_path, get_metrics,  # noqa: F401
                                ista, reset_metrics)
from .lbfgs import LBFGSSolver                                        # noqa: F401
from .objective_functions import compute_objective                    # noqa: F401
from .operators import ElasticNetProx, L1Prox, LeastSquares           # noqa: F401
from .prox_operators import prox_elastic_net, prox_l1                 # noqa: F401

__all__ = ["fista", "fista_delta", "fista_path", "ista", "estimate_lipschitz", "reset_metrics", "get_metrics", "LBFGSSolver",
           "compute_objective", "prox_l1", "prox_elastic_net", "LeastSquares", "L1Prox", "ElasticNetProx",
           "prepare", "Problem", "FosError"]
