"""Host mirror of the model-load consumer of the Gramian kernel:
the solver part of net.myrrix.online.generation.Generation
(online/src/net/myrrix/online/generation/Generation.java:68-70,132-158,181-205) and the
net.myrrix.common.math.Solver it hands to the serving layer's fold-in
(ServerRecommender.java:570-592,877-885).  Same names, argument meaning and error behaviour; the
Gramian runs on the GPU (K1), the k x k factorization on the host in fp64, both behind the C-ABI.
Candidate filters, known-item maps, clusters and locks of Generation are serving-plane state and
out of scope."""
import numpy as np

from . import _lib
from .core import ALSCore, HostSolver, IllConditioned, SingularSystem
from .factorizer import SingularMatrixSolverException, SolverException, System


class IllConditionedSolverException(SolverException):
    """net.myrrix.common.math.IllConditionedSolverException"""


class Solver:
    """net.myrrix.common.math.Solver (Solver.java:27-43)."""

    def __init__(self, host_solver):
        self._s = host_solver

    def solveDToF(self, b):
        return self._s.solve_dtof(b)

    def solveFToD(self, b):
        return self._s.solve_ftod(b)


def _threshold():
    return float(System.getProperty("common.matrix.singularityThreshold", 1.0e-5))  # LSS:33-34


def getSolver(M):
    """MatrixUtils.getSolver (MU:137-139, CMLSS:37-55): M is a square array or None."""
    if M is None:
        return None
    try:
        return Solver(HostSolver.create(np.asarray(M, dtype=np.float64), _threshold()))
    except SingularSystem as e:
        raise SingularMatrixSolverException(e.apparent_rank, e.message)


def isNonSingular(M):
    """MatrixUtils.isNonSingular (MU:130-132, CMLSS:58-63)."""
    try:
        HostSolver.create(np.asarray(M, dtype=np.float64), _threshold()).close()
        return True
    except SingularSystem:
        return False


class Generation:
    """X, Y: dict id -> float32[k] (the reference's FastByIDMap<float[]>)."""

    def __init__(self, X, Y, device=0):
        self.X, self.Y = X, Y
        self.device = device
        self.XTXsolver = None
        self.YTYsolver = None
        self.recomputeState()

    def recomputeState(self):
        """Generation.java:132-140."""
        if str(System.getProperty("model.solver.xtx.compute", "true")).lower() == "true":
            self.XTXsolver = self._recomputeSolver(self.X)
        if str(System.getProperty("model.solver.yty.compute", "true")).lower() == "true":
            self.YTYsolver = self._recomputeSolver(self.Y)

    def _recomputeSolver(self, M):
        """Generation.java:142-158."""
        if M is None or len(M) == 0:
            return None
        rows = np.stack([np.asarray(v, dtype=np.float32) for v in M.values()])
        with ALSCore(rows.shape[1], device=self.device, singularity_threshold=_threshold()) as core:
            core.set_factor_rows(_lib.SIDE_X, rows.shape[0])
            core.set_factors(_lib.SIDE_X, rows)
            try:
                solver, _ = core.recompute_solver(_lib.SIDE_X)
            except IllConditioned as e:
                raise IllConditionedSolverException("infNorm: %r" % e.inf_norm)
            except SingularSystem as e:
                raise SingularMatrixSolverException(e.apparent_rank, e.message)
        return Solver(solver)

    def getXTXSolver(self):
        return self.XTXsolver

    def getYTYSolver(self):
        return self.YTYsolver

    def getNumUsers(self):
        return len(self.X)

    def getNumItems(self):
        return len(self.Y)

    def getX(self):
        return self.X

    def getY(self):
        return self.Y
