"""myrrix-recommender_amd: MI355X-native ALS matrix-factorization core behind myrrix-recommender's
MatrixFactorizer / AlternatingLeastSquares surface.

  include/myrrix_als.h          the C-ABI (drop-in boundary)
  csrc/                         hand-written gfx950 kernels + host library (libmyrrix_als.so)
  core.ALSCore                  one handle = one GPU (plumbing over the C-ABI)
  factorizer                    host mirror of the reference interface (same names / errors)
  generation                    host mirror of Generation.recomputeSolver + Solver (8(f) row 1)
  serializer                    model.bin.gz reader / writer, mirror of GenerationSerializer (8(f) row 5)
  ingest                        records -> CSR on the device, mirror of InputFilesReader's record loop (8(f) row 2)
  sharded.ShardedALS            one process per GPU, torch.distributed (RCCL) all-gather between
                                half-iterations
  synth                         seeded synthetic interaction matrices

Importing the package does not need a GPU; creating an ALSCore does, and fails loudly without
one.  There is no CPU fallback anywhere in this package.
"""
from . import _lib
from .core import ALSCore, Cancelled, HostSolver, IllConditioned, MalsError, SingularSystem
from .factorizer import (AlternatingLeastSquares, ExecutionException, InterruptedException,
                         MatrixFactorizer, MatrixUtils, SingularMatrixSolverException,
                         SolverException, System)
from .group import GroupALS, plan_shards
from .generation import Generation, IllConditionedSolverException, Solver
from .ingest import Ingest, readInputRecords
from .serializer import GenerationSerializer, SerializedGeneration
from ._lib import (FLAG_LOSS_IGNORES_UNSPECIFIED, FLAG_RECONSTRUCT_R, SIDE_X, SIDE_Y)

__all__ = ["GroupALS", "plan_shards", "GenerationSerializer", "SerializedGeneration", "ALSCore", "HostSolver", "IllConditioned", "Generation", "Solver", "Ingest", "readInputRecords",
           "IllConditionedSolverException", "MalsError", "SingularSystem", "Cancelled", "AlternatingLeastSquares",
           "MatrixFactorizer", "MatrixUtils", "System", "ExecutionException",
           "InterruptedException", "SolverException", "SingularMatrixSolverException",
           "SIDE_X", "SIDE_Y", "FLAG_RECONSTRUCT_R", "FLAG_LOSS_IGNORES_UNSPECIFIED"]
