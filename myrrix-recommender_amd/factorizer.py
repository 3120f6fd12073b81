"""Host-side mirror of the reference's factorizer interface over the C-ABI.

Same names, argument meaning and error behaviour as
  net.myrrix.online.factorizer.MatrixFactorizer            (online/src/.../MatrixFactorizer.java:31-77)
  net.myrrix.online.factorizer.als.AlternatingLeastSquares (online/src/.../als/AlternatingLeastSquares.java:66-262)
  net.myrrix.common.math.MatrixUtils.addTo / multiplyXYT   (common/src/.../math/MatrixUtils.java:64-92,155-165)
so that the parity tests read like the reference's own unit tests.  The JVM is not available in
this image, so this mirror stands where the Java adapter of INTEGRATION.md would; like it, it only
densifies ids, streams CSR over the C-ABI and copies factors back -- all arithmetic of the hot path
runs in the HIP library.
"""
import logging
import math

import numpy as np

from . import _lib
from .random_mt import MersenneTwister
from .core import ALSCore, Cancelled, MalsError, SingularSystem


# the reference logs under its own class name (ALS:68)
log = logging.getLogger("net.myrrix.online.factorizer.als.AlternatingLeastSquares")


class _System:
    """java.lang.System property store: the reference takes every model knob from system
    properties (SURVEY.md section 5 "Config")."""

    def __init__(self):
        self._p = {}

    def setProperty(self, key, value):
        self._p[key] = str(value)

    def getProperty(self, key, default=None):
        return self._p.get(key, default)

    def clearProperty(self, key):
        self._p.pop(key, None)


System = _System()


class SolverException(RuntimeError):
    """net.myrrix.common.math.SolverException"""


class SingularMatrixSolverException(SolverException):
    """net.myrrix.common.math.SingularMatrixSolverException (carries apparentRank)."""

    def __init__(self, apparentRank, message=None):
        super().__init__(message or ("Apparent rank: %d" % apparentRank))
        self.apparentRank = apparentRank

    def getApparentRank(self):
        return self.apparentRank


class ExecutionException(Exception):
    """java.util.concurrent.ExecutionException: worker failures surface wrapped (ALS:349)."""

    def __init__(self, cause):
        super().__init__(repr(cause))
        self.cause = cause

    def getCause(self):
        return self.cause


class InterruptedException(Exception):
    """java.lang.InterruptedException (MatrixFactorizer.java:43-44)."""


class MatrixUtils:
    @staticmethod
    def addTo(row, column, value, RbyRow, RbyColumn):
        """MU:64-92: increment an entry in two parallel sparse matrices (duplicates sum)."""
        value = np.float32(value)
        r = RbyRow.setdefault(row, {})
        r[column] = np.float32(r.get(column, np.float32(0)) + value)
        c = RbyColumn.setdefault(column, {})
        c[row] = np.float32(c.get(row, np.float32(0)) + value)

    @staticmethod
    def remove(row, column, RbyRow, RbyColumn):
        """MU:94-125: remove an entry; an emptied row is deleted."""
        for a, b, M in ((row, column, RbyRow), (column, row, RbyColumn)):
            the_row = M.get(a)
            if the_row is not None:
                the_row.pop(b, None)
                if not the_row:
                    del M[a]

    @staticmethod
    def multiplyXYT(X, Y):
        """MU:155-165: dense product over ids 0..n-1; dot = float product, double sum (SVM:34-41)."""
        nx, ny = len(X), len(Y)
        out = np.zeros((nx, ny), dtype=np.float64)
        for i in range(nx):
            for j in range(ny):
                p = (np.asarray(X[i], np.float32) * np.asarray(Y[j], np.float32)).astype(np.float32)
                out[i, j] = float(np.sum(p.astype(np.float64)))
        return out


class MatrixFactorizer:
    """MatrixFactorizer.java:31-77"""
    DEFAULT_FEATURES = 30

    def call(self):
        raise NotImplementedError

    def setPreviousX(self, previousX):
        raise NotImplementedError

    def setPreviousY(self, previousY):
        raise NotImplementedError

    def getX(self):
        raise NotImplementedError

    def getY(self):
        raise NotImplementedError


def _random_unit_vector_far_from(k, far_from, rng):
    """RandomUtils.randomUnitVectorFarFrom (common/src/.../random/RandomUtils.java:110-140) on the reference's own
    random stream (random_mt.MersenneTwister = commons-math3's generator as RandomManager seeds it): the Gaussians of
    doRandomUnitVector (RU:88-100), then per sampled earlier vector one nextInt(size) (RU:124, only when there are more
    than 100 of them), then one nextDouble() for the acceptance (RU:136) -- in that order, so that the stream stays
    aligned with the JVM's."""
    size = len(far_from)
    num_samples = min(100, size)
    while True:
        d = rng.nextGaussians(k)                                   # RU:91-95
        v = d.astype(np.float32)
        total = 0.0
        for x in d:                                                # total += d * d in that order (RU:94)
            total += x * x
        v /= np.float32(math.sqrt(total))                          # RU:96-98: float divisor
        smallest = math.inf
        picks = range(num_samples) if size == num_samples else rng.nextInts(num_samples, size)
        for s in picks:
            other = far_from[int(s)]
            dot = float(np.sum((v * other).astype(np.float32).astype(np.float64)))   # SimpleVectorMath.dot: float products
            dist2 = 2.0 - 2.0 * dot
            if math.isfinite(dist2) and dist2 < smallest:
                smallest = dist2
        if math.isfinite(smallest) and not (k == 1 and smallest == 0.0):
            if rng.nextDouble() < smallest / 4.0:
                return v
        else:
            return v


def _choose_about_n(n, ids, rng):
    """RandomUtils.chooseAboutNFromStream (RandomUtils.java:202-217): everything when
    n >= stream size, else geometric-skip sampling at rate n/size."""
    size = len(ids)
    if n >= size:
        return list(range(size))
    # SamplingLongPrimitiveIterator.doNext skips PascalDistribution(random, 1, rate).sample() elements, which in
    # commons-math3 is inverseCumulativeProbability(random.nextDouble()): the smallest x with 1 - (1-rate)^(x+1) >= u,
    # i.e. floor(log(1-u) / log(1-rate)) -- one nextDouble() per skip, the same value as the JVM's numerical inverse
    # except for a u within rounding of a step of the CDF
    rate = n / size
    out, pos = [], -1
    while True:
        u = rng.nextDouble()
        skip = int(math.floor(math.log1p(-u) / math.log1p(-rate))) if u < 1.0 else 0
        pos += 1 + skip
        if pos >= size:
            break
        out.append(pos)
    return out


class AlternatingLeastSquares(MatrixFactorizer):
    """AlternatingLeastSquares.java:66 -- same constructor and methods; the work happens in
    libmyrrix_als.so."""
    DEFAULT_ALPHA = 1.0                       # ALS:71
    DEFAULT_LAMBDA = 0.1                      # ALS:73
    DEFAULT_CONVERGENCE_THRESHOLD = 0.001     # ALS:74
    DEFAULT_MAX_ITERATIONS = 30               # ALS:75
    NUM_USER_ITEMS_TO_TEST_CONVERGENCE = 100  # ALS:80
    MAX_FAR_FROM_VECTORS = 100000             # ALS:83

    def __init__(self, RbyRow, RbyColumn, features, estimateErrorConvergenceThreshold, maxIterations,
                 device=0):
        if RbyRow is None or RbyColumn is None:
            raise ValueError("RbyRow/RbyColumn must not be null")             # ALS:137-138
        if not features > 0:
            raise ValueError("features must be positive: %s" % features)      # ALS:139
        if not (0.0 < estimateErrorConvergenceThreshold < 1.0):
            raise ValueError("threshold must be in (0,1): %s" % estimateErrorConvergenceThreshold)
        self.RbyRow = RbyRow
        self.RbyColumn = RbyColumn
        self.features = int(features)
        self.estimateErrorConvergenceThreshold = float(estimateErrorConvergenceThreshold)
        self.maxIterations = int(maxIterations)
        self.device = device
        self.previousY = None
        self.X = None
        self.Y = None
        self.iterations = 0
        self.convergenceValue = float("nan")
        self._core = None

    def getX(self):
        return self.X

    def getY(self):
        return self.Y

    def setPreviousX(self, previousX):
        pass  # ALS:162-165 "Does nothing."

    def setPreviousY(self, previousY):
        self.previousY = previousY

    def cancel(self):
        """Thread interruption analogue (MatrixFactorizer.java:43-44)."""
        if self._core is not None:
            self._core.cancel()

    # ---------------------------------------------------------------------------------------------
    def _construct_initial_y(self, rng):
        """ALS:264-335 constructInitialY, host side (one-off, not GPU work)."""
        k = self.features
        prev = self.previousY
        if not prev:
            Y = {}
        else:
            old_k = len(next(iter(prev.values())))
            if old_k > k:      # ALS:277-287 project down + renormalise
                Y = {}
                for id_, vec in prev.items():
                    v = np.array(vec[:k], dtype=np.float32)
                    nrm = np.float32(math.sqrt(float(np.sum((v * v).astype(np.float64)))))
                    Y[id_] = v / nrm
            elif old_k < k:    # ALS:289-302 pad with N(0,1) + renormalise
                Y = {}
                for id_, vec in prev.items():
                    v = np.zeros(k, dtype=np.float32)
                    v[:old_k] = vec
                    v[old_k:] = rng.nextGaussians(k - old_k).astype(np.float32)   # ALS:297-299
                    nrm = np.float32(math.sqrt(float(np.sum((v * v).astype(np.float64)))))
                    Y[id_] = v / nrm
            else:              # ALS:304-308 same feature count: use as is
                Y = {id_: np.asarray(vec, dtype=np.float32) for id_, vec in prev.items()}
        recent = list(Y.values())[:self.MAX_FAR_FROM_VECTORS]
        for id_ in self.RbyColumn.keys():   # ALS:318-328
            if id_ not in Y:
                v = _random_unit_vector_far_from(k, recent, rng)
                Y[id_] = v
                if len(recent) < self.MAX_FAR_FROM_VECTORS:
                    recent.append(v)
        return Y

    @classmethod
    def _convergence_sample(cls, seed, user_ids, item_ids):
        """ALS:205-215: dense indices of the ~100 x ~100 test pairs, users first, from ONE fresh generator."""
        rng_sample = MersenneTwister(seed)
        tu = _choose_about_n(cls.NUM_USER_ITEMS_TO_TEST_CONVERGENCE, user_ids, rng_sample)  # ALS:206-209
        ti = _choose_about_n(cls.NUM_USER_ITEMS_TO_TEST_CONVERGENCE, item_ids, rng_sample)  # ALS:210-213
        return tu, ti

    @staticmethod
    def _csr(rows_by_id, row_ids, col_index, what):
        row_ptr = np.zeros(len(row_ids) + 1, dtype=np.int64)
        cols, vals = [], []
        for i, rid in enumerate(row_ids):
            for cid, v in rows_by_id[rid].items():
                j = col_index.get(cid)
                if j is None:
                    # the reference logs "No vector for {}. This should not happen." (ALS:460-463)
                    raise ValueError("%s references id %r that has no row on the other side" % (what, cid))
                cols.append(j)
                vals.append(v)
            row_ptr[i + 1] = len(cols)
        return row_ptr, np.array(cols, dtype=np.int32), np.array(vals, dtype=np.float32)

    def call(self):
        """ALS:176-262."""
        k = self.features
        alpha = float(System.getProperty("model.als.alpha", self.DEFAULT_ALPHA))          # ALS:506-509
        lam = float(System.getProperty("model.als.lambda", self.DEFAULT_LAMBDA))          # ALS:511-514
        flags = 0
        if str(System.getProperty("model.reconstructRMatrix", "false")).lower() == "true":
            flags |= _lib.FLAG_RECONSTRUCT_R                                               # ALS:85-87
        if str(System.getProperty("model.lossIgnoresUnspecified", "false")).lower() == "true":
            flags |= _lib.FLAG_LOSS_IGNORES_UNSPECIFIED                                    # ALS:89-91
        iterate = str(System.getProperty("model.als.iterate", "true")).lower() == "true"  # ALS:196
        sing = float(System.getProperty("common.matrix.singularityThreshold", 1.0e-5))
        seed = int(System.getProperty("model.test.seed", 1234567890))   # RandomManager.java:52
        rng = MersenneTwister(seed)                                      # RM:63-73: commons-math3 MersenneTwister

        random_y = not self.previousY                                    # ALS:181
        Y0 = self._construct_initial_y(rng)                              # ALS:182
        user_ids = list(self.RbyRow.keys())
        item_ids = list(self.RbyColumn.keys())
        stale_ids = [i for i in Y0.keys() if i not in self.RbyColumn]
        y_ids = item_ids + stale_ids
        user_index = {u: i for i, u in enumerate(user_ids)}
        item_index = {v: i for i, v in enumerate(y_ids)}
        n_users, n_items, n_y = len(user_ids), len(item_ids), len(y_ids)
        if n_users == 0 or n_y == 0:
            self.X, self.Y = {}, {id_: Y0[id_] for id_ in y_ids}
            return None
        r_csr = self._csr(self.RbyRow, user_ids, item_index, "RbyRow")
        c_csr = self._csr(self.RbyColumn, item_ids, user_index, "RbyColumn")
        Y0m = np.stack([Y0[i] for i in y_ids]).astype(np.float32)

        # ALS:206 asks RandomManager for ANOTHER generator (constructInitialY had its own, ALS:266): under the test seed
        # every getRandom() is a fresh MersenneTwister(TEST_SEED) (RandomManager.java:61-64), so the sample's skips
        # start at the beginning of the stream whatever constructInitialY consumed
        tu, ti = self._convergence_sample(seed, user_ids, item_ids)

        try:
            core = ALSCore(k, alpha=alpha, lam=lam, flags=flags, device=self.device,
                           singularity_threshold=sing)
        except MalsError as e:
            raise ExecutionException(RuntimeError(str(e)))
        self._core = core
        try:
            core.set_factor_rows(_lib.SIDE_X, n_users)
            core.set_factor_rows(_lib.SIDE_Y, n_y)
            core.set_matrix(_lib.SIDE_X, *r_csr)
            core.set_matrix(_lib.SIDE_Y, *c_csr)
            core.set_factors(_lib.SIDE_Y, Y0m)
            log.info("Iterating using 1 GPU(s) [%d] (%d features, %d users, %d items)", self.device, k, n_users, n_items)  # ALS:193
            threshold, max_it = self.estimateErrorConvergenceThreshold, self.maxIterations

            def on_iteration(info):   # the reference's per-iteration lines, ALS:241-256, 351-358
                secs = max(info["seconds"], 1e-9)
                log.info("%d X/tag rows computed, %d Y/tag rows computed (%d entries gathered in %.3f s: %.3g rows/s, %.1f GB/s)",
                         info["x_rows"], info["y_rows"], info["entries_gathered"], info["seconds"],
                         (info["x_rows"] + info["y_rows"]) / secs, info["algorithmic_bytes"] / secs / 1e9)
                log.info("Finished iteration %d", info["iteration"])
                if max_it > 0 and info["iteration"] >= max_it:
                    log.info("Reached iteration limit")
                    return
                log.info("Avg absolute difference in estimate vs prior iteration: %s", info["avg_abs_difference"])
                if not math.isfinite(info["avg_abs_difference"]):
                    log.warning("Invalid convergence value, aborting iteration! %s", info["avg_abs_difference"])
                elif not (random_y and info["iteration"] == 1) and info["avg_abs_difference"] < threshold:
                    log.info("Converged")
            self.iterationLog = []
            core.set_iteration_callback(lambda info: (self.iterationLog.append(info), on_iteration(info)))
            self.iterations, self.convergenceValue = core.factorize(
                self.estimateErrorConvergenceThreshold, self.maxIterations, random_y, tu, ti,
                iterate=iterate)
            Xm = core.get_factors(_lib.SIDE_X, 0, n_users)
            Ym = core.get_factors(_lib.SIDE_Y, 0, n_y)
        except SingularSystem as e:
            raise ExecutionException(SingularMatrixSolverException(e.apparent_rank, e.message))
        except Cancelled:
            raise InterruptedException()
        except MalsError as e:
            raise ExecutionException(RuntimeError(str(e)))
        finally:
            core.close()
            self._core = None
        self.X = {u: Xm[i].copy() for i, u in enumerate(user_ids)}
        self.Y = {v: Ym[i].copy() for i, v in enumerate(y_ids)}
        return None
