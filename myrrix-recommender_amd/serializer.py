"""Host mirror of the model file of the reference:
net.myrrix.online.generation.GenerationSerializer.readGeneration / writeGeneration
(online-local/src/net/myrrix/online/generation/GenerationSerializer.java:84-95) -- the `model.bin.gz`
DelegateGenerationManager saves after a build (DGM:270-289) and loads at start-up (DGM:303-305).
Same names, argument meaning and error behaviour; the bytes are produced and parsed by the C-ABI
(`mals_model_*`, csrc/model_io.cpp), this file only moves arrays in and out.

The reference holds the model in maps (FastByIDMap<float[]>, FastByIDMap<FastIDSet>); at 10^7 rows
those are arrays here: ids + a dense row-major matrix, and the known-item sets in CSR form."""
import ctypes
from dataclasses import dataclass, field

import numpy as np

from . import _lib


class IOException(IOError):
    """java.io.IOException (unreadable, truncated or corrupt model file)."""


class IllegalStateException(RuntimeError):
    """Preconditions.checkState failures (non-finite factor, GS:175,196) / IllegalArgumentException."""


def _i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64).reshape(-1))


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


@dataclass
class SerializedGeneration:
    """The seven constructor arguments of Generation that the file carries (GS:117-123)."""
    userIDs: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    X: np.ndarray = field(default_factory=lambda: np.zeros((0, 0), np.float32))        # n_users x features
    itemIDs: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    Y: np.ndarray = field(default_factory=lambda: np.zeros((0, 0), np.float32))        # n_items x features
    # knownItemIDs: None (model.noKnownItems) or (user ids, offsets[n+1], item ids)
    knownItemIDs: tuple = None
    itemTagIDs: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    userTagIDs: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int64))
    userClusters: list = field(default_factory=list)                                   # [(member ids, centroid)]
    itemClusters: list = field(default_factory=list)

    def getX(self):
        """id -> row view, the shape AlternatingLeastSquares.setPreviousY expects (ALS:172-174)."""
        return {int(i): self.X[n] for n, i in enumerate(self.userIDs)}

    def getY(self):
        return {int(i): self.Y[n] for n, i in enumerate(self.itemIDs)}

    def getKnownItemIDs(self):
        if self.knownItemIDs is None:
            return None
        ids, ptr, items = self.knownItemIDs
        return {int(u): items[ptr[n]:ptr[n + 1]] for n, u in enumerate(ids)}


def _check(status):
    if status == _lib.OK:
        return
    msg = _lib.load().mals_model_last_error().decode()
    if status == _lib.IO_ERROR:
        raise IOException(msg)
    if status == _lib.INVALID_ARG:
        raise IllegalStateException(msg)
    raise _lib_error(status, msg)


def _lib_error(status, msg):
    from .core import MalsError
    return MalsError(status, msg)


def _clusters_to_arrays(clusters):
    mptr, cptr = [0], [0]
    members, cent = [], []
    for m, c in clusters:
        members.append(_i64(m))
        cent.append(_f32(c).reshape(-1))
        mptr.append(mptr[-1] + len(members[-1]))
        cptr.append(cptr[-1] + len(cent[-1]))
    cat = lambda parts, dt: np.concatenate(parts) if parts else np.zeros(0, dt)  # noqa: E731
    return _i64(mptr), cat(members, np.int64), _i64(cptr), cat(cent, np.float32)


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a.size else None


class GenerationSerializer:
    """GenerationSerializer.java:46-95."""

    @staticmethod
    def writeGeneration(generation, f):
        g = generation
        X, Y = _f32(g.X), _f32(g.Y)
        uid, iid = _i64(g.userIDs), _i64(g.itemIDs)
        if X.ndim != 2 or Y.ndim != 2 or X.shape[0] != uid.size or Y.shape[0] != iid.size:
            raise IllegalStateException("X / Y must be (n x features) arrays matching their id arrays")
        if X.shape[0] and Y.shape[0] and X.shape[1] != Y.shape[1]:
            raise IllegalStateException("X and Y differ in their number of features")
        v = _lib.ModelView()
        v.struct_size = ctypes.sizeof(_lib.ModelView)
        v.features = X.shape[1] if X.shape[0] else (Y.shape[1] if Y.shape[0] else 0)
        keep = [X, Y, uid, iid]
        v.n_users, v.user_ids, v.X = uid.size, _ptr(uid), _ptr(X)
        v.n_items, v.item_ids, v.Y = iid.size, _ptr(iid), _ptr(Y)
        if g.knownItemIDs is None:
            v.n_known = -1
        else:
            ku, kp, ki = (_i64(a) for a in g.knownItemIDs)
            if kp.size != ku.size + 1 or (kp.size and (kp[0] != 0 or kp[-1] != ki.size)) or np.any(np.diff(kp) < 0):
                raise IllegalStateException("knownItemIDs offsets are inconsistent")
            keep += [ku, kp, ki]
            v.n_known, v.known_user_ids, v.known_ptr, v.known_item_ids = ku.size, _ptr(ku), _ptr(kp), _ptr(ki)
        it, ut = _i64(g.itemTagIDs), _i64(g.userTagIDs)
        keep += [it, ut]
        v.n_item_tags, v.item_tag_ids, v.n_user_tags, v.user_tag_ids = it.size, _ptr(it), ut.size, _ptr(ut)
        uc, ic = _clusters_to_arrays(g.userClusters), _clusters_to_arrays(g.itemClusters)
        keep += list(uc) + list(ic)
        v.n_user_clusters = len(g.userClusters)
        (v.user_cluster_member_ptr, v.user_cluster_members, v.user_cluster_centroid_ptr,
         v.user_cluster_centroids) = (_ptr(a) for a in uc)
        v.n_item_clusters = len(g.itemClusters)
        (v.item_cluster_member_ptr, v.item_cluster_members, v.item_cluster_centroid_ptr,
         v.item_cluster_centroids) = (_ptr(a) for a in ic)
        _check(_lib.load().mals_model_write(str(f).encode(), ctypes.byref(v)))
        del keep

    @staticmethod
    def readGeneration(f):
        L = _lib.load()
        h = ctypes.c_void_p()
        _check(L.mals_model_read(str(f).encode(), ctypes.byref(h)))
        try:
            v = _lib.ModelView()
            _check(L.mals_model_get(h, ctypes.byref(v)))

            def arr(ptr, n, ctype, dtype):
                if not n:
                    return np.zeros(0, dtype)
                return np.ctypeslib.as_array(ctypes.cast(ptr, ctypes.POINTER(ctype)), shape=(n,)).astype(dtype, copy=True)

            def i64(ptr, n):
                return arr(ptr, n, ctypes.c_int64, np.int64)

            def f32(ptr, n):
                return arr(ptr, n, ctypes.c_float, np.float32)

            def clusters(n, mptr, members, cptr, cent):
                mp, cp = i64(mptr, n + 1), i64(cptr, n + 1)
                mem, ce = i64(members, int(mp[-1])), f32(cent, int(cp[-1]))
                return [(mem[mp[c]:mp[c + 1]], ce[cp[c]:cp[c + 1]]) for c in range(n)]

            k = v.features
            g = SerializedGeneration(
                userIDs=i64(v.user_ids, v.n_users), X=f32(v.X, v.n_users * k).reshape(v.n_users, k),
                itemIDs=i64(v.item_ids, v.n_items), Y=f32(v.Y, v.n_items * k).reshape(v.n_items, k),
                itemTagIDs=i64(v.item_tag_ids, v.n_item_tags), userTagIDs=i64(v.user_tag_ids, v.n_user_tags),
                userClusters=clusters(v.n_user_clusters, v.user_cluster_member_ptr, v.user_cluster_members,
                                      v.user_cluster_centroid_ptr, v.user_cluster_centroids),
                itemClusters=clusters(v.n_item_clusters, v.item_cluster_member_ptr, v.item_cluster_members,
                                      v.item_cluster_centroid_ptr, v.item_cluster_centroids))
            if v.n_known >= 0:
                kp = i64(v.known_ptr, v.n_known + 1)
                g.knownItemIDs = (i64(v.known_user_ids, v.n_known), kp, i64(v.known_item_ids, int(kp[-1])))
            return g
        finally:
            L.mals_model_destroy(h)
