"""Ingest -> CSR on the device: plumbing over mals_ingest_* and the
host mirror of the reference entry point.

`Ingest` is one mals_ingest object.  `readInputRecords` mirrors the part of
InputFilesReader.readInputFiles (online-local/.../generation/InputFilesReader.java:64-211) that
follows line parsing: it takes the parsed records in file order and returns what the reference
leaves in RbyRow / RbyColumn -- as id tables + two CSR matrices."""
import ctypes

import numpy as np

from . import _lib
from ._lib import MEM_DEVICE, MEM_HOST, SIDE_X, SIDE_Y
from .core import MalsError, _is_torch


class Ingest:
    def __init__(self, device=0, zero_threshold=1.0e-4):
        self._L = _lib.load()
        self._g = ctypes.c_void_p()
        rc = self._L.mals_ingest_create(int(device), float(zero_threshold), ctypes.byref(self._g))
        if rc != _lib.OK:
            self._g = ctypes.c_void_p()
            raise MalsError(rc, "mals_ingest_create failed (device=%d): a HIP device is required, "
                                "there is no CPU fallback" % device)

    def close(self):
        if getattr(self, "_g", None) is not None and self._g.value:
            self._L.mals_ingest_destroy(self._g)
            self._g = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _chk(self, rc):
        if rc != _lib.OK:
            raise MalsError(rc, (self._L.mals_ingest_last_error(self._g) or b"").decode("utf-8", "replace"))

    def append(self, user_ids, item_ids, values):
        """Records in file order; value NaN = remove.  numpy arrays or torch CUDA tensors."""
        if _is_torch(user_ids):
            assert user_ids.is_cuda and item_ids.is_cuda and values.is_cuda
            assert str(user_ids.dtype) == "torch.int64" and str(item_ids.dtype) == "torch.int64" \
                and str(values.dtype) == "torch.float32"
            u, i, v = user_ids.contiguous(), item_ids.contiguous(), values.contiguous()
            self._chk(self._L.mals_ingest_append(self._g, int(u.shape[0]), ctypes.c_void_p(u.data_ptr()),
                                                 ctypes.c_void_p(i.data_ptr()), ctypes.c_void_p(v.data_ptr()), MEM_DEVICE))
        else:
            u = np.ascontiguousarray(user_ids, dtype=np.int64)
            i = np.ascontiguousarray(item_ids, dtype=np.int64)
            v = np.ascontiguousarray(values, dtype=np.float32)
            assert u.shape == i.shape == v.shape and u.ndim == 1
            self._chk(self._L.mals_ingest_append(self._g, len(u), u.ctypes.data_as(ctypes.c_void_p),
                                                 i.ctypes.data_as(ctypes.c_void_p), v.ctypes.data_as(ctypes.c_void_p), MEM_HOST))

    def finish(self):
        self._chk(self._L.mals_ingest_finish(self._g))

    def counts(self):
        a = [ctypes.c_int64() for _ in range(4)]
        self._chk(self._L.mals_ingest_counts(self._g, *[ctypes.byref(x) for x in a]))
        return {"records": a[0].value, "users": a[1].value, "items": a[2].value, "nnz": a[3].value}

    def ids(self, side):
        c = self.counts()
        out = np.empty(c["users"] if side == SIDE_X else c["items"], dtype=np.int64)
        self._chk(self._L.mals_ingest_get_ids(self._g, side, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def csr(self, side):
        c = self.counts()
        rows = c["users"] if side == SIDE_X else c["items"]
        rp = np.empty(rows + 1, dtype=np.int64)
        col = np.empty(c["nnz"], dtype=np.int32)
        val = np.empty(c["nnz"], dtype=np.float32)
        self._chk(self._L.mals_ingest_get_csr(self._g, side, rp.ctypes.data_as(ctypes.c_void_p),
                                              col.ctypes.data_as(ctypes.c_void_p), val.ctypes.data_as(ctypes.c_void_p)))
        return rp, col, val

    def install(self, core):
        """Hand both matrices to an ALSCore on the same device (borrowed: keep this object alive)."""
        self._chk(self._L.mals_ingest_install(self._g, core._h))
        core._keep[("ingest",)] = self

    def stats(self):
        ms, ws, by, p = ctypes.c_double(), ctypes.c_double(), ctypes.c_double(), ctypes.c_int32()
        self._chk(self._L.mals_ingest_stats(self._g, ctypes.byref(ms), ctypes.byref(ws), ctypes.byref(by), ctypes.byref(p)))
        return {"finish_ms": ms.value, "workspace_ms": ws.value, "bytes_moved": by.value, "radix_passes": p.value}


def readInputRecords(user_ids, item_ids, values, device=0, zero_threshold=None):
    """The record-processing half of InputFilesReader.readInputFiles (IFR:112-199): returns
    (user id table, item id table, R by user as CSR, R^T by item as CSR)."""
    from .factorizer import System
    if zero_threshold is None:
        zero_threshold = float(System.getProperty("model.decay.zeroThreshold", "0.0001"))  # IFR:58-59
    with Ingest(device, zero_threshold) as g:
        g.append(user_ids, item_ids, values)
        g.finish()
        return g.ids(SIDE_X), g.ids(SIDE_Y), g.csr(SIDE_X), g.csr(SIDE_Y)


__all__ = ["Ingest", "readInputRecords"]
