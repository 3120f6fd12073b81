"""Ingest -> CSR on the device: plumbing over mals_ingest_* and the
host mirror of the reference entry point.

`Ingest` is one mals_ingest object.  `readInputFiles` mirrors InputFilesReader.readInputFiles
(online-local/.../generation/InputFilesReader.java:64-211): a directory of *.csv / *.csv.gz files in,
what the reference leaves in RbyRow / RbyColumn, itemTagIDs, userTagIDs and knownItemIDs out -- as id
tables + CSR matrices; `readInputRecords` is its second half alone, for callers that parse lines
themselves."""
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

    def set_option(self, option, value):
        self._chk(self._L.mals_ingest_set_option(self._g, int(option), int(value)))

    def append_text(self, data, end_of_file=True):
        """Bytes of an input file (bytes-like, or a torch uint8 CUDA tensor already in HBM).  A file may arrive in any
        number of pieces, split anywhere; the last piece says end_of_file."""
        if _is_torch(data):
            assert data.is_cuda and str(data.dtype) == "torch.uint8"
            d = data.contiguous()
            self._chk(self._L.mals_ingest_append_text(self._g, ctypes.c_void_p(d.data_ptr()), int(d.numel()), MEM_DEVICE, int(end_of_file)))
        else:
            b = bytes(data)
            self._chk(self._L.mals_ingest_append_text(self._g, ctypes.c_char_p(b) if b else None, len(b), MEM_HOST, int(end_of_file)))

    def read_file(self, path):
        self._chk(self._L.mals_ingest_read_file(self._g, str(path).encode()))

    def read_dir(self, path):
        n = ctypes.c_int32()
        self._chk(self._L.mals_ingest_read_dir(self._g, str(path).encode(), ctypes.byref(n)))
        return n.value

    def text_info(self):
        info = _lib.IngestTextInfo()
        info.struct_size = ctypes.sizeof(info)
        self._chk(self._L.mals_ingest_text_info(self._g, ctypes.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_ if k not in ("struct_size", "reserved")}

    def tag_ids(self, which):
        n = self.text_info()["n_item_tag_ids" if which == _lib.ITEM_TAG_IDS else "n_user_tag_ids"]
        out = np.empty(max(n, 0), dtype=np.int64)
        self._chk(self._L.mals_ingest_get_tag_ids(self._g, which, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def known_items(self):
        """knownItemIDs as (row_ptr over the dense users, dense item indices)."""
        c = self.counts()
        n = self.text_info()["n_known_items"]
        ptr = np.empty(c["users"] + 1, dtype=np.int64)
        idx = np.empty(max(n, 0), dtype=np.int32)
        self._chk(self._L.mals_ingest_get_known_items(self._g, ptr.ctypes.data_as(ctypes.c_void_p), idx.ctypes.data_as(ctypes.c_void_p)))
        return ptr, idx

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

    def install_group(self, group, copy=False):
        """mals_ingest_install_group: both matrices cut at the group's bounds, every member its slices device to device,
        knownItemIDs and the userTagIDs mask included.  copy=False: slices are borrowed where the member shares the device
        (this object is kept alive by the group wrapper); copy=True (MALS_INSTALL_COPY): every member owns its slices and
        this object may be closed."""
        self._chk(self._L.mals_ingest_install_group(self._g, group._g, _lib.INSTALL_COPY if copy else 0))
        if not copy:
            group._keep_ingest = self

    def tag_items(self):
        """dense item index per userTagID (ascending ids), -1 = the tag owns no row of R^T"""
        n = len(self.tag_ids(_lib.USER_TAG_IDS))
        out = np.empty(n, dtype=np.int64)
        self._chk(self._L.mals_ingest_get_tag_items(self._g, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def partitions(self):
        """(user-id ranges, item ranges) the last finish was cut into; (0, 0): one sort pipeline held everything"""
        a, b = ctypes.c_int32(0), ctypes.c_int32(0)
        self._chk(self._L.mals_ingest_partitions(self._g, ctypes.byref(a), ctypes.byref(b)))
        return a.value, b.value

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


def readInputFiles(inputDir, device=0, zero_threshold=None, known_items=True):
    """InputFilesReader.readInputFiles(knownItemIDs, rbyRow, rbyColumn, itemTagIDs, userTagIDs, inputDir) (IFR:64-192):
    returns a dict with the id tables, both CSR matrices, the two tag id sets and knownItemIDs (None when the caller
    runs with model.noKnownItems, IFR:173)."""
    import logging
    from .factorizer import System
    log = logging.getLogger("net.myrrix.online.generation.InputFilesReader")
    if zero_threshold is None:
        zero_threshold = float(System.getProperty("model.decay.zeroThreshold", "0.0001"))  # IFR:58-59
    with Ingest(device, zero_threshold) as g:
        if known_items:
            g.set_option(_lib.INGEST_OPT_KNOWN_ITEMS, 1)
        n_files = g.read_dir(inputDir)
        if n_files == 0:
            log.info("No input files in %s", inputDir)
        log.info("Pruning near-zero entries")
        g.finish()
        info = g.text_info()
        return {"user_ids": g.ids(SIDE_X), "item_ids": g.ids(SIDE_Y), "RbyRow": g.csr(SIDE_X), "RbyColumn": g.csr(SIDE_Y),
                "itemTagIDs": g.tag_ids(_lib.ITEM_TAG_IDS), "userTagIDs": g.tag_ids(_lib.USER_TAG_IDS),
                "knownItemIDs": g.known_items() if known_items else None, "info": info}


__all__ = ["Ingest", "readInputRecords", "readInputFiles"]
