"""The C-ABI library loads and exports every symbol include/myrrix_als.h declares (no compute
calls here: this runs on the CPU-only box)."""
import ctypes
import os
import re

import pytest

import myrrix_recommender_amd as pkg
from myrrix_recommender_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, "include", "myrrix_als.h")).read()


def declared_functions():
    body = re.sub(r"/\*.*?\*/", "", HEADER, flags=re.S)
    return sorted(set(re.findall(r"\b(mals_[a-z_]+)\s*\(", body)))


def test_header_and_binding_agree():
    assert set(declared_functions()) == set(_lib.SYMBOLS.keys())


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    L = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_functions():
        assert hasattr(L, name), name
    assert _lib.load().mals_abi_version() == _lib.ABI_VERSION == 5


def test_struct_layouts_match_header():
    # mals_config: 2 x int32, 3 x double, 6 x int32 ; mals_stats (ABI 2): 2 x int32, 4 x double, 4 x int64, 4 x double,
    # 2 x int64, then 2 x double, 2 x int64, 2 x double, int64, double
    assert ctypes.sizeof(_lib.Config) == 56
    assert ctypes.sizeof(_lib.Stats) == 192
    cfg = _lib.Config()
    assert _lib.load().mals_default_config(ctypes.byref(cfg)) == _lib.OK
    assert cfg.struct_size == 56 and cfg.features == 30            # MatrixFactorizer.java:34
    assert cfg.alpha == 1.0 and abs(cfg.lam - 0.1) < 1e-15         # ALS:71-73
    assert cfg.singularity_threshold == 1e-5                       # LinearSystemSolver.java:33-34


def test_invalid_arguments_are_rejected_without_a_gpu():
    L = _lib.load()
    h = ctypes.c_void_p()
    cfg = _lib.Config()
    L.mals_default_config(ctypes.byref(cfg))
    cfg.features = 0                                                # ALS:139 features must be positive
    assert L.mals_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.INVALID_ARG
    cfg.features = 129
    assert L.mals_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.INVALID_ARG
    assert L.mals_destroy(None) == _lib.INVALID_ARG
    assert L.mals_solve_side(None, 0) == _lib.INVALID_ARG


def test_create_failures_say_why():
    """mals_create_error / mals_group_create_error: a create that fails leaves no handle to ask -- the reason (which the
    JNI shim hands to the JVM, jni/myrrix_als_jni.c nativeCreateError) is kept per thread."""
    L = _lib.load()
    h = ctypes.c_void_p()
    cfg = _lib.Config()
    L.mals_default_config(ctypes.byref(cfg))
    cfg.features = 129
    assert L.mals_create(ctypes.byref(cfg), ctypes.byref(h)) == _lib.INVALID_ARG
    assert "features" in _lib.create_error() and "129" in _lib.create_error()
    buf = ctypes.create_string_buffer(8)                      # a short buffer: truncated, NUL-terminated, full length returned
    assert L.mals_group_create_error(buf, 8) > 8 and len(buf.value) == 7
    cfg.features = 16
    cfg.device = 4095
    rc = L.mals_create(ctypes.byref(cfg), ctypes.byref(h))
    assert rc == _lib.HIP_ERROR
    why = _lib.create_error()
    assert "4095" in why or "no HIP device" in why, why      # no GPU here: "no HIP device"; on a GPU box: the ordinal
    g = ctypes.c_void_p()
    devs = (ctypes.c_int32 * 1)(4095)
    assert L.mals_group_create(ctypes.byref(cfg), devs, 1, 1, ctypes.byref(g)) != _lib.OK and not g.value
    assert _lib.create_error().startswith("mals_group_create")


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.MalsError) as ei:
        pkg.ALSCore(30)
    assert ei.value.status == _lib.HIP_ERROR
    als = pkg.AlternatingLeastSquares({0: {0: 1.0}}, {0: {0: 1.0}}, 2, 0.001, 1)
    with pytest.raises(pkg.ExecutionException):
        als.call()


def test_new_entry_points_reject_bad_arguments_without_a_gpu():
    """Host-only argument checks of the section 8(f) entry points (no device work)."""
    L = _lib.load()
    g = ctypes.c_void_p()
    assert L.mals_ingest_create(0, ctypes.c_float(-1.0), ctypes.byref(g)) == _lib.INVALID_ARG     # negative threshold
    assert L.mals_ingest_create(0, ctypes.c_float(1e-4), None) == _lib.INVALID_ARG
    assert L.mals_ingest_finish(None) == _lib.INVALID_ARG
    assert L.mals_ingest_append(None, 0, None, None, None, 0) == _lib.INVALID_ARG
    assert L.mals_recommend(None, None, 0, 10, 0, None, None, None) == _lib.INVALID_ARG
    assert L.mals_recommend_vectors(None, None, 0, 10, None, None, None, None, None) == _lib.INVALID_ARG
    s, c = ctypes.c_double(), ctypes.c_int64()
    assert L.mals_reconstruction_error(None, ctypes.byref(s), ctypes.byref(c)) == _lib.INVALID_ARG
    solver, rank = ctypes.c_void_p(), ctypes.c_int32()
    assert L.mals_solver_create(None, 3, 1e-5, ctypes.byref(solver), ctypes.byref(rank)) == _lib.INVALID_ARG
    assert L.mals_solver_dim(None) == 0
    # group / planner entry points
    grp = ctypes.c_void_p()
    cfg = _lib.Config()
    L.mals_default_config(ctypes.byref(cfg))
    assert L.mals_group_create(ctypes.byref(cfg), None, 2, 0, ctypes.byref(grp)) == _lib.INVALID_ARG
    assert L.mals_group_create_rank(ctypes.byref(cfg), 2, 5, None, ctypes.byref(grp)) == _lib.INVALID_ARG   # rank >= world
    assert L.mals_group_half_iteration(None, 0) == _lib.INVALID_ARG
    assert L.mals_group_world(None) == 0
    assert L.mals_plan_shards(None, 0, 2, -1.0, 64, None) == _lib.INVALID_ARG
    assert L.mals_set_chunk_rows(None, 0, 10) == _lib.INVALID_ARG
    assert L.mals_set_refine_limit(None, 1024.0) == _lib.INVALID_ARG


def test_round6_entry_points_reject_null_handles_without_a_gpu():
    """The serving front, tag items, the group install and router, the ingest's range option: null handles and bad arguments
    are status codes, not crashes (no device work)."""
    L = _lib.load()
    n = ctypes.c_int64(-1)
    out4 = (ctypes.c_int64 * 4)()
    assert L.mals_features(None) == 0
    assert L.mals_recommend_front_stats(None, out4) == _lib.INVALID_ARG
    assert L.mals_recommend_set_depth(None, 2) == _lib.INVALID_ARG
    assert L.mals_recommend_set_spin_us(None, 0) == _lib.INVALID_ARG
    assert L.mals_set_tag_items(None, 0, None, _lib.MEM_HOST) == _lib.INVALID_ARG
    assert L.mals_get_tag_item_count(None, ctypes.byref(n)) == _lib.INVALID_ARG
    assert L.mals_ingest_install_group(None, None, 0) == _lib.INVALID_ARG
    assert L.mals_group_recommend(None, None, 0, 10, 0, None, None, None) == _lib.INVALID_ARG
    dev, a, b = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert L.mals_ingest_device(None, ctypes.byref(dev)) == _lib.INVALID_ARG
    assert L.mals_ingest_partitions(None, ctypes.byref(a), ctypes.byref(b)) == _lib.INVALID_ARG
    assert L.mals_ingest_get_tag_items(None, None) == _lib.INVALID_ARG
    assert L.mals_ingest_set_option(None, _lib.INGEST_OPT_PARTITION_RECORDS, 1000) == _lib.INVALID_ARG
    assert _lib.INSTALL_COPY == 1 and _lib.GRAMIAN_SPLIT3_F16 == 3 and _lib.INGEST_OPT_PARTITION_RECORDS == 4


def test_round3_entry_points_reject_bad_arguments_without_a_gpu():
    """Host-only argument checks of the entry points added in round 3 (no device work)."""
    L = _lib.load()
    out4 = (ctypes.c_double * 4)()
    assert L.mals_get_timeline(None, out4) == _lib.INVALID_ARG
    assert L.mals_group_features(None) == 0
    n = ctypes.c_int64(-1)
    assert L.mals_group_pending_entries(None, 0, 1, ctypes.byref(n)) == _lib.INVALID_ARG
    a, b, c = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
    assert L.mals_group_comm_info(None, 0, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), None, 0) == _lib.INVALID_ARG
    # the transport is chosen by an API call of the host process (never by the environment) and only before it is loaded
    assert L.mals_group_use_transport(None) == _lib.OK
    src = open(os.path.join(ROOT, "myrrix-recommender_amd", "csrc", "mals_group.cpp")).read()
    assert "getenv" not in src, "the group layer must not pick its transport (or anything else) from the environment"


def test_unloadable_transport_is_a_status_not_a_crash():
    """the error text of a failed dlopen used to be built from two dlerror() calls (the second returns NULL:
    std::string + nullptr) -- mals_group_unique_id must come back with MALS_COMM_ERROR.  Child process: the choice of
    transport is made once per process."""
    import subprocess
    import sys
    code = ("import ctypes, sys; sys.path.insert(0, %r); from myrrix_recommender_amd import _lib; L = _lib.load(); "
            "assert L.mals_group_use_transport(b'/nonexistent/librccl_missing.so') == 0; "
            "buf = (ctypes.c_uint8 * 128)(); rc = L.mals_group_unique_id(buf); print('rc', rc); sys.exit(0 if rc == _lib.COMM_ERROR else 1)" % ROOT)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)


def test_graft_entry_build_runs_clean():
    """The driver's "does it build" check is __graft_entry__.build(): it must pass on the tree as committed (it once kept
    asserting the previous ABI version after include/myrrix_als.h had moved on)."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    entry = importlib.import_module("__graft_entry__")
    entry.build()
    assert callable(entry.smoke)
