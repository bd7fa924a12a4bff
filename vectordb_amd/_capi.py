"""ctypes binding of the C ABI declared in include/vdb.h.

The product path loads ``vectordb_amd/lib/libvdb_hip.so`` (hand-written HIP for
gfx950) and fails loudly when it is missing or when no GPU is present — there is
no CPU fallback.  Tests may *explicitly* inject another library exporting the
same symbols (the CPU oracle under ``oracle/``) through ``load_library(path)``;
nothing in this package ever does that on its own.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
#: the HIP engine; VECTORDB_AMD_LIB names another build of the SAME library (kernel A/B runs), never another backend
HIP_LIB_PATH = os.environ.get("VECTORDB_AMD_LIB") or os.path.join(_PKG_DIR, "lib", "libvdb_hip.so")

VDB_F32 = 0
VDB_BF16 = 1
VDB_MAX_K = 1024

#: every symbol include/vdb.h declares (tests check the built library exports all of them)
SYMBOLS = (
    "vdb_create", "vdb_create_sharded", "vdb_load_sharded", "vdb_search_dev_noalloc", "vdb_destroy", "vdb_last_error", "vdb_reserve", "vdb_set_row_offset",
    "vdb_add", "vdb_add_dev", "vdb_count", "vdb_search", "vdb_search_fetch", "vdb_search_dev", "vdb_search_masked", "vdb_search_masked_dev",
    "vdb_search_scan_dev", "vdb_search_finish_dev",
    "vdb_merge_topk_dev",
    "vdb_selfjoin", "vdb_fetch", "vdb_save", "vdb_load", "vdb_last_timing", "vdb_timing_stats", "vdb_set_option", "vdb_get_stat",
    "vdb_synth_rows_dev", "vdb_add_synth", "vdb_rebalance", "vdb_warm",
)


class VdbError(RuntimeError):
    """Raised when a vdb_* call returns a negative status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"vdb error {code}: {message}")
        self.code = code


class SynthSpec(C.Structure):
    """Mirror of ``vdb_synth_spec`` (include/vdb.h)."""

    _fields_ = [
        ("seed", C.c_uint64), ("kind", C.c_int32), ("n_clusters", C.c_int32), ("noise", C.c_float),
        ("base_kind", C.c_int32), ("base_seed", C.c_uint64), ("base_n", C.c_int64),
        ("base_n_clusters", C.c_int32), ("base_noise", C.c_float),
    ]


def _declare(lib: C.CDLL) -> C.CDLL:
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    pf, pi64 = C.c_void_p, C.c_void_p  # raw addresses (numpy .ctypes.data / tensor.data_ptr())
    lib.vdb_create.argtypes = [C.POINTER(vp), i32, i32, i32, i32]
    lib.vdb_create_sharded.argtypes = [C.POINTER(vp), i32, i32, i32, C.POINTER(i32), i32]
    lib.vdb_load_sharded.argtypes = [C.POINTER(vp), C.c_char_p, i32, i32, C.POINTER(i32), i32]
    lib.vdb_search_dev_noalloc.argtypes = [vp, pf, i32, i32, i32, pi64, pf, vp]
    lib.vdb_destroy.argtypes = [vp]
    lib.vdb_destroy.restype = None
    lib.vdb_last_error.argtypes = [vp]
    lib.vdb_last_error.restype = C.c_char_p
    lib.vdb_reserve.argtypes = [vp, i64]
    lib.vdb_rebalance.argtypes = [vp, i64]
    lib.vdb_warm.argtypes = [vp, i32, i32, i32]
    lib.vdb_set_row_offset.argtypes = [vp, i64]
    lib.vdb_add.argtypes = [vp, pf, i64]
    lib.vdb_add_dev.argtypes = [vp, pf, i64]
    lib.vdb_count.argtypes = [vp, C.POINTER(i64)]
    lib.vdb_search.argtypes = [vp, pf, i32, i32, i32, pi64, pf]
    lib.vdb_search_fetch.argtypes = [vp, pf, i32, i32, i32, pi64, pf, pf]
    lib.vdb_search_dev.argtypes = [vp, pf, i32, i32, i32, pi64, pf, vp]
    lib.vdb_search_masked.argtypes = [vp, pf, i32, i32, i32, vp, i64, pi64, pf]
    lib.vdb_search_masked_dev.argtypes = [vp, pf, i32, i32, i32, vp, i64, pi64, pf, vp]
    lib.vdb_search_scan_dev.argtypes = [vp, pf, i32, i32, i32, i32, vp]
    lib.vdb_search_finish_dev.argtypes = [vp, i32, pi64, pf, vp]
    lib.vdb_merge_topk_dev.argtypes = [vp, pf, i64, pi64, i64, i32, i32, i32, pi64, pf, vp]
    lib.vdb_selfjoin.argtypes = [vp, i64, i64, i32, i32, i32, pi64, pf]
    lib.vdb_fetch.argtypes = [vp, pi64, i64, pf]
    lib.vdb_save.argtypes = [vp, C.c_char_p]
    lib.vdb_load.argtypes = [C.POINTER(vp), C.c_char_p, i32, i32, i32]
    lib.vdb_last_timing.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_double)]
    lib.vdb_timing_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double), i32]
    lib.vdb_get_stat.argtypes = [vp, C.c_char_p, C.POINTER(i64)]
    lib.vdb_set_option.argtypes = [vp, C.c_char_p, i64]
    lib.vdb_synth_rows_dev.argtypes = [vp, pf, i64, i64, C.POINTER(SynthSpec), vp]
    lib.vdb_add_synth.argtypes = [vp, i64, i64, C.POINTER(SynthSpec)]
    for name in SYMBOLS:
        if name not in ("vdb_destroy", "vdb_last_error"):
            getattr(lib, name).restype = C.c_int
    return lib


_default_lib: Optional[C.CDLL] = None


def load_library(path: Optional[str] = None) -> C.CDLL:
    """Load a library exporting the vdb_* ABI.  ``path=None`` -> the HIP engine."""
    global _default_lib
    if path is None:
        if _default_lib is None:
            if not os.path.exists(HIP_LIB_PATH):
                raise ImportError(
                    f"{HIP_LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "(hipcc --offload-arch=gfx950).  vectordb_amd has no CPU fallback.")
            # PyTorch-ROCm bundles its own HIP runtime.  If THIS library were loaded first it would bind to
            # the system runtime, the process would hold two, and whichever initialises second may find no
            # device (seen on MI355X: torch initialised first, then vdb_create -> hipErrorNoDevice).
            # Importing torch first makes the engine bind to the runtime torch uses: one runtime per process,
            # which is also what lets torch tensors' device pointers be passed to the *_dev entry points.
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
            _default_lib = _declare(C.CDLL(HIP_LIB_PATH))
        return _default_lib
    return _declare(C.CDLL(path))


def check(lib: C.CDLL, handle, rc: int) -> None:
    if rc != 0:
        msg = lib.vdb_last_error(handle)
        raise VdbError(rc, msg.decode("utf-8", "replace") if msg else "unknown error")
