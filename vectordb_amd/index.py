"""FlatIndex — numpy-level handle on one row shard of the exact cosine store.

Thin object wrapper over the C ABI (include/vdb.h).  It plays the role that the
hnswlib vector segment plays inside chromadb for the reference
(src/vectordb.py:149-160,215-219,259-264): it owns the vectors, normalises them
for cosine space and answers k-nearest queries — here exactly, on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _capi
from ._capi import SynthSpec, VdbError  # noqa: F401  (re-exported)

_DTYPES = {"f32": _capi.VDB_F32, "fp32": _capi.VDB_F32, "float32": _capi.VDB_F32,
           "bf16": _capi.VDB_BF16, "bfloat16": _capi.VDB_BF16}


def _as_f32_2d(a, dim: int, what: str) -> np.ndarray:
    arr = np.ascontiguousarray(a, dtype=np.float32)
    if arr.ndim == 1:
        arr = arr.reshape(1, -1)
    if arr.ndim != 2 or arr.shape[1] != dim:
        raise ValueError(f"{what} must have shape (n, {dim}); got {arr.shape}")
    return arr


class FlatIndex:
    """One device-resident shard: rows are L2-normalised at add time and stored
    fp32 (``dtype='f32'``) or bf16 (+ optional fp32 copy for the rerank)."""

    def __init__(self, dim: int, dtype: str = "f32", keep_f32_for_rerank: bool = True,
                 device: int = 0, lib=None, _handle=None, devices: Optional[Sequence[int]] = None):
        """devices=[d0, d1, ...]: ONE handle row-sharded over those GPUs inside this process
        (vdb_create_sharded; an id may repeat).  Call reserve(total_rows) before adding rows: it
        fixes the partition.  Results are bit-identical to a single store of the same rows."""
        if dtype not in _DTYPES:
            raise ValueError(f"dtype must be one of {sorted(_DTYPES)}; got {dtype!r}")
        self._lib = lib if lib is not None else _capi.load_library()
        self.dim = int(dim)
        self.dtype = "bf16" if _DTYPES[dtype] == _capi.VDB_BF16 else "f32"
        self.keep_f32 = bool(keep_f32_for_rerank) or self.dtype == "f32"
        self.devices = [int(d) for d in devices] if devices is not None else None
        self.device = self.devices[0] if self.devices else int(device)
        if _handle is not None:
            self._h = _handle
        else:
            h = C.c_void_p()
            if self.devices is not None:
                ids = (C.c_int * len(self.devices))(*self.devices)
                rc = self._lib.vdb_create_sharded(C.byref(h), self.dim, _DTYPES[dtype], int(self.keep_f32), ids, len(self.devices))
            else:
                rc = self._lib.vdb_create(C.byref(h), self.dim, _DTYPES[dtype], int(self.keep_f32), self.device)
            _capi.check(self._lib, None, rc)
            self._h = h

    # -- lifecycle ---------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.vdb_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int) -> None:
        _capi.check(self._lib, self._h, rc)

    # -- add ---------------------------------------------------------------
    def reserve(self, n_rows: int) -> None:
        self._chk(self._lib.vdb_reserve(self._h, int(n_rows)))

    def rebalance(self, total_rows: int) -> None:
        """Re-partition a row-sharded index for total_rows rows (rows move device to device, bits verbatim)."""
        self._chk(self._lib.vdb_rebalance(self._h, int(total_rows)))

    def warm(self, max_batch: int, k: int, rerank_kprime: int = 0) -> None:
        """Pre-size the workspaces of searches up to max_batch queries, so none grows inside a later search."""
        self._chk(self._lib.vdb_warm(self._h, int(max_batch), int(k), int(rerank_kprime)))

    def set_row_offset(self, off: int) -> None:
        self._chk(self._lib.vdb_set_row_offset(self._h, int(off)))

    def add(self, rows) -> None:
        arr = _as_f32_2d(rows, self.dim, "rows") if np.size(rows) else np.zeros((0, self.dim), np.float32)
        self._chk(self._lib.vdb_add(self._h, arr.ctypes.data, arr.shape[0]))

    def add_device(self, ptr: int, n: int) -> None:
        """Append n fp32 rows already resident on this index's GPU (raw address)."""
        self._chk(self._lib.vdb_add_dev(self._h, int(ptr), int(n)))

    def add_synth(self, row0: int, n: int, spec: SynthSpec) -> None:
        self._chk(self._lib.vdb_add_synth(self._h, int(row0), int(n), C.byref(spec)))

    def synth_rows_device(self, ptr: int, row0: int, n: int, spec: SynthSpec, stream: int = 0) -> None:
        self._chk(self._lib.vdb_synth_rows_dev(self._h, int(ptr), int(row0), int(n), C.byref(spec), stream or None))

    def count(self) -> int:
        n = C.c_int64()
        self._chk(self._lib.vdb_count(self._h, C.byref(n)))
        return int(n.value)

    def __len__(self) -> int:
        return self.count()

    # -- search ------------------------------------------------------------
    def search(self, queries, k: int, rerank_kprime: int = 0) -> Tuple[np.ndarray, np.ndarray]:
        """Exact top-k: returns (idx int64 [b,k], dist fp32 [b,k]); dist = 1 - cosine,
        ascending; rows beyond the store size are padded with -1 / +inf."""
        q = _as_f32_2d(queries, self.dim, "queries")
        b = q.shape[0]
        idx = np.empty((b, k), np.int64)
        dist = np.empty((b, k), np.float32)
        self._chk(self._lib.vdb_search(self._h, q.ctypes.data, b, int(k), int(rerank_kprime),
                                       idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def search_fetch(self, queries, k: int, rerank_kprime: int = 0) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
        """search() plus the stored (normalised fp32) rows of the hits, [b, k, dim], in one round trip
        (zero rows behind an index of -1): collection.query(include=["embeddings"]) as src/vectordb.py:259-264 makes it."""
        q = _as_f32_2d(queries, self.dim, "queries")
        b = q.shape[0]
        idx = np.empty((b, k), np.int64)
        dist = np.empty((b, k), np.float32)
        rows = np.empty((b, k, self.dim), np.float32)
        self._chk(self._lib.vdb_search_fetch(self._h, q.ctypes.data, b, int(k), int(rerank_kprime),
                                             idx.ctypes.data, dist.ctypes.data, rows.ctypes.data))
        return idx, dist, rows

    @staticmethod
    def pack_allow(allow: np.ndarray) -> np.ndarray:
        """The bitmap vdb_search_masked takes: bit r of 32-bit word r/32 (little-endian) = row r is eligible."""
        n = allow.shape[0]
        bits = np.packbits(allow, bitorder="little")
        words = np.zeros(max(1, (n + 31) // 32) * 4, np.uint8)
        words[: bits.size] = bits
        return words

    def search_masked(self, queries, k: int, allow, rerank_kprime: int = 0, packed: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        """Exact top-k over the rows `allow` marks eligible (the `where=` pre-filter of
        collection.query, reference src/vectordb.py:230,259-264).  allow: boolean array,
        one entry per stored row; packed: pack_allow(allow), for callers that keep it."""
        q = _as_f32_2d(queries, self.dim, "queries")
        allow = np.asarray(allow)
        n = self.count()
        if allow.dtype != np.bool_ or allow.shape != (n,):
            raise ValueError(f"allow must be a boolean array of shape ({n},), got {allow.dtype} {allow.shape}")
        words = packed if packed is not None else self.pack_allow(allow)
        if words.dtype != np.uint8 or words.shape != (max(1, (n + 31) // 32) * 4,) or not words.flags.c_contiguous:
            raise ValueError("packed must be pack_allow(allow)")
        b = q.shape[0]
        idx = np.empty((b, k), np.int64)
        dist = np.empty((b, k), np.float32)
        self._chk(self._lib.vdb_search_masked(self._h, q.ctypes.data, b, int(k), int(rerank_kprime), words.ctypes.data, n,
                                              idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def search_masked_device(self, q_ptr: int, b: int, k: int, rerank_kprime: int, allow_bits_ptr: int, n_bits: int,
                             out_idx_ptr: int, out_dist_ptr: int, stream: int = 0) -> None:
        self._chk(self._lib.vdb_search_masked_dev(self._h, int(q_ptr), int(b), int(k), int(rerank_kprime),
                                                  int(allow_bits_ptr), int(n_bits), int(out_idx_ptr), int(out_dist_ptr),
                                                  stream or None))

    def search_scan_device(self, q_ptr: int, b: int, k: int, rerank_kprime: int, slot: int, stream: int = 0) -> bool:
        """Phase 1 of the two-phase search (normalise + corpus scan) into workspace set `slot` (0/1).
        Returns False — nothing done — when the batch does not split (use search_device)."""
        rc = self._lib.vdb_search_scan_dev(self._h, int(q_ptr), int(b), int(k), int(rerank_kprime), int(slot), stream or None)
        if rc == -5:          # VDB_ERR_STATE: not a batch that splits
            return False
        self._chk(rc)
        return True

    def search_finish_device(self, slot: int, out_idx_ptr: int, out_dist_ptr: int, stream: int = 0) -> None:
        """Phase 2 (merge, rerank, ranking) of the scan pending in `slot`; may run on another stream."""
        self._chk(self._lib.vdb_search_finish_dev(self._h, int(slot), int(out_idx_ptr), int(out_dist_ptr), stream or None))

    def search_device(self, q_ptr: int, b: int, k: int, rerank_kprime: int, out_idx_ptr: int,
                      out_dist_ptr: int, stream: int = 0) -> None:
        self._chk(self._lib.vdb_search_dev(self._h, int(q_ptr), int(b), int(k), int(rerank_kprime),
                                           int(out_idx_ptr), int(out_dist_ptr), stream or None))

    def search_device_noalloc(self, q_ptr: int, b: int, k: int, rerank_kprime: int, out_idx_ptr: int,
                              out_dist_ptr: int, stream: int = 0) -> None:
        """search_device for callers that cannot allocate (HIP graph capture, serving loops): at most 8 queries,
        k (k') <= 64; workspaces come from the library's fixed per-GPU scratch array."""
        self._chk(self._lib.vdb_search_dev_noalloc(self._h, int(q_ptr), int(b), int(k), int(rerank_kprime),
                                                   int(out_idx_ptr), int(out_dist_ptr), stream or None))

    def merge_topk_device(self, cand_dist_ptr: int, cand_idx_ptr: int, g: int, b: int, k: int,
                          out_idx_ptr: int, out_dist_ptr: int, stream: int = 0,
                          dist_shard_stride: int = 0, idx_shard_stride: int = 0) -> None:
        """Merge g per-shard result lists ([g][b][k], strides in elements, 0 = dense)."""
        self._chk(self._lib.vdb_merge_topk_dev(self._h, int(cand_dist_ptr), int(dist_shard_stride), int(cand_idx_ptr),
                                               int(idx_shard_stride), int(g), int(b), int(k), int(out_idx_ptr),
                                               int(out_dist_ptr), stream or None))

    def selfjoin(self, k: int, exclude_self: bool = True, rerank_kprime: int = 0,
                 q_row0: int = 0, q_n: Optional[int] = None) -> Tuple[np.ndarray, np.ndarray]:
        if q_n is None:
            q_n = self.count() - q_row0
        idx = np.empty((q_n, k), np.int64)
        dist = np.empty((q_n, k), np.float32)
        self._chk(self._lib.vdb_selfjoin(self._h, int(q_row0), int(q_n), int(k), int(bool(exclude_self)),
                                         int(rerank_kprime), idx.ctypes.data, dist.ctypes.data))
        return idx, dist

    def fetch(self, idx) -> np.ndarray:
        ii = np.ascontiguousarray(idx, dtype=np.int64).reshape(-1)
        out = np.empty((ii.shape[0], self.dim), np.float32)
        self._chk(self._lib.vdb_fetch(self._h, ii.ctypes.data, ii.shape[0], out.ctypes.data))
        return out

    # -- persistence -------------------------------------------------------
    def save(self, directory: str) -> None:
        self._chk(self._lib.vdb_save(self._h, str(directory).encode()))

    @classmethod
    def load(cls, directory: str, dtype: str = "f32", keep_f32_for_rerank: bool = True, device: int = 0,
             lib=None, devices: Optional[Sequence[int]] = None) -> "FlatIndex":
        lib = lib if lib is not None else _capi.load_library()
        h = C.c_void_p()
        if devices is not None:
            ids = (C.c_int * len(devices))(*[int(d) for d in devices])
            rc = lib.vdb_load_sharded(C.byref(h), str(directory).encode(), _DTYPES[dtype], int(bool(keep_f32_for_rerank)), ids, len(devices))
        else:
            rc = lib.vdb_load(C.byref(h), str(directory).encode(), _DTYPES[dtype], int(bool(keep_f32_for_rerank)), int(device))
        _capi.check(lib, None, rc)
        # dim is recorded in the file header (bytes 8..12)
        with open(f"{directory}/vectors.vdb", "rb") as f:
            dim = int(np.frombuffer(f.read(16)[8:12], dtype=np.int32)[0])
        return cls(dim, dtype, keep_f32_for_rerank, device, lib=lib, _handle=h, devices=devices)

    # -- measurement -------------------------------------------------------
    def last_timing(self) -> dict:
        a, b, by = C.c_float(), C.c_float(), C.c_double()
        self._chk(self._lib.vdb_last_timing(self._h, C.byref(a), C.byref(b), C.byref(by)))
        return {"scan_ms": a.value, "merge_ms": b.value, "bytes_streamed": by.value}

    def timing_stats(self, reset: bool = True) -> dict:
        n, a, b = C.c_int64(), C.c_double(), C.c_double()
        self._chk(self._lib.vdb_timing_stats(self._h, C.byref(n), C.byref(a), C.byref(b), int(reset)))
        return {"n": int(n.value), "scan_ms_sum": a.value, "merge_ms_sum": b.value}

    def get_stat(self, name: str) -> int:
        v = C.c_int64()
        self._chk(self._lib.vdb_get_stat(self._h, name.encode(), C.byref(v)))
        return int(v.value)

    def set_option(self, name: str, value: int) -> None:
        self._chk(self._lib.vdb_set_option(self._h, name.encode(), int(value)))
