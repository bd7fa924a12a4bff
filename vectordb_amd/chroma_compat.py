"""Chroma-shaped client/collection over the MI355X exact cosine engine.

Reproduces exactly the slice of the chromadb API the reference touches
("Complete Chroma API surface"):

  chromadb.PersistentClient(path=)                      src/vectordb.py:149
  client.get_or_create_collection(name=, embedding_function=,
        metadata={"hnsw:space": "cosine"})              src/vectordb.py:156,183
  client.delete_collection(name=)                       src/vectordb.py:180
  client.get_collection(name) / collection.count()      tests/test_vectordb.py:98,101
  collection.add(embeddings=, metadatas=, ids=)         src/vectordb.py:215
  collection.query(query_embeddings=, n_results=, include=, where=)
        src/vectordb.py:259; analysis/bidirectional_similarity.py:110;
        analysis/run_bidirectional_test.py:139; analysis/generate_best_usda_matches.py:196
  collection.get(ids=, include=)                        analysis/generate_best_usda_matches.py:185

The vector segment (chromadb's HNSW index) is a FlatIndex on the GPU; ids and
metadatas stay on the host, the role sqlite plays inside chromadb.  Embeddings
are returned from the store itself (vdb_fetch): the L2-NORMALISED row (bf16-rounded
on a bf16 store without the fp32 copy), not the embedding as it was added — no second
copy is kept in host memory.  Whether chromadb hands back normalised rows depends on
its version (parity unpinned: chromadb is not importable here); the reference uses
the returned embeddings only inside cosine formulas (src/vectordb.py:288-292,
analysis/bidirectional_similarity.py:170-173), which do not depend on their scale.  Search is exact, so results equal Chroma's only up to
HNSW's approximation error.

Persistence is incremental (the reference adds 14 777 rows in three batches,
src/vectordb.py:205-219, and chromadb itself persists per add): an add APPENDS
its records to `records.jsonl` and its embeddings, as given, to `embeddings.f32`
and then replaces the small `collection.json` whose `count` is the commit point
(a torn append behind it is ignored on reopen).  The engine's own file
`vectors.vdb` is only a cache for a fast reopen: it is written by persist() /
close() when rows were added since, never per add, and a missing or stale one is
rebuilt from `embeddings.f32`.
"""
from __future__ import annotations

import json
import os
import shutil
from collections import OrderedDict
from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from ._capi import VdbError
from .index import FlatIndex

_VALID_INCLUDE = {"embeddings", "metadatas", "documents", "distances", "uris", "data"}
_DEFAULT_QUERY_INCLUDE = ("metadatas", "documents", "distances")
_DEFAULT_GET_INCLUDE = ("metadatas", "documents")


def _match_where(meta: Optional[dict], where: dict) -> bool:
    """Chroma metadata filter: {"k": v}, {"k": {"$eq"|"$ne"|"$in"|"$nin"|"$gt"|"$gte"|"$lt"|"$lte": v}},
    {"$and": [...]}, {"$or": [...]}."""
    meta = meta or {}
    for key, cond in where.items():
        if key == "$and":
            if not all(_match_where(meta, w) for w in cond):
                return False
        elif key == "$or":
            if not any(_match_where(meta, w) for w in cond):
                return False
        elif isinstance(cond, dict):
            if key not in meta:
                return False
            v = meta[key]
            for op, ref in cond.items():
                ok = {"$eq": lambda: v == ref, "$ne": lambda: v != ref, "$in": lambda: v in ref,
                      "$nin": lambda: v not in ref, "$gt": lambda: v > ref, "$gte": lambda: v >= ref,
                      "$lt": lambda: v < ref, "$lte": lambda: v <= ref}.get(op)
                if ok is None:
                    raise ValueError(f"Expected where operator to be one of $eq,$ne,$in,$nin,$gt,$gte,$lt,$lte; got {op}")
                if not ok():
                    return False
        else:
            if meta.get(key, object()) != cond:
                return False
    return True


class Collection:
    """Cosine-space collection.  ``embeddings_as='numpy'`` mirrors recent chromadb
    (query: list of 2-D arrays; get: one 2-D array), which is what the reference's
    saved output was produced with (analysis_results/best_usda_matches.csv:2 holds
    the list-valued product_id that only that return shape produces); 'list' gives
    nested Python lists as in chromadb 0.4.x."""

    def __init__(self, name: str, path: Optional[str], embedding_function=None, metadata: Optional[dict] = None,
                 *, dtype: str = "f32", device: int = 0, rerank_kprime: Optional[int] = None, lib=None,
                 embeddings_as: str = "numpy", auto_persist: bool = True, fp32_copy: bool = True,
                 devices: Optional[Sequence[int]] = None, expected_rows: Optional[int] = None):
        """devices=[d0, d1, ...]: the vector segment is ONE store row-sharded over those GPUs inside this process
        (vdb_create_sharded) — same calls, same answers; the partition follows the collection as it grows
        (vdb_rebalance at every doubling; expected_rows sizes it up front)."""
        metadata = dict(metadata or {})
        space = metadata.get("hnsw:space", "l2")
        if space != "cosine":
            raise ValueError(f"only the cosine space is implemented (metadata={{'hnsw:space': 'cosine'}}); got {space!r}")
        self.name = name
        self.metadata = metadata
        self._path = path
        self._embedding_function = embedding_function
        self._dtype = dtype
        self._devices = [int(d) for d in devices] if devices else None
        self._device = self._devices[0] if self._devices else device
        self._expected_rows = int(expected_rows or 0)
        self._partition_rows = 0                  # rows the sharded handle's partition was last made for
        self._lib = lib
        self._rerank_kprime = rerank_kprime
        self._fp32_copy = bool(fp32_copy) or dtype != "bf16"      # bf16 only: 2 instead of 6 bytes per element; the rerank and
        self._embeddings_as = embeddings_as                        # include=["embeddings"] then work from the stored bf16 rows
        self._auto_persist = auto_persist and path is not None
        self._ids: List[str] = []
        self._id_to_row: Dict[str, int] = {}
        self._metadatas: List[Optional[dict]] = []
        self._documents: List[Optional[str]] = []
        self._index: Optional[FlatIndex] = None
        self._dim: Optional[int] = None
        self._vdb_rows = -1                       # rows in the vectors.vdb on disk (-1: none / unknown)
        self._rec_bytes: Optional[int] = 0        # committed length of records.jsonl (None: unknown)
        self._where_cache: "OrderedDict[str, Tuple[np.ndarray, int, Optional[np.ndarray]]]" = OrderedDict()

    # -- internals ---------------------------------------------------------
    def _ensure_index(self, dim: int) -> FlatIndex:
        if self._index is None:
            self._dim = dim
            self._index = FlatIndex(dim, self._dtype, self._fp32_copy, self._device, lib=self._lib, devices=self._devices)
            # nothing on this face reads vdb_last_timing: leave the three timing events per search out of the stream
            self._index.set_option("timing_every", 65536)
        elif dim != self._dim:
            raise ValueError(f"Embedding dimension {dim} does not match collection dimensionality {self._dim}")
        return self._index

    def _ensure_partition(self, total: int) -> None:
        """Sharded segment: keep the row partition ahead of the collection's size (amortised doubling; the rows move
        between the GPUs as they are, so answers do not change)."""
        if self._devices and self._index is not None and total > self._partition_rows:
            want = max(2 * total if self._partition_rows else total, self._expected_rows, 256 * len(self._devices))
            self._index.rebalance(want)
            self._partition_rows = want

    def _stored_rows(self, rows) -> np.ndarray:
        """The stored (L2-normalised, fp32) embeddings of the given row indices, from the engine."""
        rows = np.asarray(rows, dtype=np.int64).reshape(-1)
        if rows.size == 0 or self._index is None:
            return np.zeros((0, self._dim or 0), np.float32)
        return self._index.fetch(rows)

    def _kprime(self, k: int) -> int:
        if self._dtype != "bf16":
            return 0
        kp = self._rerank_kprime if self._rerank_kprime is not None else max(4 * k, 32)
        return int(min(max(kp, k), 1024))

    def _embed(self, texts: Sequence[str]) -> np.ndarray:
        if self._embedding_function is None:
            raise ValueError("You must provide embeddings (no embedding_function is set on this collection)")
        return np.asarray(self._embedding_function(list(texts)), dtype=np.float32)

    def _format_embeddings(self, mat: np.ndarray):
        return mat if self._embeddings_as == "numpy" else mat.tolist()

    # -- chroma API --------------------------------------------------------
    def count(self) -> int:
        return len(self._ids)

    def add(self, ids, embeddings=None, metadatas=None, documents=None) -> None:
        ids = [ids] if isinstance(ids, str) else list(ids)
        n = len(ids)
        if len(set(ids)) != n:
            dup = sorted({i for i in ids if ids.count(i) > 1})[:5]
            raise ValueError(f"Expected IDs to be unique, found duplicates of: {', '.join(dup)}")
        if embeddings is None:
            if documents is None:
                raise ValueError("You must provide either embeddings or documents")
            embeddings = self._embed([documents] if isinstance(documents, str) else documents)
        dev_emb = None
        if getattr(embeddings, "is_cuda", False) and self._lib is None:
            # a torch tensor already on the GPU (an embedder's output): it is handed to the engine in place
            # (vdb_add_dev); the host copy below only feeds get(include=["embeddings"]) and persist()
            dev_emb = embeddings.detach().float().contiguous()
            if dev_emb.ndim == 1:
                dev_emb = dev_emb.reshape(1, -1)
            emb = dev_emb.cpu().numpy()
        else:
            if hasattr(embeddings, "detach"):
                embeddings = embeddings.detach().cpu().numpy()
            emb = np.asarray(embeddings, dtype=np.float32)
        if emb.ndim == 1:
            emb = emb.reshape(1, -1)
        if emb.ndim != 2 or emb.shape[0] != n:
            raise ValueError(f"Number of embeddings {emb.shape[0] if emb.ndim == 2 else '?'} must match number of ids {n}")
        for name, seq in (("metadatas", metadatas), ("documents", documents)):
            if seq is not None and not isinstance(seq, (str, dict)) and len(seq) != n:
                raise ValueError(f"Number of {name} {len(seq)} must match number of ids {n}")
        if n == 0:
            return
        if isinstance(metadatas, dict):
            metadatas = [metadatas]
        if isinstance(documents, str):
            documents = [documents]
        # chromadb ignores ids that already exist on add()
        keep = [j for j, i in enumerate(ids) if i not in self._id_to_row]
        if not keep:
            return
        if len(keep) != n:
            emb = emb[keep]
            if dev_emb is not None:
                dev_emb = dev_emb[keep].contiguous()
        index = self._ensure_index(emb.shape[1])
        self._ensure_partition(len(self._ids) + emb.shape[0])
        if dev_emb is not None and dev_emb.device.index == index.device:
            import torch
            torch.cuda.current_stream(dev_emb.device).synchronize()     # the engine ingests on its own stream
            index.add_device(dev_emb.data_ptr(), dev_emb.shape[0])
        else:
            index.add(emb)
        base = len(self._ids)
        for off, j in enumerate(keep):
            self._id_to_row[ids[j]] = base + off
            self._ids.append(ids[j])
            self._metadatas.append(dict(metadatas[j]) if metadatas is not None and metadatas[j] is not None else None)
            self._documents.append(documents[j] if documents is not None else None)
        if self._auto_persist:
            self._append_to_disk(base, np.ascontiguousarray(emb[: len(keep)], dtype=np.float32))

    def query(self, query_embeddings=None, query_texts=None, n_results: int = 10, where: Optional[dict] = None,
              where_document: Optional[dict] = None, include: Sequence[str] = _DEFAULT_QUERY_INCLUDE) -> Dict[str, Any]:
        if where_document:
            raise NotImplementedError("where_document filtering is not implemented")
        include = list(include)
        bad = [i for i in include if i not in _VALID_INCLUDE]
        if bad:
            raise ValueError(f"Expected include item to be one of {sorted(_VALID_INCLUDE)}, got {bad[0]}")
        if query_embeddings is None:
            if query_texts is None:
                raise ValueError("You must provide one of query_embeddings or query_texts")
            query_embeddings = self._embed([query_texts] if isinstance(query_texts, str) else query_texts)
        if not isinstance(n_results, int) or isinstance(n_results, bool) or n_results <= 0:
            raise ValueError(f"Number of requested results {n_results} must be a positive integer")
        q = np.asarray(query_embeddings, dtype=np.float32)
        if q.ndim == 1:
            q = q.reshape(1, -1)
        b = q.shape[0]
        total = self.count()
        if total and q.shape[1] != self._dim:
            raise ValueError(f"Embedding dimension {q.shape[1]} does not match collection dimensionality {self._dim}")

        rows: List[np.ndarray]
        dists: List[np.ndarray]
        hit_rows = None
        if total == 0:
            rows = [np.zeros(0, np.int64) for _ in range(b)]
            dists = [np.zeros(0, np.float32) for _ in range(b)]
        elif where:
            rows, dists = self._query_filtered(q, n_results, where)
        else:
            k = min(n_results, total)           # chromadb clamps n_results to the collection size
            if "embeddings" in include and b * k * (self._dim or 0) * 4 <= (256 << 20):
                # the reference's hot call (src/vectordb.py:259-264): hits AND their stored rows in one round trip
                idx, dist, hit_rows = self._index.search_fetch(q, k, self._kprime(k))
            else:
                idx, dist = self._index.search(q, k, self._kprime(k))
            rows = [idx[i][idx[i] >= 0] for i in range(b)]
            dists = [dist[i][: len(rows[i])] for i in range(b)]

        row_lists = [rr.tolist() for rr in rows]          # plain ints index Python lists ~3x faster than numpy scalars
        ids_, metas_, docs_ = self._ids, self._metadatas, self._documents
        out: Dict[str, Any] = {"ids": [[ids_[r] for r in rl] for rl in row_lists],
                               "distances": None, "metadatas": None, "embeddings": None, "documents": None,
                               "uris": None, "data": None, "included": include}
        if "distances" in include:
            out["distances"] = [np.asarray(dd, dtype=np.float64).tolist() for dd in dists]
        if "metadatas" in include:
            out["metadatas"] = [[metas_[r] for r in rl] for rl in row_lists]
        if "documents" in include:
            out["documents"] = [[docs_[r] for r in rl] for rl in row_lists]
        if "embeddings" in include:
            if hit_rows is not None:
                out["embeddings"] = [self._format_embeddings(hit_rows[i, : len(rows[i])]) for i in range(b)]
            else:
                flat = self._stored_rows(np.concatenate(rows) if rows else np.zeros(0, np.int64))   # one fetch for the batch
                cuts = np.cumsum([0] + [len(rr) for rr in rows])
                out["embeddings"] = [self._format_embeddings(flat[cuts[i]:cuts[i + 1]]) for i in range(b)]
        return out

    def _where_mask(self, where: dict):
        """(eligibility mask, eligible rows, packed bitmap) of a `where=` predicate.  The predicate is evaluated row by row
        in Python (≈ 0.3 µs per row and clause — 4 ms at the reference's 14 777 products, far more than the search), so the
        last few predicates' masks are kept: rows are only ever appended, so a kept mask stays valid for the rows it
        covers and only the rows added since are evaluated.  (Metadata dicts are the collection's own: callers who
        mutate one they got from get() in place bypass this, as they bypass persistence.)"""
        n = len(self._metadatas)
        try:
            key = json.dumps(where, sort_keys=True)
        except (TypeError, ValueError):
            key = None
        ent = self._where_cache.get(key) if key is not None else None
        if ent is not None and ent[0].size == n:
            self._where_cache.move_to_end(key)
            return ent
        if ent is not None and ent[0].size < n:
            old = ent[0]
            tail = np.fromiter((_match_where(m, where) for m in self._metadatas[old.size:]), dtype=np.bool_, count=n - old.size)
            mask = np.concatenate([old, tail])
        else:
            mask = np.fromiter((_match_where(m, where) for m in self._metadatas), dtype=np.bool_, count=n)
        ent = (mask, int(np.count_nonzero(mask)), FlatIndex.pack_allow(mask))
        if key is not None:
            self._where_cache[key] = ent
            self._where_cache.move_to_end(key)
            while len(self._where_cache) > 16:
                self._where_cache.popitem(last=False)
        return ent

    def _query_filtered(self, q: np.ndarray, n_results: int, where: dict):
        """Metadata pre-filter (exact), Chroma's `where=` (src/vectordb.py:230,259-264): the
        predicate is evaluated on the host into an eligibility bitmap and the scan skips
        ineligible rows (vdb_search_masked).  A large batch against a very small eligible
        set is cheaper as a temporary shard of just those rows (uploading a row costs
        about 250 scanned rows; the masked scan serves 4 queries per corpus pass)."""
        mask, n_allowed, packed = self._where_mask(where)
        b = q.shape[0]
        if n_allowed == 0:
            return [np.zeros(0, np.int64)] * b, [np.zeros(0, np.float32)] * b
        passes = (b + 3) // 4      # up to 4 queries are one masked pass, which reads only the tiles with an eligible row
        if passes == 1 or 250 * n_allowed >= passes * mask.size:
            k = min(n_results, n_allowed)
            idx, dist = self._index.search_masked(q, k, mask, self._kprime(k), packed=packed)
            rows = [idx[i][idx[i] >= 0] for i in range(b)]
            return rows, [dist[i][: len(rows[i])] for i in range(b)]
        allowed = np.flatnonzero(mask)
        sub = FlatIndex(self._dim, self._dtype, self._fp32_copy, self._device, lib=self._lib)
        try:
            sub.add(self._stored_rows(allowed))
            k = min(n_results, allowed.size)
            idx, dist = sub.search(q, k, self._kprime(k))
        finally:
            sub.close()
        rows = [allowed[idx[i][idx[i] >= 0]] for i in range(b)]
        return rows, [dist[i][: len(rows[i])] for i in range(b)]

    def get(self, ids=None, where: Optional[dict] = None, limit: Optional[int] = None, offset: Optional[int] = None,
            include: Sequence[str] = _DEFAULT_GET_INCLUDE) -> Dict[str, Any]:
        include = list(include)
        if ids is None:
            rows = list(range(self.count()))
        else:
            ids = [ids] if isinstance(ids, str) else list(ids)
            rows = [self._id_to_row[i] for i in ids if i in self._id_to_row]  # unknown ids are dropped
        if where:
            rows = [r for r in rows if _match_where(self._metadatas[r], where)]
        if offset:
            rows = rows[offset:]
        if limit is not None:
            rows = rows[:limit]
        out: Dict[str, Any] = {"ids": [self._ids[r] for r in rows], "embeddings": None, "metadatas": None,
                               "documents": None, "uris": None, "data": None, "included": include}
        if "metadatas" in include:
            out["metadatas"] = [self._metadatas[r] for r in rows]
        if "documents" in include:
            out["documents"] = [self._documents[r] for r in rows]
        if "embeddings" in include:
            out["embeddings"] = self._format_embeddings(self._stored_rows(rows))
        return out

    def peek(self, limit: int = 10) -> Dict[str, Any]:
        return self.get(limit=limit, include=["embeddings", "metadatas", "documents"])

    # -- persistence (flat store) ----------------------
    def _dir(self) -> str:
        return os.path.join(self._path, self.name)

    def _write_manifest(self, d: str) -> None:
        tmp = os.path.join(d, "collection.json.tmp")
        with open(tmp, "w") as f:
            json.dump({"name": self.name, "metadata": self.metadata, "dim": self._dim, "count": self.count()}, f)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, os.path.join(d, "collection.json"))      # the commit point

    def _append_to_disk(self, base: int, emb: np.ndarray) -> None:
        """Persist one add: O(rows added), whatever the collection's size."""
        d = self._dir()
        os.makedirs(d, exist_ok=True)
        rec_path, emb_path = os.path.join(d, "records.jsonl"), os.path.join(d, "embeddings.f32")
        # a torn append of an earlier run may sit behind the committed count: cut it off before appending
        if base == 0 or not os.path.exists(rec_path):
            open(rec_path, "w").close()
            open(emb_path, "wb").close()
        elif self._rec_bytes is not None and os.path.getsize(rec_path) != self._rec_bytes:
            os.truncate(rec_path, self._rec_bytes)
        want = base * (self._dim or 0) * 4
        if os.path.exists(emb_path) and os.path.getsize(emb_path) != want:
            os.truncate(emb_path, want)
        with open(rec_path, "a") as f:
            for r in range(base, len(self._ids)):
                f.write(json.dumps({"id": self._ids[r], "metadata": self._metadatas[r], "document": self._documents[r]}) + "\n")
            f.flush()
            os.fsync(f.fileno())
            self._rec_bytes = f.tell()
        with open(emb_path, "ab") as f:
            emb.tofile(f)
            f.flush()
            os.fsync(f.fileno())
        self._write_manifest(d)

    def persist(self) -> None:
        """Bring the directory fully up to date: the manifest, and the engine's store file (a cache for a fast reopen)
        when rows were added since it was written.  add() has already made the records and embeddings durable."""
        if self._path is None:
            return
        d = self._dir()
        os.makedirs(d, exist_ok=True)
        if not self._auto_persist and self.count():     # nothing was appended per add: write the whole collection once
            self._rec_bytes = None
            n = self.count()
            tmp_r, tmp_e = os.path.join(d, "records.jsonl.tmp"), os.path.join(d, "embeddings.f32.tmp")
            with open(tmp_r, "w") as f:
                for i, m, doc in zip(self._ids, self._metadatas, self._documents):
                    f.write(json.dumps({"id": i, "metadata": m, "document": doc}) + "\n")
            with open(tmp_e, "wb") as f:
                for lo in range(0, n, 65536):
                    self._stored_rows(np.arange(lo, min(n, lo + 65536))).tofile(f)
            os.replace(tmp_r, os.path.join(d, "records.jsonl"))
            os.replace(tmp_e, os.path.join(d, "embeddings.f32"))
        self._write_manifest(d)
        if self._index is not None and self._vdb_rows != self.count():
            self._index.save(d)
            self._vdb_rows = self.count()

    def _load(self) -> None:
        d = self._dir()
        with open(os.path.join(d, "collection.json")) as f:
            info = json.load(f)
        self.metadata = info.get("metadata") or self.metadata
        self._dim = info.get("dim")
        count = int(info.get("count") or 0)
        rec_path = os.path.join(d, "records.jsonl")
        self._rec_bytes = 0
        if count:
            with open(rec_path, "rb") as f:
                for _ in range(count):               # records behind the committed count are a torn append: ignored
                    line = f.readline()
                    if not line.endswith(b"\n"):
                        raise RuntimeError(f"corrupt collection {self.name}: records.jsonl holds fewer than {count} records")
                    rec = json.loads(line)
                    self._id_to_row[rec["id"]] = len(self._ids)
                    self._ids.append(rec["id"])
                    self._metadatas.append(rec["metadata"])
                    self._documents.append(rec["document"])
                self._rec_bytes = f.tell()
        if not (self._dim and count):
            return
        vdb = os.path.join(d, "vectors.vdb")
        if os.path.exists(vdb):
            with open(vdb, "rb") as f:
                head = f.read(32)
            if len(head) == 32 and int(np.frombuffer(head[16:24], dtype=np.int64)[0]) == count:
                try:
                    self._index = FlatIndex.load(d, self._dtype, self._fp32_copy, self._device, lib=self._lib, devices=self._devices)
                    self._index.set_option("timing_every", 65536)
                    self._vdb_rows = count
                    self._partition_rows = count          # vdb_load_sharded partitions for the file's row count
                    return
                except VdbError as exc:
                    # a cache this configuration cannot use (written without the fp32 copy and reopened with one:
                    # VDB_ERR_STATE) or a damaged one (VDB_ERR_IO): rebuild below.  Device and memory errors are the caller's.
                    if exc.code not in (-5, -4):
                        raise
                    self._index = None
        # no (or a stale) store file: rebuild the index from the embeddings as they were added
        emb_path, npy_path = os.path.join(d, "embeddings.f32"), os.path.join(d, "embeddings.npy")
        if os.path.exists(emb_path) and os.path.getsize(emb_path) >= count * self._dim * 4:
            raw = np.memmap(emb_path, dtype=np.float32, mode="r", shape=(count, self._dim))
        elif os.path.exists(npy_path):                # directories written by round 1
            raw = np.load(npy_path, allow_pickle=False)[:count]
            raw.astype(np.float32).tofile(emb_path)
        else:
            raise RuntimeError(f"corrupt collection {self.name}: {count} records but no embeddings on disk")
        index = self._ensure_index(self._dim)
        if self._devices:
            self._ensure_partition(count)
        else:
            index.reserve(count)
        for lo in range(0, count, 65536):
            index.add(np.ascontiguousarray(raw[lo:lo + 65536]))
        if index.count() != count:
            raise RuntimeError(f"corrupt collection {self.name}: {index.count()} vectors, {count} records")

    def _close(self) -> None:
        if self._index is not None:
            if self._auto_persist and self._vdb_rows != self.count():
                try:
                    self.persist()
                except Exception:          # a failed cache write must not mask the caller's own error path
                    pass
            self._index.close()
            self._index = None

    def __del__(self):  # pragma: no cover - best effort: refresh the reopen cache when the last reference goes
        try:
            self._close()
        except Exception:
            pass


class PersistentClient:
    """Drop-in for ``chromadb.PersistentClient`` (src/vectordb.py:149).  Extra
    keyword arguments select the engine configuration and are not part of Chroma."""

    def __init__(self, path: str = "./chroma", *, dtype: str = "f32", device: int = 0,
                 rerank_kprime: Optional[int] = None, lib=None, embeddings_as: str = "numpy",
                 auto_persist: bool = True, fp32_copy: bool = True, devices: Optional[Sequence[int]] = None,
                 expected_rows: Optional[int] = None):
        self._path = str(path)
        os.makedirs(self._path, exist_ok=True)
        self._opts = dict(dtype=dtype, device=device, rerank_kprime=rerank_kprime, lib=lib,
                          embeddings_as=embeddings_as, auto_persist=auto_persist, fp32_copy=fp32_copy,
                          devices=list(devices) if devices else None, expected_rows=expected_rows)
        self._open: Dict[str, Collection] = {}

    def _exists(self, name: str) -> bool:
        return name in self._open or os.path.exists(os.path.join(self._path, name, "collection.json"))

    def list_collections(self) -> List[str]:
        names = set(self._open)
        for n in os.listdir(self._path):
            if os.path.exists(os.path.join(self._path, n, "collection.json")):
                names.add(n)
        return sorted(names)

    def get_collection(self, name: str, embedding_function=None) -> Collection:
        if name in self._open:
            return self._open[name]
        if not self._exists(name):
            raise ValueError(f"Collection {name} does not exist.")
        col = Collection(name, self._path, embedding_function, {"hnsw:space": "cosine"}, **self._opts)
        col._load()
        self._open[name] = col
        return col

    def create_collection(self, name: str, embedding_function=None, metadata: Optional[dict] = None,
                          get_or_create: bool = False) -> Collection:
        if self._exists(name):
            if get_or_create:
                col = self.get_collection(name, embedding_function)
                if embedding_function is not None:
                    col._embedding_function = embedding_function
                return col
            raise ValueError(f"Collection {name} already exists")
        col = Collection(name, self._path, embedding_function, metadata, **self._opts)
        self._open[name] = col
        if col._auto_persist:
            col.persist()
        return col

    def get_or_create_collection(self, name: str, embedding_function=None, metadata: Optional[dict] = None) -> Collection:
        return self.create_collection(name, embedding_function, metadata, get_or_create=True)

    def delete_collection(self, name: str) -> None:
        if not self._exists(name):
            raise ValueError(f"Collection {name} does not exist.")
        col = self._open.pop(name, None)
        if col is not None:
            col._auto_persist = False         # about to be removed: no point refreshing its store file
            col._close()
        shutil.rmtree(os.path.join(self._path, name), ignore_errors=True)

    def close(self) -> None:
        """Write the store files of collections that grew since they were last written, and release the GPU memory.
        Not part of Chroma's API (chromadb persists per add; so does this client — close() only refreshes the
        fast-reopen cache)."""
        for col in list(self._open.values()):
            col._close()
        self._open.clear()


class EphemeralClient(PersistentClient):
    """In-memory variant (no files)."""

    def __init__(self, **kw):
        kw.pop("auto_persist", None)
        self._path = None
        self._opts = dict(dtype=kw.get("dtype", "f32"), device=kw.get("device", 0),
                          rerank_kprime=kw.get("rerank_kprime"), lib=kw.get("lib"),
                          embeddings_as=kw.get("embeddings_as", "numpy"), auto_persist=False, fp32_copy=kw.get("fp32_copy", True),
                          devices=list(kw["devices"]) if kw.get("devices") else None, expected_rows=kw.get("expected_rows"))
        self._open = {}

    def _exists(self, name: str) -> bool:
        return name in self._open

    def list_collections(self) -> List[str]:
        return sorted(self._open)

    def delete_collection(self, name: str) -> None:
        if name not in self._open:
            raise ValueError(f"Collection {name} does not exist.")
        self._open.pop(name)._close()
