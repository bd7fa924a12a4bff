"""Python face of the CPU oracle.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg — never by vectordb_amd.

PARITY UNPINNED against the reference's own outputs (it holds none for this
path).  Two parts:

  * ctypes wrapper over oracle/_build/libvdb_oracle.so (exact fp64-accumulate
    cosine top-k, the synthetic generator, and the vdb_* ABI on the CPU);
  * line-by-line numpy/pandas restatements of the reference's host glue:
      get_similar_products rerank      src/vectordb.py:277-335
      bidirectional candidate filter   analysis/bidirectional_similarity.py:140-194
      find_best_usda_match             analysis/generate_best_usda_matches.py:116-143
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_DIR, "_build", "libvdb_oracle.so")


def build(force: bool = False) -> str:
    src = os.path.join(_DIR, "vdb_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _DIR, "-s"])
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    """The oracle library with the vdb_* ABI declared (usable as FlatIndex(lib=...))."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        from vectordb_amd import _capi
        _lib = _capi.load_library(LIB_PATH)
        vp, i64 = C.c_void_p, C.c_int64
        _lib.ora_synth_rows.argtypes = [vp, i64, i64, C.c_int, C.POINTER(_capi.SynthSpec)]
        _lib.ora_synth_rows.restype = None
        _lib.ora_synth_query_src.argtypes = [C.POINTER(_capi.SynthSpec), i64]
        _lib.ora_synth_query_src.restype = i64
        _lib.ora_normalize_rows.argtypes = [vp, vp, i64, C.c_int]
        _lib.ora_normalize_rows.restype = None
        _lib.ora_round_bf16.argtypes = [vp, vp, i64]
        _lib.ora_round_bf16.restype = None
        _lib.ora_search.argtypes = [vp, i64, C.c_int, vp, C.c_int, C.c_int, vp, vp, vp]
        _lib.ora_search.restype = None
        _lib.ora_num_threads.restype = C.c_int
    return _lib


def synth_rows(row0: int, n: int, dim: int, spec) -> np.ndarray:
    out = np.empty((n, dim), np.float32)
    lib().ora_synth_rows(out.ctypes.data, row0, n, dim, C.byref(spec))
    return out


def synth_query_src(spec, row: int) -> int:
    return int(lib().ora_synth_query_src(C.byref(spec), row))


def normalize_rows(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().ora_normalize_rows(x.ctypes.data, out.ctypes.data, x.shape[0], x.shape[1])
    return out


def round_bf16(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty_like(x)
    lib().ora_round_bf16(x.ctypes.data, out.ctypes.data, x.size)
    return out


def exact_topk(rows_norm: np.ndarray, queries: np.ndarray, k: int):
    """Exact cosine top-k (fp32 storage, fp64 accumulate; score desc then index asc).
    rows_norm must already be L2-normalised; queries are raw.  Returns
    (idx int64 [b,k], dist fp32 [b,k], score fp64 [b,k])."""
    rows_norm = np.ascontiguousarray(rows_norm, np.float32)
    queries = np.ascontiguousarray(queries, np.float32).reshape(-1, rows_norm.shape[1])
    b = queries.shape[0]
    idx = np.empty((b, k), np.int64)
    dist = np.empty((b, k), np.float32)
    sc = np.empty((b, k), np.float64)
    lib().ora_search(rows_norm.ctypes.data, rows_norm.shape[0], rows_norm.shape[1], queries.ctypes.data, b, k,
                     idx.ctypes.data, dist.ctypes.data, sc.ctypes.data)
    return idx, dist, sc


def exact_topk_numpy(rows_norm: np.ndarray, queries: np.ndarray, k: int):
    """Independent numpy statement of the same thing (cross-checks the C code)."""
    d = rows_norm.astype(np.float64)
    q = queries.astype(np.float64).reshape(-1, d.shape[1])
    qn = (q / (np.sqrt((q * q).sum(1, keepdims=True)) + 1e-30)).astype(np.float32).astype(np.float64)
    s = qn @ d.T
    n = d.shape[0]
    order = np.lexsort((np.broadcast_to(np.arange(n), s.shape), -s), axis=1)[:, :k]
    sc = np.take_along_axis(s, order, 1)
    return order.astype(np.int64), (1.0 - sc).astype(np.float32), sc


# ---------------------------------------------------------------------------
# host-glue restatements (loops kept as in the reference, for readability of parity)
# ---------------------------------------------------------------------------
def get_similar_products_rerank(ids, forward_distances, metadatas, embeddings, query_embedding,
                                n_results=5, similarity_threshold=None):
    """src/vectordb.py:277-335."""
    import pandas as pd
    query_embedding_array = np.array(query_embedding)
    forward_similarities = [1 - d for d in forward_distances]                      # :277
    rows = []
    for i, item_id in enumerate(ids):                                              # :282
        item = np.array(embeddings[i])
        dot = np.dot(item.astype(np.float64), query_embedding_array.astype(np.float64))   # :288 (fp64)
        ni = np.linalg.norm(item.astype(np.float64))                                # :289
        nq = np.linalg.norm(query_embedding_array.astype(np.float64))               # :290
        back = dot / (ni * nq) if ni * nq != 0 else 0                              # :292
        fwd = forward_similarities[i]
        rows.append({"id": item_id, "forward_similarity": fwd, "backward_similarity": back,
                     "bi_directional_similarity": (fwd + back) / 2,               # :296
                     "distance": forward_distances[i], **metadatas[i]})
    df = pd.DataFrame(rows)
    df = df.sort_values("bi_directional_similarity", ascending=False)              # :314
    if similarity_threshold is not None:
        df = df[df["bi_directional_similarity"] >= similarity_threshold]           # :319
    if len(df) > n_results:
        df = df.head(n_results)                                                    # :323-324
    expected = ["id", "forward_similarity", "backward_similarity", "bi_directional_similarity",
                "distance", "product_description", "product_code", "usda_code"]    # :327-328
    for col in expected:
        if col not in df.columns:
            df[col] = "N/A"
    return df[expected + [c for c in df.columns if c not in expected]]             # :335


def bidirectional_filter(query_description, ids, distances, metadatas, embeddings, documents,
                         query_embedding_text_only, n_results_final=5, thr_fwd=0.5, thr_bwd=0.5):
    """analysis/bidirectional_similarity.py:140-194 (sklearn cosine_similarity for the backward side)."""
    from sklearn.metrics.pairwise import cosine_similarity
    norm = lambda d: d.lower().strip() if isinstance(d, str) else str(d).strip()   # :18-29
    query_desc_norm = norm(query_description)                                      # :140
    out = []
    for i, _cid in enumerate(ids):                                                 # :147
        fwd = 1 - distances[i]                                                     # :149
        if fwd < thr_fwd:                                                          # :152
            continue
        meta = metadatas[i]
        cand_desc = meta.get("product_description", documents[i] if i < len(documents) else "Unknown")  # :158
        if norm(cand_desc) == query_desc_norm:                                     # :161
            continue
        back = cosine_similarity(np.array(embeddings[i]).reshape(1, -1),
                                 np.asarray(query_embedding_text_only).reshape(1, -1))[0][0]   # :170-173
        if fwd >= thr_fwd and back >= thr_bwd:                                     # :179
            out.append({"query_description": query_description, "candidate_description": cand_desc,
                        "forward_similarity": fwd, "backward_similarity": back,
                        "avg_similarity": (fwd + back) / 2, "metadata": meta})     # :181-189
    out.sort(key=lambda x: x["avg_similarity"], reverse=True)                      # :193
    return out[:n_results_final]                                                   # :194


def find_best_usda_match(product_embedding, usda_embeddings):
    """analysis/generate_best_usda_matches.py:116-143."""
    best_code, best = None, -1.0                                                   # :130-131
    for code, emb in usda_embeddings.items():                                      # :133
        sim = np.dot(product_embedding, emb) / (np.linalg.norm(product_embedding) * np.linalg.norm(emb))  # :135-137
        if sim > best:                                                             # :139
            best, best_code = sim, code
    return best_code, best
