"""Host glue of the reference's analysis scripts that sits directly on the query
path, plus their batched GPU forms.

  find_similar_products_bidirectional   analysis/bidirectional_similarity.py:67-200
  normalize_description                 analysis/bidirectional_similarity.py (lower/strip)
  find_best_usda_match                  analysis/generate_best_usda_matches.py:116-143
  best_usda_matches_batched             the 15 235 x 37 arg-max of generate_best_usda_matches.py:266
                                        as ONE k=1 query batch against a tiny collection
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Tuple

import numpy as np

from .index import FlatIndex


def normalize_description(description: str) -> str:
    """lower + strip; non-strings are stringified (analysis/bidirectional_similarity.py:18-29)."""
    return description.lower().strip() if isinstance(description, str) else str(description).strip()


def _cosine(a: np.ndarray, b: np.ndarray) -> float:
    """sklearn.metrics.pairwise.cosine_similarity on two single rows: each row is
    scaled by 1/||row|| (rows with zero norm are left as zeros), then dotted."""
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    na, nb = np.sqrt(np.dot(a, a)), np.sqrt(np.dot(b, b))
    an = a / (na if na != 0.0 else 1.0)
    bn = b / (nb if nb != 0.0 else 1.0)
    return float(np.dot(an, bn))


def find_similar_products_bidirectional(query_description: str, vector_db, n_results_initial: int = 10,
                                        n_results_final: int = 5, similarity_threshold_forward: float = 0.5,
                                        similarity_threshold_backward: float = 0.5, verbose: bool = False,
                                        embedder: Any = None) -> List[Dict[str, Any]]:
    """Forward top-k from the store, then per candidate: drop forward < tau_f, drop
    textual self matches, backward = cos(candidate, embed(lower/strip(query))),
    keep forward >= tau_f and backward >= tau_b, sort by the mean (descending,
    stable), first n_results_final."""
    emb = embedder if embedder is not None else vector_db.embedder
    if emb is None:
        return []
    q = emb.embed_query(query_description)
    try:
        res = vector_db.collection.query(query_embeddings=[np.asarray(q).tolist()], n_results=n_results_initial,
                                         include=["metadatas", "distances", "embeddings", "documents"])
    except Exception as exc:
        if verbose:
            print(f"Error querying collection: {exc}")
        return []
    if not res or not res.get("ids"):
        return []
    docs = (res.get("documents") or [[]])[0]
    q_back = np.asarray(emb.embed_query(normalize_description(query_description)))
    return bidirectional_filter_row(query_description, q_back, res["distances"][0], res["metadatas"][0],
                                    res["embeddings"][0], docs, n_results_final, similarity_threshold_forward,
                                    similarity_threshold_backward)


def bidirectional_filter_row(query_description: str, q_back: np.ndarray, dists, metas, embs, docs,
                             n_results_final: int, similarity_threshold_forward: float,
                             similarity_threshold_backward: float) -> List[Dict[str, Any]]:
    """The per-candidate stage of analysis/bidirectional_similarity.py:140-194 on ONE
    query's result row; shared by the single-query call above and the batched
    harness (analysis.test_bidirectional_similarity)."""
    q_norm_text = normalize_description(query_description)
    matches = []
    for i in range(len(dists)):
        fwd = 1 - dists[i]
        if fwd < similarity_threshold_forward:
            continue
        meta = metas[i]
        cand_desc = meta.get("product_description", docs[i] if i < len(docs) else "Unknown")
        if normalize_description(cand_desc) == q_norm_text:
            continue
        bwd = _cosine(embs[i], q_back)
        if fwd >= similarity_threshold_forward and bwd >= similarity_threshold_backward:
            matches.append({"query_description": query_description, "candidate_description": cand_desc,
                            "forward_similarity": fwd, "backward_similarity": bwd,
                            "avg_similarity": (fwd + bwd) / 2, "metadata": meta})
    matches.sort(key=lambda m: m["avg_similarity"], reverse=True)
    return matches[:n_results_final]


def find_best_usda_match(product_embedding: np.ndarray, usda_embeddings: Dict[str, np.ndarray]) -> Tuple[Optional[str], float]:
    """arg-max over the USDA vectors of cos(product, usda); strict '>' so the first
    maximum in dict order wins; starts from -1.0 / None."""
    best_code, best = None, -1.0
    p = np.asarray(product_embedding)
    pn = np.linalg.norm(p)
    for code, u in usda_embeddings.items():
        sim = np.dot(p, u) / (pn * np.linalg.norm(u))
        if sim > best:
            best, best_code = sim, code
    return best_code, best


def best_usda_matches_batched(product_embeddings, usda_embeddings: Dict[str, np.ndarray], device: int = 0,
                              lib=None) -> Tuple[List[Optional[str]], np.ndarray]:
    """All products at once on the GPU: the USDA vectors form a tiny exact store and
    every product is a k=1 query.  Ties resolve to the lowest row = first in dict
    order, as the strict '>' loop does.  A product the engine returns nothing for
    (a non-finite embedding: idx -1, distance +inf) maps to (None, -1.0), which is
    what the reference's loop reports for it (every comparison with NaN is false)."""
    codes = list(usda_embeddings)
    mat = np.stack([np.asarray(usda_embeddings[c], dtype=np.float32).reshape(-1) for c in codes])
    idx = FlatIndex(mat.shape[1], "f32", True, device, lib=lib)
    try:
        idx.add(mat)
        rows, dist = idx.search(np.asarray(product_embeddings, dtype=np.float32), 1)
    finally:
        idx.close()
    hit = rows[:, 0] >= 0
    sims = np.where(hit, 1.0 - dist[:, 0].astype(np.float64), -1.0)
    return [codes[r] if ok else None for r, ok in zip(rows[:, 0], hit)], sims
