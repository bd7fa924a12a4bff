"""The reference's analysis harnesses as batched GPU workloads.

Each harness keeps the reference's bookkeeping and output columns but replaces N
sequential batch-1 `collection.query` calls by ONE batched query (which the
engine serves on MFMA wherever its cost model finds that faster than GEMV passes —
from 3 queries on at 10M bf16 rows (two stay on the scan, which is faster there); small stores like the reference's stay on the scan):

  evaluate_usda_classification    analysis/run_bidirectional_test.py:164-268
  classify_products_by_usda_code  analysis/run_bidirectional_test.py:115-161
  create_ground_truth_dict        analysis/run_bidirectional_test.py:72-112
  test_bidirectional_similarity   analysis/bidirectional_similarity.py:202-343
  generate_best_usda_matches      analysis/generate_best_usda_matches.py:146-327
  create_product_to_usda_mapping  analysis/generate_best_usda_matches.py:87-113
  evaluate_vector_db_accuracy     analysis/evaluate_accuracy.py:56-189 (core loop)

Embedding is batched too: the embedder's callable takes a list of texts.
"""
from __future__ import annotations

import re
from typing import Any, Dict, List, Optional, Sequence, Set, Tuple

import numpy as np
import pandas as pd

from . import config
from .data_processing import read_excel
from .matching import (best_usda_matches_batched, bidirectional_filter_row, normalize_description)


# ---------------------------------------------------------------- inputs
def load_ground_truth_mapping(mapping_file=None, sheet_name=None) -> pd.DataFrame:
    """Mapping workbook with every id column stringified (so NaN becomes 'nan',
    as in the reference, run_bidirectional_test.py:49-51); empty frame on failure."""
    try:
        df = read_excel(config.GROUND_TRUTH_FILE if mapping_file is None else mapping_file,
                        sheet_name=config.GROUND_TRUTH_SHEET_NAME if sheet_name is None else sheet_name)
    except Exception as exc:
        print(f"Error loading ground truth mapping: {exc}")
        return pd.DataFrame()
    for col in config.GROUND_TRUTH_ID_COLS:
        if col in df.columns:
            df[col] = df[col].astype(str)
    print(f"Loaded {len(df)} rows from ground truth mapping file.")
    return df


def load_transaction_frame(file_path=None, sheet_name=None) -> pd.DataFrame:
    try:
        df = read_excel(config.TRANSACTION_REPORT_FILE if file_path is None else file_path,
                        sheet_name=config.TRANSACTION_SHEET_NAME if sheet_name is None else sheet_name)
    except Exception as exc:
        print(f"Error loading transaction data: {exc}")
        return pd.DataFrame()
    print(f"Loaded {len(df)} rows from transaction data file.")
    return df


def get_unique_usda_codes(mapping_df: pd.DataFrame) -> List[str]:
    if mapping_df.empty or config.GROUND_TRUTH_USDA_COL not in mapping_df.columns:
        return []
    return mapping_df[config.GROUND_TRUTH_USDA_COL].dropna().unique().tolist()


def _mapped_codes_by_usda(mapping_df: pd.DataFrame, transaction_df: pd.DataFrame):
    """(usda code, stripped mapping id) pairs, in the reference's visiting order, for
    ids that also occur among the transaction product codes."""
    present = set(transaction_df[config.TRANSACTION_PRODUCT_CODE_COL].astype(str))
    usda_col = config.GROUND_TRUTH_USDA_COL
    for usda in get_unique_usda_codes(mapping_df):
        for _, row in mapping_df[mapping_df[usda_col] == usda].iterrows():
            for col in config.GROUND_TRUTH_ID_COLS:
                if col in row and pd.notna(row[col]):
                    pid = str(row[col]).strip()
                    if pid in present:
                        yield usda, pid


def create_ground_truth_dict(mapping_df: pd.DataFrame, transaction_df: pd.DataFrame) -> Dict[str, Set[str]]:
    """{USDA code -> product codes mapped to it that occur in the transactions}; codes with none are absent."""
    out: Dict[str, Set[str]] = {}
    for usda, pid in _mapped_codes_by_usda(mapping_df, transaction_df):
        out.setdefault(usda, set()).add(pid)
    return out


def create_product_to_usda_mapping(transaction_df: pd.DataFrame, mapping_df: pd.DataFrame) -> Dict[str, str]:
    """{product code -> known USDA code}; a product mapped twice keeps the later USDA code."""
    return {pid: usda for usda, pid in _mapped_codes_by_usda(mapping_df, transaction_df)}


def _prf(tp: int, fp: int, fn: int) -> Tuple[float, float, float]:
    p = tp / (tp + fp) if (tp + fp) > 0 else 0
    r = tp / (tp + fn) if (tp + fn) > 0 else 0
    f = 2 * p * r / (p + r) if (p + r) > 0 else 0
    return p, r, f


def _embed_many(vector_db, texts: Sequence[str]) -> np.ndarray:
    if not len(texts):
        return np.zeros((0, 0), np.float32)
    return np.asarray([np.asarray(e, dtype=np.float32) for e in vector_db.embedder.model(list(texts))])


# ---------------------------------------------------------------- USDA-code classification
def _matches_from_row(usda_code, ids, dists, metas, similarity_threshold) -> List[Dict[str, Any]]:
    out = []
    for pid, d, meta in zip(ids, dists, metas):
        sim = 1 - d
        if sim >= similarity_threshold:
            out.append({"product_id": pid, "product_code": meta.get("product_code", "N/A"),
                        "product_description": meta.get("product_description", "N/A"),
                        "similarity": sim, "usda_code": usda_code})
    return out


def classify_products_by_usda_code(vector_db, usda_code: str, usda_description: str, similarity_threshold: float,
                                   max_results: int = 100) -> List[Dict[str, Any]]:
    res = vector_db.collection.query(query_embeddings=_embed_many(vector_db, [usda_description]),
                                     n_results=max_results, include=["metadatas", "distances"])
    if not res or not res["ids"][0]:
        return []
    return _matches_from_row(usda_code, res["ids"][0], res["distances"][0], res["metadatas"][0], similarity_threshold)


def evaluate_usda_classification(vector_db, mapping_df: pd.DataFrame, transaction_df: pd.DataFrame,
                                 similarity_threshold: float = 0.7, max_results: int = 100):
    """-> (precision, recall, f1, detailed_df, usda_metrics_df); all USDA codes go
    through the index as one query batch."""
    ground_truth = create_ground_truth_dict(mapping_df, transaction_df)
    if not ground_truth:
        print("Error: No valid ground truth mappings found")
        return 0.0, 0.0, 0.0, pd.DataFrame(), pd.DataFrame()
    codes = list(ground_truth)
    # the text embedded for a code is the code itself (run_bidirectional_test.py:211)
    res = vector_db.collection.query(query_embeddings=_embed_many(vector_db, codes), n_results=max_results,
                                     include=["metadatas", "distances"])
    all_matches, metrics = [], []
    tot_tp = tot_fp = tot_fn = 0
    for j, usda in enumerate(codes):
        truth = ground_truth[usda]
        matched = _matches_from_row(usda, res["ids"][j], res["distances"][j], res["metadatas"][j], similarity_threshold)
        predicted = {str(m["product_code"]) for m in matched}
        tp, fp, fn = len(truth & predicted), len(predicted - truth), len(truth - predicted)
        tot_tp, tot_fp, tot_fn = tot_tp + tp, tot_fp + fp, tot_fn + fn
        p, r, f = _prf(tp, fp, fn)
        metrics.append({"usda_code": usda, "description": usda, "n_true_products": len(truth),
                        "n_predicted": len(predicted), "true_positives": tp, "false_positives": fp,
                        "false_negatives": fn, "precision": p, "recall": r, "f1": f})
        for m in matched:
            m["is_correct"] = str(m["product_code"]) in truth
            all_matches.append(m)
    p, r, f = _prf(tot_tp, tot_fp, tot_fn)
    return p, r, f, (pd.DataFrame(all_matches) if all_matches else pd.DataFrame()), pd.DataFrame(metrics)


# ---------------------------------------------------------------- bidirectional evaluation
def test_bidirectional_similarity(vector_db, queries_df: pd.DataFrame,
                                  forward_threshold: float = config.SIMILARITY_THRESHOLD_FORWARD,
                                  backward_threshold: float = config.SIMILARITY_THRESHOLD_BACKWARD,
                                  n_final_results: int = config.N_RESULTS_FINAL,
                                  n_results_initial: int = config.N_RESULTS_INITIAL_SEARCH):
    """-> (precision, recall, f1, results_df).  Queries whose code has no USDA
    ground truth are skipped; every returned candidate with the query's USDA code
    counts as a true positive; recall's denominator is the number of evaluated
    queries (reference bidirectional_similarity.py:329-336)."""
    if "product_description" not in queries_df.columns or "product_code" not in queries_df.columns:
        print("Error: queries_df must contain 'product_description' and 'product_code' columns.")
        return 0.0, 0.0, 0.0, pd.DataFrame()
    if len(queries_df) == 0:
        print("No queries provided for evaluation.")
        return 0.0, 0.0, 0.0, pd.DataFrame()
    todo = []
    for _, row in queries_df.iterrows():
        usda = vector_db._get_usda_code(row["product_code"])
        if usda != "NOT_FOUND":
            todo.append((row["product_description"], row["product_code"], usda))
    results: List[Dict[str, Any]] = []
    tp = predicted = 0
    if todo:
        descs = [t[0] for t in todo]
        fwd_q = _embed_many(vector_db, descs)
        back_q = _embed_many(vector_db, [normalize_description(d) for d in descs])
        res = vector_db.collection.query(query_embeddings=fwd_q, n_results=n_results_initial,
                                         include=["metadatas", "distances", "embeddings", "documents"])
        docs_all = res.get("documents") or [[] for _ in todo]
        for j, (desc, code, usda) in enumerate(todo):
            matches = bidirectional_filter_row(desc, back_q[j], res["distances"][j], res["metadatas"][j],
                                               res["embeddings"][j], docs_all[j], n_final_results,
                                               forward_threshold, backward_threshold)
            predicted += len(matches)
            if not matches:
                results.append({"query_description": desc, "query_code": code, "true_usda_code": usda,
                                "candidate_description": None, "candidate_code": None, "candidate_usda_code": None,
                                "forward_similarity": None, "backward_similarity": None, "is_correct_match": False})
            for m in matches:
                meta = m["metadata"]
                cand_usda = meta.get("usda_code", "MISSING_IN_METADATA")
                ok = cand_usda not in ("NOT_FOUND", "MISSING_IN_METADATA") and cand_usda == usda
                tp += bool(ok)
                results.append({"query_description": desc, "query_code": code, "true_usda_code": usda,
                                "candidate_description": meta.get("product_description", "N/A"),
                                "candidate_code": meta.get("product_code", "N/A"), "candidate_usda_code": cand_usda,
                                "forward_similarity": m["forward_similarity"],
                                "backward_similarity": m["backward_similarity"], "is_correct_match": ok})
    precision = tp / predicted if predicted > 0 else 0.0
    recall = tp / len(todo) if todo else 0.0
    f1 = 2 * (precision * recall) / (precision + recall) if (precision + recall) > 0 else 0.0
    return precision, recall, f1, (pd.DataFrame(results) if results else pd.DataFrame())


test_bidirectional_similarity.__test__ = False      # a harness, not a pytest case


# ---------------------------------------------------------------- best USDA match per product
BEST_MATCH_COLUMNS = ["product_id", "product_code", "product_description", "best_matching_usda_code",
                      "similarity_score", "known_usda_code", "is_correct_match"]


def generate_best_usda_matches(vector_db, mapping_df: pd.DataFrame, transaction_df: pd.DataFrame,
                               usda_embeddings: Dict[str, np.ndarray], product_to_usda: Dict[str, str],
                               on_id_hit: str = "skip") -> pd.DataFrame:
    """One row per distinct (code, description) of the transactions: the nearest
    stored vector to the embedded description, against every USDA vector, best one
    kept.  The nearest-vector queries and the arg-max are each ONE batched call.

    The reference first looks the product up by id `item_<code>`, but the store's
    ids are positional (`item_<row>`, src/vectordb.py:57), so a hit is a coincidence
    (a numeric code below the row count) and, with the chromadb it was run on, such
    a hit is then dropped at generate_best_usda_matches.py:229-231 (get() returns an
    ndarray, not a list).  Its recorded output (analysis_results/best_usda_matches.csv:
    15 235 rows for 15 868 distinct pairs, every product_id a one-element list) shows
    exactly that.  on_id_hit="skip" (default) reproduces the recorded behaviour;
    "use" takes the stored vector of the id hit instead."""
    if on_id_hit not in ("skip", "use"):
        raise ValueError("on_id_hit must be 'skip' or 'use'")
    code_col, desc_col = config.TRANSACTION_PRODUCT_CODE_COL, config.TRANSACTION_DESC_COL
    uniq = transaction_df.drop_duplicates(subset=[code_col, desc_col])
    codes = [str(c) for c in uniq[code_col]]
    descs = list(uniq[desc_col])
    want = [f"item_{c}" for c in codes]
    got = vector_db.collection.get(ids=sorted(set(want)), include=["embeddings"])
    by_id = {i: np.asarray(e, dtype=np.float32) for i, e in zip(got["ids"], got["embeddings"] if got["ids"] else [])}
    emb: List[Optional[np.ndarray]] = [by_id.get(w) if on_id_hit == "use" else None for w in want]
    pid: List[Any] = [w if w in by_id else None for w in want]
    # a description that is not text (NaN) cannot be embedded: the reference's fallback query raises on it, prints,
    # and the row is skipped (generate_best_usda_matches.py:201-203) — three such rows in its own data
    miss = [j for j, w in enumerate(want) if w not in by_id and isinstance(descs[j], str)]
    if miss and vector_db.collection.count():
        q = _embed_many(vector_db, [descs[j] for j in miss])
        res = vector_db.collection.query(query_embeddings=q, n_results=1, include=["embeddings"])
        for slot, j in enumerate(miss):
            if res["ids"][slot]:
                emb[j] = np.asarray(res["embeddings"][slot][0], dtype=np.float32)
                # the reference reports the id LIST of the fallback query here
                # (generate_best_usda_matches.py:218 reads ids[0] of a nested result)
                pid[j] = list(res["ids"][slot])
    keep = [j for j, e in enumerate(emb) if e is not None]
    rows = []
    if keep and usda_embeddings:
        best_codes, sims = best_usda_matches_batched(np.stack([emb[j] for j in keep]), usda_embeddings,
                                                     device=getattr(vector_db.collection, "_device", 0),
                                                     lib=getattr(vector_db.collection, "_lib", None))
        for j, best, sim in zip(keep, best_codes, sims):
            known = product_to_usda.get(codes[j])
            rows.append({"product_id": pid[j], "product_code": codes[j], "product_description": descs[j],
                         "best_matching_usda_code": best, "similarity_score": float(sim), "known_usda_code": known,
                         "is_correct_match": (known == best) if known else None})
    return pd.DataFrame(rows, columns=BEST_MATCH_COLUMNS)


def summarize_best_matches(results_df: pd.DataFrame) -> Dict[str, Any]:
    """Counts printed at the end of generate_best_usda_matches.py:291-325."""
    known = results_df.dropna(subset=["known_usda_code"])
    correct = int((known["is_correct_match"] == True).sum())   # noqa: E712  (object column holding bools/None)
    return {"total_products": len(results_df), "known": len(known), "unknown": len(results_df) - len(known),
            "correct": correct, "incorrect": int((known["is_correct_match"] == False).sum()),  # noqa: E712
            "accuracy": correct / len(known) if len(known) else 0.0}


# ---------------------------------------------------------------- reference-mapping accuracy
def _word_contains(a: str, b: str) -> bool:
    return bool(re.search(rf"\b{re.escape(a)}\b", b) or re.search(rf"\b{re.escape(b)}\b", a))


def evaluate_vector_db_accuracy(vector_db, reference_map: Dict[str, List[str]], unique_products: pd.DataFrame,
                                n_results: int = 10) -> Dict[str, float]:
    """For each standardized product, each of its (lower-cased) descriptions that
    occurs in the transactions queries the store; the other descriptions of the
    same product are the expected matches (word-boundary containment either way).
    All queries of all products form one batch."""
    from .vectordb import bidirectional_rerank
    prods = [str(p).lower() for p in unique_products["product_description"]]
    filtered = {}
    for std, descs in reference_map.items():
        found = [d for d in descs if any(_word_contains(d.lower().strip(), p) for p in prods)]
        if found:
            filtered[std] = found
    jobs = []
    for std, descs in filtered.items():
        for q in descs:
            correct = {d.lower() for d in descs if d.lower() != q.lower()}
            if correct:
                jobs.append((q, correct))
    tp = fp = fn = 0
    if jobs:
        n_results = min(n_results, len(unique_products))
        qv = _embed_many(vector_db, [j[0] for j in jobs])
        res = vector_db.collection.query(query_embeddings=qv, n_results=config.N_RESULTS_INITIAL_SEARCH,
                                         include=["metadatas", "distances", "embeddings"])
        for j, (q, correct) in enumerate(jobs):
            if not res["ids"][j]:
                matched = set()
            else:
                df = bidirectional_rerank(res["ids"][j], res["distances"][j], res["metadatas"][j],
                                          res["embeddings"][j], qv[j], n_results, None)
                matched = {str(r).lower() for r in df["product_description"]}
            hits = sum(1 for m in matched if any(_word_contains(c, m) for c in correct))
            tp, fp, fn = tp + hits, fp + len(matched) - hits, fn + len(correct) - hits
    p, r, f = _prf(tp, fp, fn)
    return {"total_queries": len(jobs), "correct_matches": tp, "false_positives": fp, "false_negatives": fn,
            "precision": p, "recall": r, "f1_score": f}
