#!/usr/bin/env python3
"""Correctness gate at BASELINE's full size (SURVEY.md 8d "correctness gates"): 64 sampled
queries against the 10M x 768 synthetic corpus, answered by the engine three ways —

  * bf16 store + fp32 rerank, one batch of 64 (MFMA pass kernels),
  * the same store, one query at a time (HBM-bound GEMV kernel),
  * fp32 rerank OFF (bf16 x bf16 scores) for the recall figure of the candidates alone,

and checked against a streamed CPU brute force over the same corpus regenerated chunk by
chunk on the host (the oracle's generator is bit-identical to the device's; fp64 scores).
The oracle is used here as the checker only (which is why this script lives under tests/;
pytest does not collect it).  Prints one JSON document.

  python3 tests/verify_10m.py [--rows 10000000] [--queries 64]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vectordb_amd import FlatIndex, SynthSpec  # noqa: E402
from tests.util import check_topk, recall_at_k  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=250_000)
    ap.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args()
    from oracle import oracle   # checker
    n, dim, b, k, kp = args.rows, 768, args.queries, 10, 40
    spec = SynthSpec(seed=args.seed, kind=1, n_clusters=max(1, int(np.sqrt(n))), noise=0.3)
    qspec = SynthSpec(seed=args.seed + 1, kind=2, noise=0.5, base_kind=1, base_seed=args.seed, base_n=n,
                      base_n_clusters=spec.n_clusters, base_noise=0.3)
    queries = oracle.synth_rows(0, b, dim, qspec)
    qn = oracle.normalize_rows(queries)

    t0 = time.perf_counter()
    ix = FlatIndex(dim, "bf16", True)
    ix.add_synth(0, n, spec)
    idx_b, dist_b = ix.search(queries, k, kp)                       # batch of 64: MFMA path
    overflow = ix.get_stat("mfma_last_overflow")
    idx_1 = np.empty_like(idx_b)
    dist_1 = np.empty_like(dist_b)
    for q in range(b):                                              # one at a time: GEMV path
        i1, d1 = ix.search(queries[q:q + 1], k, kp)
        idx_1[q], dist_1[q] = i1[0], d1[0]
    idx_nr, dist_nr = ix.search(queries, k, 0)                      # candidates only, no rerank
    ix.close()
    t_gpu = time.perf_counter() - t0

    # streamed exact top-k on the host: fp32 rows as stored, fp64 accumulation, ties by row index
    t0 = time.perf_counter()
    best_s = np.full((b, k), -np.inf)
    best_i = np.full((b, k), -1, np.int64)
    qn64 = qn.astype(np.float64)
    for lo in range(0, n, args.chunk):
        m = min(args.chunk, n - lo)
        rows = oracle.normalize_rows(oracle.synth_rows(lo, m, dim, spec))
        s = rows.astype(np.float64) @ qn64.T                        # [m, b]
        part = np.argpartition(-s, min(k, m - 1), axis=0)[:k]       # [k, b]
        cs = np.concatenate([best_s, np.take_along_axis(s, part, axis=0).T], axis=1)
        ci = np.concatenate([best_i, (part + lo).T], axis=1)
        order = np.lexsort((ci, -cs), axis=1)[:, :k]
        best_s = np.take_along_axis(cs, order, axis=1)
        best_i = np.take_along_axis(ci, order, axis=1)
    t_cpu = time.perf_counter() - t0
    ref_dist = 1.0 - best_s

    class _Rows:   # check_topk only needs rows_norm[ids]: regenerate the few returned rows on demand
        def __getitem__(self, ids):
            ids = np.asarray(ids, np.int64)
            return np.stack([oracle.normalize_rows(oracle.synth_rows(int(i), 1, dim, spec))[0] for i in ids]) if ids.size else np.zeros((0, dim), np.float32)

    rows_view = _Rows()
    swaps_b = check_topk(idx_b, dist_b, best_i, ref_dist, best_s, rows_view, qn, what="10M batch-64 (MFMA)")
    swaps_1 = check_topk(idx_1, dist_1, best_i, ref_dist, best_s, rows_view, qn, what="10M batch-1 (GEMV)")
    out = {
        "rows": n, "dim": dim, "queries": b, "k": k, "kprime": kp, "corpus": "clustered synthetic, seed %d" % args.seed,
        "mfma_batch": {"recall_at_10": recall_at_k(idx_b, best_i), "in_tolerance_tie_swaps": swaps_b, "overflow_fallback": int(overflow),
                       "max_abs_dist_err": float(np.max(np.abs(dist_b - ref_dist)))},
        "gemv_batch1": {"recall_at_10": recall_at_k(idx_1, best_i), "in_tolerance_tie_swaps": swaps_1,
                        "max_abs_dist_err": float(np.max(np.abs(dist_1 - ref_dist)))},
        "paths_agree_bitwise": bool(np.array_equal(idx_b, idx_1) and np.array_equal(dist_b, dist_1)),
        "bf16_candidates_without_rerank": {"recall_at_10": recall_at_k(idx_nr, best_i), "max_abs_dist_err": float(np.max(np.abs(dist_nr - ref_dist)))},
        "tolerance": "scores within 1e-5 relative of the exact fp64 cosine (tests/util.py); ids identical except ties inside it",
        "seconds": {"gpu_build_and_search": t_gpu, "cpu_streamed_brute_force": t_cpu},
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
