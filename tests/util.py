"""Tolerance-aware top-k comparison (BASELINE.json: scores within 1e-5 relative,
identical id lists except for ties inside that tolerance)."""
import numpy as np

REL_TOL = 1e-5      # north_star tolerance on scores
ABS_FLOOR = 2e-6    # fp32 rounding of 1 - s near s ~ 1


def exact_scores(rows_norm, qn, ids):
    """fp64 cosine of normalised query qn against the given rows."""
    return rows_norm[ids].astype(np.float64) @ qn.astype(np.float64)


def check_topk(idx, dist, ref_idx, ref_dist, ref_score, rows_norm, queries_norm, row_offset=0, what=""):
    """Returns the number of positions whose id differs from the oracle (all of
    which must be in-tolerance ties)."""
    b, k = ref_idx.shape
    assert idx.shape == (b, k) and dist.shape == (b, k), f"{what}: shape {idx.shape} vs {(b, k)}"
    swaps = 0
    for q in range(b):
        valid = ref_idx[q] >= 0
        nv = int(valid.sum())
        assert np.all(idx[q, nv:] == -1) and np.all(np.isinf(dist[q, nv:])), f"{what}: q{q} padding"
        got = idx[q, :nv] - row_offset
        assert len(set(got.tolist())) == nv and got.min(initial=0) >= 0, f"{what}: q{q} duplicate/invalid ids {got}"
        tol = REL_TOL * np.maximum(np.abs(ref_score[q, :nv]), 1e-3) + ABS_FLOOR
        # distances agree position by position
        assert np.all(np.abs(dist[q, :nv].astype(np.float64) - ref_dist[q, :nv]) <= tol), \
            f"{what}: q{q} distances differ: {dist[q, :nv]} vs {ref_dist[q, :nv]}"
        # returned distance is the true distance of the returned row
        true_s = exact_scores(rows_norm, queries_norm[q], got)
        assert np.all(np.abs((1.0 - true_s) - dist[q, :nv]) <= tol), f"{what}: q{q} distance != 1-cos of returned row"
        # ascending
        assert np.all(np.diff(dist[q, :nv].astype(np.float64)) >= -tol[1:] if nv > 1 else True), f"{what}: q{q} not ascending"
        diff = got != ref_idx[q, :nv]
        if diff.any():
            # every mismatching row must tie (within tolerance) with the oracle's row at that rank
            assert np.all(np.abs(true_s[diff] - ref_score[q, :nv][diff]) <= tol[diff]), \
                f"{what}: q{q} ids differ outside tolerance: {got} vs {ref_idx[q, :nv]}"
            swaps += int(diff.sum())
    return swaps


def recall_at_k(idx, ref_idx):
    hits = 0
    total = 0
    for q in range(ref_idx.shape[0]):
        ref = set(int(i) for i in ref_idx[q] if i >= 0)
        hits += len(ref & set(int(i) for i in idx[q]))
        total += len(ref)
    return hits / max(total, 1)
