"""GPU suite, round 3: the one-launch small-batch search (scan_fused.h: query normalisation in the scan's prologue,
fold + fp32 rerank + ranking by the last block to arrive).  Everything goes through the C ABI and is checked bit for
bit against the four-launch form of the same search (option fuse_small_batch = 0) and against the CPU oracle."""
import numpy as np
import pytest

from vectordb_amd import FlatIndex, SynthSpec
from tests.util import check_topk

pytestmark = pytest.mark.gpu


def _qn(oracle_mod, queries):
    return oracle_mod.normalize_rows(np.ascontiguousarray(queries, np.float32))


def _fused_vs_unfused(ix, queries, k, kp, what, allow=None):
    """The same search through the fused kernel and through normalise + scan + fold + finish: identical bits."""
    out = []
    for fuse in (1, 0, 1):      # the third call re-uses the arrival counter the first one reset
        ix.set_option("fuse_small_batch", fuse)
        out.append(ix.search(queries, k, kp) if allow is None else ix.search_masked(queries, k, allow, kp))
    ix.set_option("fuse_small_batch", 1)
    for idx, dist in out[1:]:
        assert np.array_equal(idx, out[0][0]), f"{what}: ids differ between the fused and the four-launch search"
        assert np.array_equal(dist.view(np.uint32), out[0][1].view(np.uint32)), f"{what}: distances differ in their bits"
    return out[0]


@pytest.mark.parametrize("dtype,dim", [("bf16", 768), ("f32", 768), ("bf16", 384), ("f32", 384)])
def test_fused_search_equals_four_launches_and_oracle(hip_lib, oracle_mod, dtype, dim):
    n = 40_011                                   # ragged against the 8/16-row scan tiles
    spec = SynthSpec(seed=31, kind=1, n_clusters=199, noise=0.3)
    qspec = SynthSpec(seed=32, kind=2, noise=0.5, base_kind=1, base_seed=31, base_n=n, base_n_clusters=199, base_noise=0.3)
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    queries = oracle_mod.synth_rows(0, 4, dim, qspec)
    queries[1] *= 37.5                           # raw queries are not unit vectors: the prologue normalises them
    ix = FlatIndex(dim, dtype, True)
    ix.add_synth(0, n, spec)
    qn = _qn(oracle_mod, queries)
    for b in (1, 2, 3, 4):
        for k, kp in ((10, 40), (1, 1), (3, 64), (64, 64), (30, 0)):
            if dtype == "f32" or kp == 0:
                kp = 0
            idx, dist = _fused_vs_unfused(ix, queries[:b], k, kp, f"{dtype} {dim} b={b} k={k} kp={kp}")
            assert ix.get_stat("last_path") == 0 and ix.get_stat("last_corpus_passes") == (2 if b == 3 else 1)
            if dtype == "bf16" and kp < 4 * k:
                continue                          # bf16 candidates with little or no head-room: checked against the four-launch form only
            r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries[:b], k)
            check_topk(idx, dist, r_idx, r_dist, r_sc, rows, qn[:b], what=f"fused {dtype} {dim} b={b} k={k}")
    # five queries and k' above 64 stay on the unfused kernels and still agree with themselves
    i5, d5 = ix.search(queries[[0, 1, 2, 3, 0]], 10, 40 if dtype == "bf16" else 0)
    i1, d1 = ix.search(queries[:1], 10, 40 if dtype == "bf16" else 0)
    assert np.array_equal(i5[0], i1[0]) and np.array_equal(i5[4], i1[0]) and np.array_equal(d5[4], d1[0])
    ix.close()


def test_fused_search_tiny_stores_masks_and_offsets(hip_lib, oracle_mod):
    """Stores smaller than k, than one scan tile, than the bound step's 4 x ceil(k'/4) lists; where= bitmaps; a shard
    offset; an empty store."""
    dim = 768
    spec = SynthSpec(seed=5, kind=0)
    queries = oracle_mod.synth_rows(0, 3, dim, SynthSpec(seed=6, kind=0))
    for n in (1, 7, 63, 64, 65, 700, 2500, 2560, 5000):
        rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
        for dtype, kp in (("f32", 0), ("bf16", 40)):
            ix = FlatIndex(dim, dtype, True)
            ix.add_synth(0, n, spec)
            ix.set_row_offset(1_000_000)
            idx, dist = _fused_vs_unfused(ix, queries, 10, kp, f"n={n} {dtype}")
            r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries, 10)
            check_topk(idx, dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries), row_offset=1_000_000, what=f"tiny n={n} {dtype}")
            if n >= 63:
                allow = (np.arange(n) % 3) == 1
                idx, dist = _fused_vs_unfused(ix, queries[:2], 10, kp, f"masked n={n} {dtype}", allow=allow)
                got = idx[idx >= 0] - 1_000_000
                assert np.all(allow[got]), "a masked search returned an ineligible row"
                sub = np.flatnonzero(allow)
                r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows[sub], queries[:2], 10)
                r_idx = np.where(r_idx >= 0, sub[np.maximum(r_idx, 0)], -1)
                check_topk(idx, dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries[:2]), row_offset=1_000_000, what=f"masked n={n} {dtype}")
            ix.close()
    ix = FlatIndex(dim, "f32", True)
    idx, dist = ix.search(queries[:1], 5)
    assert np.all(idx == -1) and np.all(np.isinf(dist))
    ix.close()


def test_fused_fold_survives_adversarial_orders(hip_lib, oracle_mod):
    """Orders that defeat the bound of the fused fold: the best rows all in ONE block's stripe, and thousands of
    identical rows (every candidate ties with the bound: more survivors than the LDS buffer holds, so the exact
    list-by-list fold runs).  Bit-identical to the four-launch search either way."""
    dim, n = 768, 120_000
    rng = np.random.default_rng(11)
    base = rng.standard_normal((n, dim)).astype(np.float32)
    q = rng.standard_normal((2, dim)).astype(np.float32)
    # (a) a run of near-copies of the query in consecutive rows (they fall into a handful of scan blocks)
    near = base.copy()
    near[5000:5300] = q[0] + 0.05 * rng.standard_normal((300, dim)).astype(np.float32)
    # (b) every row identical: all scores tie, order is by row index alone
    same = np.tile(base[:1], (n, 1))
    for name, rows_raw in (("clustered-run", near), ("all-identical", same)):
        rows = oracle_mod.normalize_rows(rows_raw)
        for dtype, kp in (("f32", 0), ("bf16", 40), ("bf16", 64)):
            ix = FlatIndex(dim, dtype, True)
            ix.add(rows_raw)
            for k in (10, 64) if kp == 0 else (10,):
                idx, dist = _fused_vs_unfused(ix, q, k, kp, f"{name} {dtype} k={k}")
                r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, q, k)
                if name == "all-identical":
                    assert np.array_equal(idx, r_idx), "ties must come back in row order"
                elif dtype == "f32":     # (300 near-copies sit inside bf16 resolution: k' = 40 candidates cannot rank them — the
                    check_topk(idx, dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, q), what=f"{name} {dtype} k={k}")   # documented bf16 limit; bf16 is checked against the four-launch form)
            ix.close()


def test_fused_search_on_device_pointers_and_a_sharded_handle(hip_lib, oracle_mod):
    """vdb_search_dev on one store and on a sharded handle (three shards on the one GPU): every shard's last block
    stores its k results straight into the gather buffer; identical to the single store."""
    import torch
    dim, n = 768, 90_001
    spec = SynthSpec(seed=77, kind=1, n_clusters=300, noise=0.3)
    qspec = SynthSpec(seed=78, kind=2, noise=0.5, base_kind=1, base_seed=77, base_n=n, base_n_clusters=300, base_noise=0.3)
    queries = oracle_mod.synth_rows(0, 4, dim, qspec)
    one = FlatIndex(dim, "bf16", True)
    one.add_synth(0, n, spec)
    sh = FlatIndex(dim, "bf16", True, devices=[0, 0, 0])
    sh.reserve(n)
    sh.add_synth(0, n, spec)
    st = torch.cuda.current_stream().cuda_stream
    q_t = torch.from_numpy(queries).cuda()
    for b in (1, 2, 4):
        res = []
        for ix in (one, sh):
            oi = torch.full((b, 10), -7, dtype=torch.int64, device="cuda")
            od = torch.full((b, 10), -7.0, dtype=torch.float32, device="cuda")
            for _ in range(3):          # back to back on one stream: the counter resets between launches
                ix.search_device(q_t.data_ptr(), b, 10, 40, oi.data_ptr(), od.data_ptr(), st)
            torch.cuda.synchronize()
            res.append((oi.cpu().numpy(), od.cpu().numpy()))
        assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])
        hi, hd = one.search(queries[:b], 10, 40)
        assert np.array_equal(hi, res[0][0]) and np.array_equal(hd, res[0][1])
    one.close()
    sh.close()
