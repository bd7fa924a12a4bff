"""GPU suite (-m gpu): the HIP engine, called through the C ABI, against the CPU
oracle on the same seeded inputs, the committed golden vectors, and — at sizes
the oracle cannot scan — size-independent properties.

Tolerances (BASELINE.json): scores within 1e-5 relative of the exact fp64
cosine; id lists identical except ties inside that tolerance; bf16 + fp32
rerank recall@10 >= 0.999.
"""
import os

import numpy as np
import pytest

from vectordb_amd import FlatIndex, SynthSpec
from tests.util import check_topk, recall_at_k

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _spec(g, prefix):
    return SynthSpec(**{f: g[f"{prefix}_{f}"].item() for f, _ in SynthSpec._fields_})


def _qn(oracle_mod, queries):
    return oracle_mod.normalize_rows(np.ascontiguousarray(queries, np.float32))


@pytest.fixture(scope="module")
def corpus20k(oracle_mod):
    n, dim = 20011, 768  # not a multiple of any tile size
    spec = SynthSpec(seed=1234, kind=1, n_clusters=141, noise=0.3)
    qspec = SynthSpec(seed=99, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=141, base_noise=0.3)
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    queries = oracle_mod.synth_rows(0, 9, dim, qspec)
    return n, dim, spec, rows, queries


def test_device_generator_bit_exact(hip_lib, oracle_mod):
    """synth rows generated in HBM == the oracle's, bit for bit (via add_synth + fetch
    of the normalised rows, and via the raw generator into a torch buffer)."""
    import torch
    for spec in (SynthSpec(seed=5, kind=0), SynthSpec(seed=5, kind=1, n_clusters=7, noise=0.3),
                 SynthSpec(seed=8, kind=2, noise=0.25, base_kind=1, base_seed=5, base_n=1000, base_n_clusters=7, base_noise=0.3)):
        ix = FlatIndex(768, "f32")
        buf = torch.empty((130, 768), dtype=torch.float32, device="cuda")
        ix.synth_rows_device(buf.data_ptr(), 1000, 130, spec)
        torch.cuda.synchronize()
        raw = oracle_mod.synth_rows(1000, 130, 768, spec)
        assert np.array_equal(buf.cpu().numpy(), raw)
        ix.add_synth(1000, 130, spec)
        got = ix.fetch(np.arange(130))
        ref = oracle_mod.normalize_rows(raw)
        assert np.max(np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))) <= 1  # <= 1 ulp
        ix.close()


@pytest.mark.parametrize("name", ["iso", "clustered"])
def test_golden_synth_f32(hip_lib, oracle_mod, name):
    g = np.load(os.path.join(GOLD, f"synth_{name}_n4096.npz"))
    n, dim, b = int(g["n"]), int(g["dim"]), int(g["b"])
    spec, qspec = _spec(g, "spec"), _spec(g, "qspec")
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    queries = oracle_mod.synth_rows(0, b, dim, qspec)
    ix = FlatIndex(dim, "f32")
    ix.add_synth(0, n, spec)
    assert ix.count() == n
    for k in (1, 10, 30, 100):
        idx, dist = ix.search(queries, k)
        check_topk(idx, dist, g[f"idx_k{k}"], g[f"dist_k{k}"], g[f"score_k{k}"], rows, _qn(oracle_mod, queries), what=f"{name} k={k}")
        # batch-1 calls (every reference call site is batch 1) give the same answer
        i1, d1 = ix.search(queries[:1], k)
        assert np.array_equal(i1, idx[:1]) and np.array_equal(d1, dist[:1])
    ix.close()


def test_golden_adversarial(hip_lib, oracle_mod):
    """duplicates, exact ties, a zero row, a zero query, k > N; generic (non-768) dim path."""
    g = np.load(os.path.join(GOLD, "adversarial_d64.npz"))
    rows_n = oracle_mod.normalize_rows(g["rows"])
    for dtype, kpf in (("f32", 0), ("bf16", 1)):
        ix = FlatIndex(64, dtype)
        ix.add(g["rows"])
        for k in (1, 5, 31, 40):
            kp = 0 if not kpf else min(max(4 * k, 32), 1024)
            idx, dist = ix.search(g["queries"], k, kp)
            swaps = check_topk(idx, dist, g[f"idx_k{k}"], g[f"dist_k{k}"], g[f"score_k{k}"], rows_n, _qn(oracle_mod, g["queries"]),
                               what=f"adversarial {dtype} k={k}")
            if dtype == "f32" and k == 1:
                assert swaps == 0
        ix.close()


@pytest.mark.parametrize("dtype,kp", [("f32", 0), ("bf16", 0), ("bf16", 40)])
def test_parity_20k(hip_lib, oracle_mod, corpus20k, dtype, kp):
    n, dim, spec, rows, queries = corpus20k
    ix = FlatIndex(dim, dtype)
    ix.add_synth(0, n, spec)
    qn = _qn(oracle_mod, queries)
    for b in (1, 2, 3, 9):   # exercises the 1/2/4-queries-per-pass kernels
        for k in (1, 10, 30):
            if kp and kp < k:
                continue
            idx, dist = ix.search(queries[:b], k, kp)
            if dtype == "bf16" and kp == 0:
                # no rerank: scores are those of the bf16-rounded corpus (fp32 query)
                rows_b = oracle_mod.round_bf16(rows)
                r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows_b, queries[:b], k)
                check_topk(idx, dist, r_idx, r_dist, r_sc, rows_b, qn[:b], what=f"bf16 raw b={b} k={k}")
            else:
                r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries[:b], k)
                swaps = check_topk(idx, dist, r_idx, r_dist, r_sc, rows, qn[:b], what=f"{dtype} b={b} k={k}")
                if kp:  # bf16 + fp32 rerank: raw id recall loses nothing but in-tolerance tie swaps
                    assert recall_at_k(idx, r_idx) >= 1.0 - swaps / idx.size - 1e-12
    ix.close()


def test_large_k_and_host_add(hip_lib, oracle_mod):
    """k = 100..1000 (multi-entry wave lists), rows added from host memory in two calls."""
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((3000, 384)).astype(np.float32)
    rows_n = oracle_mod.normalize_rows(rows)
    queries = rows[[10, 2000]] + 0.05 * rng.standard_normal((2, 384)).astype(np.float32)
    ix = FlatIndex(384, "f32")
    ix.add(rows[:1234]); ix.add(rows[1234:])
    got = ix.fetch([0, 1233, 1234, 2999])
    ref = rows_n[[0, 1233, 1234, 2999]]
    assert np.max(np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))) <= 1
    for k in (100, 200, 500, 1000):
        idx, dist = ix.search(queries, k)
        r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows_n, queries, k)
        check_topk(idx, dist, r_idx, r_dist, r_sc, rows_n, _qn(oracle_mod, queries), what=f"k={k}")
    ix.close()


def test_empty_and_tiny_store(hip_lib):
    ix = FlatIndex(768, "f32")
    idx, dist = ix.search(np.ones((2, 768), np.float32), 5)
    assert np.all(idx == -1) and np.all(np.isinf(dist))
    ix.add(np.eye(768, dtype=np.float32)[:3])
    idx, dist = ix.search(np.eye(768, dtype=np.float32)[1:2], 5)
    assert idx[0].tolist() == [1, 0, 2, -1, -1]
    assert abs(dist[0, 0]) < 1e-6 and abs(dist[0, 1] - 1) < 1e-6 and np.isinf(dist[0, 3])
    ix.close()


def test_selfjoin_and_self_first(hip_lib, oracle_mod, corpus20k):
    """(i) a stored vector queried against the store returns itself first with distance ~0
    (tests/test_vectordb.py:168-171); self-join with exclude_self equals the oracle's."""
    n, dim, spec, rows, _ = corpus20k
    m = 3000
    ix = FlatIndex(dim, "f32")
    ix.add_synth(0, m, spec)
    idx, dist = ix.selfjoin(1, exclude_self=False, q_row0=0, q_n=200)
    assert np.array_equal(idx[:, 0], np.arange(200)) and np.all(np.abs(dist[:, 0]) < 2e-6)
    idx, dist = ix.selfjoin(5, exclude_self=True, q_row0=100, q_n=64)
    assert not np.any(idx == (100 + np.arange(64))[:, None])
    sub = rows[:m]
    for j in (0, 63):
        masked = sub.copy()
        masked[100 + j] = 0  # the oracle sees the query's own row zeroed out (score 0 < neighbours')
        r_idx, r_dist, r_sc = oracle_mod.exact_topk(masked, sub[100 + j:101 + j], 5)
        check_topk(idx[j:j + 1], dist[j:j + 1], r_idx, r_dist, r_sc, sub, sub[100 + j:101 + j], what=f"selfjoin q{j}")
    ix.close()


def test_save_load_roundtrip(hip_lib, oracle_mod, tmp_path, corpus20k):
    n, dim, spec, rows, queries = corpus20k
    ix = FlatIndex(dim, "f32")
    ix.add_synth(0, 5000, spec)
    idx, dist = ix.search(queries, 10)
    ix.save(str(tmp_path))
    ix.close()
    ix2 = FlatIndex.load(str(tmp_path), "f32")
    assert ix2.count() == 5000
    i2, d2 = ix2.search(queries, 10)
    assert np.array_equal(idx, i2) and np.array_equal(dist, d2)
    ix2.close()
    ix3 = FlatIndex.load(str(tmp_path), "bf16", True)
    i3, d3 = ix3.search(queries, 10, 40)
    assert recall_at_k(i3, idx) >= 0.999 and np.allclose(d3, dist, atol=2e-6)
    ix3.close()


def test_shard_merge_equals_whole(hip_lib, oracle_mod, corpus20k):
    """Partition property: merging the per-shard top-k (device merge kernel) equals the
    whole-corpus answer bit for bit — results are independent of the shard count."""
    import torch
    n, dim, spec, rows, queries = corpus20k
    b, k = queries.shape[0], 10
    whole = FlatIndex(dim, "f32")
    whole.add_synth(0, n, spec)
    w_idx, w_dist = whole.search(queries, k)
    whole.close()
    for g in (2, 3, 8):
        per = -(-n // g)
        cd = torch.empty((g, b, k), dtype=torch.float32, device="cuda")
        ci = torch.empty((g, b, k), dtype=torch.int64, device="cuda")
        q_dev = torch.from_numpy(queries).cuda()
        shards = []
        for s in range(g):
            lo, hi = min(n, s * per), min(n, (s + 1) * per)
            ix = FlatIndex(dim, "f32")
            ix.set_row_offset(lo)
            ix.add_synth(lo, hi - lo, spec)
            ix.search_device(q_dev.data_ptr(), b, k, 0, ci[s].data_ptr(), cd[s].data_ptr(), torch.cuda.current_stream().cuda_stream)
            shards.append(ix)
        out_i = torch.empty((b, k), dtype=torch.int64, device="cuda")
        out_d = torch.empty((b, k), dtype=torch.float32, device="cuda")
        shards[0].merge_topk_device(cd.data_ptr(), ci.data_ptr(), g, b, k, out_i.data_ptr(), out_d.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(out_i.cpu().numpy(), w_idx), g
        assert np.array_equal(out_d.cpu().numpy(), w_dist), g
        for ix in shards:
            ix.close()


def test_million_rows_properties(hip_lib, oracle_mod):
    """BASELINE config 2 size (1M x 768 fp32, batch 1): planted neighbour found first,
    returned distances equal the CPU-recomputed ones for the returned rows, ascending,
    and bf16+rerank agrees with fp32."""
    n, dim, k = 1_000_000, 768, 10
    spec = SynthSpec(seed=1234, kind=0)
    qspec = SynthSpec(seed=4242, kind=2, noise=0.5, base_kind=0, base_seed=1234, base_n=n)
    queries = oracle_mod.synth_rows(0, 4, dim, qspec)
    qn = _qn(oracle_mod, queries)
    ix = FlatIndex(dim, "f32")
    ix.add_synth(0, n, spec)
    idx, dist = ix.search(queries, k)
    t = ix.last_timing()
    assert t["bytes_streamed"] == n * dim * 4 * 1  # 4 queries share one corpus pass
    ix.close()
    for q in range(4):
        assert idx[q, 0] == oracle_mod.synth_query_src(qspec, q)
        got_rows = np.concatenate([oracle_mod.normalize_rows(oracle_mod.synth_rows(int(r), 1, dim, spec)) for r in idx[q]])
        true_d = 1.0 - got_rows.astype(np.float64) @ qn[q].astype(np.float64)
        assert np.allclose(true_d, dist[q], rtol=0, atol=1e-5 * np.abs(1 - true_d) + 2e-6)
        assert np.all(np.diff(dist[q]) >= 0)
    ixb = FlatIndex(dim, "bf16", True)
    ixb.add_synth(0, n, spec)
    ib, db = ixb.search(queries, k, 40)
    ixb.close()
    assert recall_at_k(ib, idx) >= 0.999 and np.allclose(db, dist, atol=2e-6)


# ----------------------------------------------------------------------------
# batched MFMA path (bf16 store, B >= 16)
# ----------------------------------------------------------------------------
@pytest.mark.parametrize("b", [16, 100, 128, 200, 256, 300, 512])   # 200/256/300/512 take the 256-wide tile, 16/100/128 the 128-wide
def test_mfma_batch_parity(hip_lib, oracle_mod, corpus20k, b):
    """Q.D^T on MFMA + fused top-k' + fp32 rerank vs the exact oracle.  Every corpus
    tile's first pass overflows the LDS candidate queue (all scores beat -inf), so the
    exact re-filter path is exercised here too."""
    n, dim, spec, rows, _ = corpus20k
    qspec = SynthSpec(seed=321, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=141, base_noise=0.3)
    queries = oracle_mod.synth_rows(0, b, dim, qspec)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)   # the MFMA path is what this test is about
    ix.add_synth(0, n, spec)
    idx, dist = ix.search(queries, 10, 40)
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries, 10)
    swaps = check_topk(idx, dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries), what=f"mfma b={b}")
    # check_topk already proved every differing id is a tie inside the 1e-5 tolerance, i.e.
    # tolerance-aware recall is 1.0; raw id recall only loses those in-tolerance swaps
    assert recall_at_k(idx, r_idx) >= 1.0 - swaps / idx.size - 1e-12
    # same answer as the GEMV path (queries fed one at a time stay below the MFMA threshold)
    for q in (0, b // 2, b - 1):
        i1, d1 = ix.search(queries[q:q + 1], 10, 40)
        assert np.array_equal(i1[0], idx[q]) and np.array_equal(d1[0], dist[q])
    # no rerank: bf16 x bf16 candidate scores, loose check
    i0, d0 = ix.search(queries, 10, 0)
    assert recall_at_k(i0, r_idx) >= 0.9 and np.allclose(d0, r_dist, atol=3e-3)
    ix.close()


def test_mfma_multi_tile_per_block_and_selfjoin(hip_lib, oracle_mod):
    """> 256 corpus tiles so every block walks several tiles (thresholds carried across
    tiles), ragged last tile, ragged query tile; self-join with exclude_self."""
    n, dim, b = 70001, 768, 130
    spec = SynthSpec(seed=77, kind=1, n_clusters=265, noise=0.3)
    qspec = SynthSpec(seed=78, kind=2, noise=0.5, base_kind=1, base_seed=77, base_n=n, base_n_clusters=265, base_noise=0.3)
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    queries = oracle_mod.synth_rows(0, b, dim, qspec)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)   # the MFMA path is what this test is about
    ix.add_synth(0, n, spec)
    idx, dist = ix.search(queries, 10, 40)
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries, 10)
    check_topk(idx, dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries), what="mfma 70k")
    sj_idx, sj_dist = ix.selfjoin(5, exclude_self=True, rerank_kprime=40, q_row0=1000, q_n=64)
    assert not np.any(sj_idx == (1000 + np.arange(64))[:, None])
    for j in (0, 31, 63):
        masked = rows.copy()
        masked[1000 + j] = 0
        q = rows[1000 + j:1001 + j]
        m_idx, m_dist, m_sc = oracle_mod.exact_topk(masked, q, 5)
        check_topk(sj_idx[j:j + 1], sj_dist[j:j + 1], m_idx, m_dist, m_sc, rows, q, what=f"mfma selfjoin q{j}")
    ix.close()


@pytest.mark.parametrize("variant", [4])
def test_mfma_kernel_variants_agree(hip_lib, oracle_mod, corpus20k, variant):
    """The tiled pass-kernel formulations (256x128 and 256x256 tiles of the wide kernel) are
    interchangeable: same ids and distances on the same batch."""
    n, dim, spec, rows, _ = corpus20k
    qspec = SynthSpec(seed=4242, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=141, base_noise=0.3)
    queries = oracle_mod.synth_rows(0, 256, dim, qspec)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)   # the MFMA path is what this test is about
    ix.add_synth(0, n, spec)
    i5, d5 = ix.search(queries, 10, 40)
    ix.set_option("mfma_variant", variant)
    iv, dv = ix.search(queries, 10, 40)
    assert np.array_equal(iv, i5) and np.array_equal(dv, d5)
    ix.close()


def test_mfma_wide_multi_tile_and_selfjoin(hip_lib, oracle_mod):
    """The 256-wide tile with several corpus tiles per block, a ragged last corpus tile, a
    ragged query tile (450 of 512) and the self-join exclusion."""
    n, dim, b = 150001, 768, 450
    spec = SynthSpec(seed=177, kind=1, n_clusters=387, noise=0.3)
    qspec = SynthSpec(seed=178, kind=2, noise=0.5, base_kind=1, base_seed=177, base_n=n, base_n_clusters=387, base_noise=0.3)
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    queries = oracle_mod.synth_rows(0, b, dim, qspec)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)   # the MFMA path is what this test is about
    ix.add_synth(0, n, spec)
    idx, dist = ix.search(queries, 10, 40)
    assert ix.get_stat("mfma_last_overflow") == 0
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries, 10)
    check_topk(idx, dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries), what="mfma wide 150k")
    sj_idx, sj_dist = ix.selfjoin(5, exclude_self=True, rerank_kprime=40, q_row0=149700, q_n=256)
    assert not np.any(sj_idx == (149700 + np.arange(256))[:, None])
    for j in (0, 100, 255):
        masked = rows.copy()
        masked[149700 + j] = 0
        q = rows[149700 + j:149701 + j]
        m_idx, m_dist, m_sc = oracle_mod.exact_topk(masked, q, 5)
        check_topk(sj_idx[j:j + 1], sj_dist[j:j + 1], m_idx, m_dist, m_sc, rows, q, what=f"mfma wide selfjoin q{j}")
    ix.close()


def test_mfma_adversarial_order(hip_lib, oracle_mod):
    """Scores increase with the row index (the worst case for a running threshold: every
    later row beats every earlier one), plus exact duplicates of the best rows."""
    rng = np.random.default_rng(5)
    n, dim, b = 6000, 768, 20
    q = rng.standard_normal((b, dim)).astype(np.float32)
    centre = q.mean(0)
    t = np.linspace(0.0, 1.0, n, dtype=np.float32)[:, None]
    rows = t * centre[None, :] * 4 + (1 - t) * rng.standard_normal((n, dim)).astype(np.float32)
    rows[-3:] = rows[-4]          # duplicates at the very top
    rows_n = oracle_mod.normalize_rows(rows)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)   # the MFMA path is what this test is about
    ix.add(rows)
    idx, dist = ix.search(q, 10, 48)
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows_n, q, 10)
    check_topk(idx, dist, r_idx, r_dist, r_sc, rows_n, _qn(oracle_mod, q), what="mfma adversarial")
    ix.close()


def test_mfma_overflow_fallback(hip_lib, oracle_mod, corpus20k):
    """The append path's candidate buffer overflowing (tiny capacity) or the forced-fallback
    hook must route the batch through the exact list kernel and give the same answer."""
    n, dim, spec, rows, _ = corpus20k
    qspec = SynthSpec(seed=555, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=141, base_noise=0.3)
    queries = oracle_mod.synth_rows(0, 40, dim, qspec)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)   # the MFMA path is what this test is about
    ix.add_synth(0, n, spec)
    base_idx, base_dist = ix.search(queries, 10, 40)
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries, 10)
    check_topk(base_idx, base_dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries), what="append path")
    ix.set_option("mfma_force_fallback", 1)
    i1, d1 = ix.search(queries, 10, 40)
    assert np.array_equal(i1, base_idx) and np.array_equal(d1, base_dist)
    ix.set_option("mfma_force_fallback", 0)
    ix.set_option("mfma_append_cap", 256)      # 256 < expected hits per query -> real overflow
    i2, d2 = ix.search(queries, 10, 40)
    assert np.array_equal(i2, base_idx) and np.array_equal(d2, base_dist)
    ix.close()


def test_packed_allgather_layout_merge(hip_lib, oracle_mod, corpus20k):
    """What ShardedIndex does on N GPUs, on one: every shard's packed [idx int64 | dist f32]
    result laid out back to back as all_gather_into_tensor would, merged in place with
    shard strides; plus the world-size-1 ShardedIndex path on the device."""
    import torch
    from vectordb_amd.distributed import ShardedIndex
    n, dim, spec, rows, queries = corpus20k
    b, k, g = queries.shape[0], 10, 4
    words = b * k + (b * k + 1) // 2
    gathered = torch.zeros(g * words, dtype=torch.int64, device="cuda")
    q_dev = torch.from_numpy(queries).cuda()
    per = -(-n // g)
    st = torch.cuda.current_stream().cuda_stream
    shards = []
    for s in range(g):
        lo, hi = min(n, s * per), min(n, (s + 1) * per)
        ix = FlatIndex(dim, "bf16", True)
        ix.set_row_offset(lo)
        ix.add_synth(lo, hi - lo, spec)
        base = gathered.data_ptr() + 8 * s * words
        ix.search_device(q_dev.data_ptr(), b, k, 40, base, base + 8 * b * k, st)
        shards.append(ix)
    out_i = torch.empty((b, k), dtype=torch.int64, device="cuda")
    out_d = torch.empty((b, k), dtype=torch.float32, device="cuda")
    shards[0].merge_topk_device(gathered.data_ptr() + 8 * b * k, gathered.data_ptr(), g, b, k, out_i.data_ptr(), out_d.data_ptr(),
                                st, dist_shard_stride=2 * words, idx_shard_stride=words)
    torch.cuda.synchronize()
    for ix in shards:
        ix.close()
    sh = ShardedIndex(dim, "bf16", True, device=torch.device("cuda", 0))
    sh.add_synth_partition(n, spec)
    w_idx, w_dist = sh.search(q_dev, k, 40)
    torch.cuda.synchronize()
    assert np.array_equal(out_i.cpu().numpy(), w_idx.cpu().numpy())
    assert np.array_equal(out_d.cpu().numpy(), w_dist.cpu().numpy())
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries, k)
    check_topk(w_idx.cpu().numpy(), w_dist.cpu().numpy(), r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries), what="sharded world=1")
    sh.close()


@pytest.mark.parametrize("two_phase", [False, True])
def test_sharded_exchange_on_side_stream(hip_lib, oracle_mod, corpus20k, two_phase):
    """The N>1 code path on one GPU: RCCL with a single rank, exchange forced on.  The
    all-gather + merge run on the side stream (with two_phase, the batch's own merge/rerank
    too); wait=False results are valid after wait_results(), a ring of four buffer sets is reused
    (a set again on the fourth-next call), and every answer equals the plain single-index answer."""
    import socket
    import torch
    import torch.distributed as dist
    from vectordb_amd.distributed import ShardedIndex
    n, dim, spec, rows, _ = corpus20k
    qspec = SynthSpec(seed=2024, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=141, base_noise=0.3)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sh = ShardedIndex(dim, "bf16", True, device=torch.device("cuda", 0), force_collective=True, two_phase=two_phase)
        sh.add_synth_partition(n, spec)
        plain = FlatIndex(dim, "bf16", True)
        plain.add_synth(0, n, spec)
        batches = [oracle_mod.synth_rows(16 * j, 3, dim, qspec) for j in range(5)]
        want = [plain.search(q, 10, 40) for q in batches]
        qd = [torch.from_numpy(q).cuda() for q in batches]
        got = []
        for j, q in enumerate(qd):                      # pipelined: results read only after wait_results()
            idx, dd = sh.search(q, 10, 40, wait=False)
            sh.wait_results()
            got.append((idx.cpu().numpy().copy(), dd.cpu().numpy().copy()))
        for (gi, gd), (wi, wd) in zip(got, want):
            assert np.array_equal(gi, wi) and np.array_equal(gd, wd)
        # back to back without reading in between, then one device sync: the last two answers are still intact
        outs = [sh.search(q, 10, 40, wait=False) for q in qd]
        torch.cuda.synchronize()
        for j in (3, 4):
            assert np.array_equal(outs[j][0].cpu().numpy(), want[j][0]) and np.array_equal(outs[j][1].cpu().numpy(), want[j][1])
        idx, dd = sh.search(qd[0], 10, 40)               # wait=True: ordered on the current stream
        assert np.array_equal(idx.cpu().numpy(), want[0][0]) and np.array_equal(dd.cpu().numpy(), want[0][1])
        sh.close()
        plain.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("b,k,kp", [(1024, 10, 40), (256, 30, 64), (300, 64, 0), (64, 1, 4), (40, 30, 120)])
def test_mfma_shapes_at_the_limits(hip_lib, oracle_mod, corpus20k, b, k, kp):
    """The batched path at its limits: the full 4-query-tile batch of the bench, k' = 64 (its
    maximum), k = 64 without rerank, k = 1, and k' = 120 (> 64: the batch must take the GEMV
    path and still be right)."""
    n, dim, spec, rows, _ = corpus20k
    qspec = SynthSpec(seed=777, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=n, base_n_clusters=141, base_noise=0.3)
    queries = oracle_mod.synth_rows(0, b, dim, qspec)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)   # the MFMA path is what this test is about
    ix.add_synth(0, n, spec)
    idx, dist = ix.search(queries, k, kp)
    assert ix.get_stat("mfma_last_overflow") == 0
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries, k)
    if kp:
        swaps = check_topk(idx, dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries), what=f"mfma b={b} k={k} k'={kp}")
        assert recall_at_k(idx, r_idx) >= 1.0 - swaps / idx.size - 1e-12
    else:   # bf16 x bf16 scores only: loose
        assert recall_at_k(idx, r_idx) >= 0.9 and np.allclose(dist, r_dist, atol=3e-3)
        assert np.all(np.diff(dist, axis=1) >= 0)
    ix.close()


def _masks(n, rng):
    few = np.zeros(n, bool); few[[5, n // 2, n - 1]] = True
    runs = (np.arange(n) // 37) % 3 == 1
    return {"half": rng.random(n) < 0.5, "one_percent": rng.random(n) < 0.01, "runs": runs, "three_rows": few,
            "none": np.zeros(n, bool), "all": np.ones(n, bool)}


@pytest.mark.parametrize("dtype,kp", [("f32", 0), ("bf16", 0), ("bf16", 40)])
def test_masked_search_equals_search_of_eligible_rows(hip_lib, oracle_mod, corpus20k, dtype, kp):
    """where= pre-filter (src/vectordb.py:230,259-264): the masked scan returns the exact
    top-k of the eligible rows — checked against the oracle run on those rows alone."""
    n, dim, spec, rows, queries = corpus20k
    rng = np.random.default_rng(17)
    q40 = np.concatenate([queries, rows[rng.integers(0, n, 31)] + 0.05 * rng.standard_normal((31, dim)).astype(np.float32)])
    ix = FlatIndex(dim, dtype)
    ix.add_synth(0, n, spec)
    src = oracle_mod.round_bf16(rows) if (dtype == "bf16" and kp == 0) else rows
    for name, mask in _masks(n, rng).items():
        allowed = np.flatnonzero(mask)
        for b in (1, 3, 40):                                   # 40 would take the MFMA path unmasked; masked stays on the GEMV scan
            q = q40[:b]
            idx, dist = ix.search_masked(q, 10, mask, kp)
            if name == "all":
                i0, d0 = ix.search(q, 10, kp)
                if b < 3:                                      # same kernel, same order of operations
                    assert np.array_equal(idx, i0) and np.array_equal(dist, d0)
            if allowed.size == 0:
                assert np.all(idx == -1) and np.all(np.isinf(dist))
                continue
            kk = min(10, allowed.size)
            r_idx, r_dist, r_sc = oracle_mod.exact_topk(src[allowed], q, kk)
            assert np.all(idx[:, kk:] == -1) and np.all(np.isinf(dist[:, kk:]))
            assert np.all(mask[idx[:, :kk]])
            check_topk(idx[:, :kk], dist[:, :kk], allowed[r_idx], r_dist, r_sc, src, _qn(oracle_mod, q), what=f"masked {name} {dtype} b={b}")
    with pytest.raises(Exception):
        ix.search_masked(q40[:1], 10, np.ones(n - 1, bool))
    ix.close()


def test_masked_search_other_dims_and_device_entry(hip_lib, oracle_mod):
    """generic-dim kernel (64), the 384 fast path, and vdb_search_masked_dev with a device bitmap."""
    import torch
    rng = np.random.default_rng(23)
    for dim, n in ((64, 3001), (384, 5003)):
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        rows_n = oracle_mod.normalize_rows(rows)
        q = rows[:5] + 0.1 * rng.standard_normal((5, dim)).astype(np.float32)
        mask = rng.random(n) < 0.3
        allowed = np.flatnonzero(mask)
        ix = FlatIndex(dim, "f32")
        ix.add(rows)
        idx, dist = ix.search_masked(q, 7, mask)
        r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows_n[allowed], q, 7)
        check_topk(idx, dist, allowed[r_idx], r_dist, r_sc, rows_n, _qn(oracle_mod, q), what=f"masked dim={dim}")
        bits = np.zeros(((n + 31) // 32) * 4, np.uint8)
        packed = np.packbits(mask, bitorder="little")
        bits[: packed.size] = packed
        d_bits = torch.from_numpy(bits.view(np.int32)).cuda()
        d_q = torch.from_numpy(q).cuda()
        d_i = torch.empty((5, 7), dtype=torch.int64, device="cuda")
        d_d = torch.empty((5, 7), dtype=torch.float32, device="cuda")
        ix.search_masked_device(d_q.data_ptr(), 5, 7, 0, d_bits.data_ptr(), n, d_i.data_ptr(), d_d.data_ptr(),
                                torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(d_i.cpu().numpy(), idx) and np.array_equal(d_d.cpu().numpy(), dist)
        ix.close()


def test_random_shapes_against_oracle(hip_lib, oracle_mod):
    """Seeded sweep over odd shapes: dims that are not a multiple of the 16-byte chunk, stores smaller
    than a wave / a tile / k, k up to 300, batches 0..7, both dtypes, with and without rerank, masked or not."""
    rng = np.random.default_rng(2024)
    dims = [1, 3, 8, 31, 64, 100, 384, 385, 768, 1000]
    for case in range(48):
        dim = dims[case % len(dims)]
        n = int(rng.choice([1, 2, 7, 63, 64, 65, 200, 513, 1500, 4099]))
        b = int(rng.integers(0, 8))
        k = int(rng.choice([1, 2, 10, 33, 64, 65, 128, 300]))
        dtype = "bf16" if case % 3 == 1 else "f32"
        kp = 0 if dtype == "f32" or case % 2 else min(1024, max(k, 4 * k))
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        if n > 3:
            rows[3] = rows[1]                                    # a duplicate row: tie broken by index
        q = rng.standard_normal((b, dim)).astype(np.float32)
        rows_n = oracle_mod.normalize_rows(rows)
        src = oracle_mod.round_bf16(rows_n) if (dtype == "bf16" and kp == 0) else rows_n
        ix = FlatIndex(dim, dtype)
        ix.add(rows)
        what = f"case {case}: n={n} dim={dim} b={b} k={k} {dtype} kp={kp}"
        idx, dist = ix.search(q, k, kp)
        assert idx.shape == (b, k) and dist.shape == (b, k), what
        kk = min(k, n)
        if b:
            r_idx, r_dist, r_sc = oracle_mod.exact_topk(src, q, kk)
            assert np.all(idx[:, kk:] == -1) and np.all(np.isinf(dist[:, kk:])), what
            check_topk(idx[:, :kk], dist[:, :kk], r_idx, r_dist, r_sc, src, _qn(oracle_mod, q), what=what)
            mask = rng.random(n) < 0.5
            allowed = np.flatnonzero(mask)
            mi, md = ix.search_masked(q, k, mask, kp)
            km = min(k, allowed.size)
            assert np.all(mi[:, km:] == -1) and np.all(np.isinf(md[:, km:])), what
            if km:
                m_idx, m_dist, m_sc = oracle_mod.exact_topk(src[allowed], q, km)
                check_topk(mi[:, :km], md[:, :km], allowed[m_idx], m_dist, m_sc, src, _qn(oracle_mod, q), what=what + " masked")
        ix.close()


def test_gemv_batches_beyond_one_query_chunk(hip_lib, oracle_mod, corpus20k):
    """f32 stores (and masked searches) stay on the GEMV scan at any batch size: 150 queries run as
    three 64-query chunks of 4-query passes; every row equals the oracle and the batch-1 answer."""
    n, dim, spec, rows, queries = corpus20k
    rng = np.random.default_rng(5)
    q = rows[rng.integers(0, n, 150)] + 0.05 * rng.standard_normal((150, dim)).astype(np.float32)
    ix = FlatIndex(dim, "f32")
    ix.add_synth(0, n, spec)
    idx, dist = ix.search(q, 10)
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, q, 10)
    check_topk(idx, dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, q), what="f32 b=150")
    for j in (0, 63, 64, 127, 128, 149):
        i1, d1 = ix.search(q[j:j + 1], 10)
        assert np.array_equal(i1[0], idx[j]) and np.array_equal(d1[0], dist[j])
    mask = rng.random(n) < 0.25
    allowed = np.flatnonzero(mask)
    mi, md = ix.search_masked(q, 10, mask)
    m_idx, m_dist, m_sc = oracle_mod.exact_topk(rows[allowed], q, 10)
    check_topk(mi, md, allowed[m_idx], m_dist, m_sc, rows, _qn(oracle_mod, q), what="f32 masked b=150")
    assert ix.last_timing()["scan_ms"] > 0
    ix.close()


@pytest.mark.parametrize("b", [100, 300])                       # 128-wide and 256-wide query tiles
def test_mfma_last_pass_skips_the_sampled_tiles(hip_lib, oracle_mod, corpus20k, b):
    """Two append passes (forced at 20k rows by a small candidate buffer): the last pass leaves out the tiles
    the sample pass multiplied and the final selection is seeded with the sample's list.  Exact against the
    oracle, and identical to the plain form (mfma_skip_sample = 0) that re-multiplies every tile."""
    n, dim, spec, rows, queries = corpus20k
    rng = np.random.default_rng(b)
    q = rows[rng.integers(0, n, b)] + 0.05 * rng.standard_normal((b, dim)).astype(np.float32)
    ix = FlatIndex(dim, "bf16", True)
    ix.add_synth(0, n, spec)
    ix.set_option("mfma_cost_model", 0)
    ix.set_option("mfma_dense_tiles", 16)                      # (round 3's default sample, 64 tiles, would cover this small store in one pass)
    ix.set_option("mfma_append_cap", 512)                      # R = 512/(4*40) = 3.2 -> dense sample, 36-tile sample at stride 2, rest
    out = {}
    for skip in (1, 0):
        ix.set_option("mfma_skip_sample", skip)
        out[skip] = ix.search(q, 10, 40)
        assert ix.get_stat("mfma_last_overflow") == 0 and ix.get_stat("mfma_last_hits") > 0
        assert ix.get_stat("mfma_last_skipped_tiles") == (36 if skip else 0)          # ceil(16 * sqrt(79/16)) sampled tiles
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1], out[0][1])
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, q, 10)
    swaps = check_topk(out[1][0], out[1][1], r_idx, r_dist, r_sc, rows, _qn(oracle_mod, q), what=f"skip-sample b={b}")
    assert recall_at_k(out[1][0], r_idx) >= 1.0 - swaps / out[1][0].size - 1e-12
    # self-join form (the excluded row may sit in a sampled tile or in the rest)
    sj1 = ix.selfjoin(5, exclude_self=True, rerank_kprime=40, q_row0=1000, q_n=b)
    ix.set_option("mfma_skip_sample", 1)
    sj2 = ix.selfjoin(5, exclude_self=True, rerank_kprime=40, q_row0=1000, q_n=b)
    assert np.array_equal(sj1[0], sj2[0]) and not np.any(sj2[0] == (1000 + np.arange(b))[:, None])
    ix.close()


def test_mfma_two_passes_at_a_million_rows_equal_the_gemv_scan(hip_lib):
    """Default candidate buffer, 1M rows (3 907 tiles: dense sample, strided sample pass, rest): the batched
    path returns bit for bit what the GEMV scan returns after the fp32 rerank."""
    n, dim = 1_000_000, 768
    spec = SynthSpec(seed=77, kind=1, n_clusters=1000, noise=0.3)
    qspec = SynthSpec(seed=78, kind=2, noise=0.5, base_kind=1, base_seed=77, base_n=n, base_n_clusters=1000, base_noise=0.3)
    import torch
    ix = FlatIndex(dim, "bf16", True)
    ix.add_synth(0, n, spec)
    qd = torch.empty((300, dim), dtype=torch.float32, device="cuda")
    ix.synth_rows_device(qd.data_ptr(), 0, 300, qspec, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    q = qd.cpu().numpy()
    ix.set_option("mfma_cost_model", 0)
    res = {}
    for b in (64, 300):
        res[b] = ix.search(q[:b], 10, 40)
        assert ix.get_stat("mfma_last_hits") > 0 and ix.get_stat("mfma_last_overflow") == 0
        assert ix.get_stat("mfma_last_skipped_tiles") > 100
    ix.set_option("batch_mfma_min_b", 1 << 20)                  # everything on the GEMV scan
    gi, gd = ix.search(q, 10, 40)
    for b in (64, 300):
        assert np.array_equal(res[b][0], gi[:b]) and np.array_equal(res[b][1], gd[:b])
    ix.close()


@pytest.mark.parametrize("dim,k", [(8192, 5), (32768, 5), (32768, 1000)])
def test_largest_dimensions(hip_lib, oracle_mod, dim, k):
    """The generic scan keeps the query in LDS: dim up to 32 768 (128 KiB) beside k = 1000 lists (32 KiB)
    is the whole 160 KiB; beyond that vdb_create refuses."""
    rng = np.random.default_rng(dim + k)
    n = 1200
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    q = rows[[7, 900]] + 0.01 * rng.standard_normal((2, dim)).astype(np.float32)
    rows_n = oracle_mod.normalize_rows(rows)
    for dtype, kp in (("f32", 0), ("bf16", min(1024, 4 * k))):
        ix = FlatIndex(dim, dtype)
        ix.add(rows)
        idx, dist = ix.search(q, k, kp)
        r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows_n, q, k)
        check_topk(idx, dist, r_idx, r_dist, r_sc, rows_n, _qn(oracle_mod, q), what=f"dim={dim} k={k} {dtype}")
        ix.close()
    with pytest.raises(Exception, match="dim must be"):
        FlatIndex(32769, "f32")


def test_mfma_small_stores_and_odd_batches(hip_lib, oracle_mod):
    """The batched path at the small end: stores of 16..48 tiles (4 096 rows is its minimum), batches that
    pad to one or two query tiles of either kernel, k' from 4 to 64 (k' >= 4k: with fewer candidates the bf16
    ranking may legitimately miss a boundary row, which is what the rerank margin is for)."""
    rng = np.random.default_rng(99)
    dim = 768
    for n, b, k, kp in ((4096, 3, 1, 4), (4097, 17, 10, 40), (5000, 130, 5, 64), (8191, 257, 10, 40), (12345, 40, 16, 64)):
        rows = rng.standard_normal((n, dim)).astype(np.float32)
        rows[n // 2] = rows[n // 3]                                   # one exact duplicate pair
        q = rows[rng.integers(0, n, b)] + 0.1 * rng.standard_normal((b, dim)).astype(np.float32)
        rows_n = oracle_mod.normalize_rows(rows)
        ix = FlatIndex(dim, "bf16", True)
        ix.add(rows)
        ix.set_option("mfma_cost_model", 0)
        idx, dist = ix.search(q, k, kp)
        assert ix.get_stat("mfma_last_hits") > 0, (n, b)                # served by the MFMA path
        r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows_n, q, k)
        swaps = check_topk(idx, dist, r_idx, r_dist, r_sc, rows_n, _qn(oracle_mod, q), what=f"n={n} b={b} k={k} kp={kp}")
        if kp >= 4 * k:
            assert recall_at_k(idx, r_idx) >= 1.0 - swaps / idx.size - 1e-12
        ix.set_option("batch_mfma_min_b", 1 << 20)
        gi, gd = ix.search(q, k, kp)                                    # the GEMV scan on the same store
        if kp >= 4 * k:
            assert np.array_equal(gi, idx) and np.array_equal(gd, dist)
        ix.close()


@pytest.mark.parametrize("dtype,kp", [("f32", 0), ("bf16", 40)])
def test_two_phase_search_equals_one_call(hip_lib, oracle_mod, corpus20k, dtype, kp):
    """vdb_search_scan_dev + vdb_search_finish_dev (phase 2 on another stream, both workspace slots in flight)
    return bit for bit what vdb_search_dev returns; batches that do not split say so and do nothing."""
    import torch
    n, dim, spec, rows, queries = corpus20k
    ix = FlatIndex(dim, dtype)
    ix.add_synth(0, n, spec)
    rng = np.random.default_rng(7)
    main, side = torch.cuda.current_stream(), torch.cuda.Stream()
    for b in (1, 3, 9, 64):
        qa = torch.from_numpy(rows[rng.integers(0, n, b)] + 0.05 * rng.standard_normal((b, dim)).astype(np.float32)).cuda()
        qb = torch.from_numpy(rows[rng.integers(0, n, b)] + 0.05 * rng.standard_normal((b, dim)).astype(np.float32)).cuda()
        ref = []
        for q in (qa, qb):
            ri = torch.empty((b, 10), dtype=torch.int64, device="cuda"); rd = torch.empty((b, 10), dtype=torch.float32, device="cuda")
            ix.set_option("batch_mfma_min_b", 1 << 20)                    # the reference call on the GEMV scan as well
            ix.search_device(q.data_ptr(), b, 10, kp, ri.data_ptr(), rd.data_ptr(), main.cuda_stream)
            ref.append((ri, rd))
        torch.cuda.synchronize()
        outs = []
        for slot, q in enumerate((qa, qb)):                               # two scans queued back to back, one per slot
            assert ix.search_scan_device(q.data_ptr(), b, 10, kp, slot, main.cuda_stream)
            ev = torch.cuda.Event(); ev.record(main)
            oi = torch.empty((b, 10), dtype=torch.int64, device="cuda"); od = torch.empty((b, 10), dtype=torch.float32, device="cuda")
            with torch.cuda.stream(side):
                side.wait_event(ev)
                ix.search_finish_device(slot, oi.data_ptr(), od.data_ptr(), side.cuda_stream)
            outs.append((oi, od))
        torch.cuda.synchronize()
        for (oi, od), (ri, rd) in zip(outs, ref):
            assert torch.equal(oi, ri) and torch.equal(od, rd), (dtype, b)
        ix.set_option("batch_mfma_min_b", 3)
    q65 = torch.zeros((65, dim), dtype=torch.float32, device="cuda")
    assert not ix.search_scan_device(q65.data_ptr(), 65, 10, kp, 0, main.cuda_stream)          # more than one scan round
    if dtype == "bf16":
        ix.set_option("mfma_cost_model", 0)
        assert not ix.search_scan_device(q65.data_ptr(), 32, 10, kp, 0, main.cuda_stream)      # would take the MFMA path
    with pytest.raises(Exception):
        ix.search_finish_device(0, q65.data_ptr(), q65.data_ptr(), main.cuda_stream)           # nothing pending
    ix.close()


def test_mfma_all_scores_negative_or_zero(hip_lib, oracle_mod):
    """Thresholds that are not positive (every cosine negative; a store of zero rows): the integer block
    filter of the pass kernels must stand aside and the float compares decide."""
    rng = np.random.default_rng(31)
    n, dim = 6000, 768
    base = rng.standard_normal(dim).astype(np.float32)
    rows = -base[None, :] + 0.4 * rng.standard_normal((n, dim)).astype(np.float32)      # all rows point away from `base`
    for b in (40, 300):                                                                  # 128-wide and 256-wide kernels
        q = base[None, :] + 0.4 * rng.standard_normal((b, dim)).astype(np.float32)
        rows_n = oracle_mod.normalize_rows(rows)
        ix = FlatIndex(dim, "bf16", True)
        ix.add(rows)
        ix.set_option("mfma_cost_model", 0)
        idx, dist = ix.search(q, 10, 40)
        assert ix.get_stat("mfma_last_hits") > 0
        r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows_n, q, 10)
        assert np.all(r_sc < 0)                                                          # the premise: every score negative
        check_topk(idx, dist, r_idx, r_dist, r_sc, rows_n, _qn(oracle_mod, q), what=f"negative scores b={b}")
        ix.close()
    zr = np.zeros((5000, dim), np.float32)
    zr[:50] = rng.standard_normal((50, dim)).astype(np.float32)                          # 50 real rows, 4 950 zero rows (score 0)
    ix = FlatIndex(dim, "bf16", True)
    ix.add(zr)
    ix.set_option("mfma_cost_model", 0)
    q = -zr[:40] + 0.01 * rng.standard_normal((40, dim)).astype(np.float32)              # best matches are the zero rows (0 > negative)
    idx, dist = ix.search(q, 10, 40)
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(oracle_mod.normalize_rows(zr), q, 10)
    assert np.allclose(dist, r_dist, atol=2e-6) and np.all(np.diff(dist, axis=1) >= 0)
    assert np.array_equal(idx, r_idx)                                                    # ties at score 0 broken by row index
    ix.close()


@pytest.mark.parametrize("dtype,kp", [("f32", 0), ("bf16", 40)])
def test_nan_and_inf_inputs_behave_like_the_oracle(hip_lib, oracle_lib, dtype, kp):
    """A stored row holding NaN or Inf normalises to NaN and is never returned; a query holding NaN or Inf
    returns nothing (-1 / +inf).  Same on the GEMV scan (b = 4), the batched path (b = 40) and the oracle."""
    rng = np.random.default_rng(3)
    dim, n = 768, 5000
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows[17, 5] = np.nan
    rows[99, 7] = np.inf
    q = rows[[3, 17, 99, 200]] + 0.01
    out = {}
    for name, lib in (("hip", None), ("ora", oracle_lib)):
        ix = FlatIndex(dim, dtype, True, lib=lib)
        ix.add(rows)
        if lib is None:
            ix.set_option("mfma_cost_model", 0)
        out[name] = [ix.search(np.concatenate([q] * (b // 4)), 5, kp) for b in (4, 40)]
        ix.close()
    for (hi, hd), (oi, od) in zip(out["hip"], out["ora"]):
        assert np.array_equal(hi, oi) and np.allclose(hd, od, atol=2e-6)
        assert not np.isin(hi, [17, 99]).any() and np.all(hi[1] == -1) and np.all(hi[2] == -1) and hi[0, 0] == 3
