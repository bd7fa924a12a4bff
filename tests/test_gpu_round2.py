"""GPU suite, round 2: the streaming MFMA kernel (2..64 queries), the row-sharded handle behind the C ABI
(vdb_create_sharded, several shards on the one GPU of the test box), store-file validation and the batched
path's query chunking.  Everything goes through the C ABI and is checked against the CPU oracle or against the
single-store answer, bit for bit where the contract says so."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

from vectordb_amd import FlatIndex, SynthSpec
from vectordb_amd._capi import VdbError
from tests.util import check_topk

pytestmark = pytest.mark.gpu


def _qn(oracle_mod, queries):
    return oracle_mod.normalize_rows(np.ascontiguousarray(queries, np.float32))


@pytest.fixture(scope="module")
def corpus33k(oracle_mod):
    n, dim = 33003, 768          # ragged against 16-row wave tiles, 256-row protocol tiles and the dense sample
    spec = SynthSpec(seed=4321, kind=1, n_clusters=181, noise=0.3)
    qspec = SynthSpec(seed=77, kind=2, noise=0.5, base_kind=1, base_seed=4321, base_n=n, base_n_clusters=181, base_noise=0.3)
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    queries = oracle_mod.synth_rows(0, 64, dim, qspec)
    return n, dim, spec, rows, queries


@pytest.mark.parametrize("nt", [1, 0])
def test_stream_kernel_against_oracle(hip_lib, oracle_mod, corpus33k, nt):
    """Every query-block count of the streaming kernel (1, 2, 4 blocks of 16) with ragged batches, plain and
    non-temporal loads; the path taken is asserted, not assumed."""
    n, dim, spec, rows, queries = corpus33k
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)
    ix.set_option("stream_nt_loads", nt)
    ix.add_synth(0, n, spec)
    qn = _qn(oracle_mod, queries)
    for b in (2, 3, 16, 17, 32, 33, 64):
        for k, kp in ((10, 40), (1, 8), (30, 64)):
            idx, dist = ix.search(queries[:b], k, kp)
            assert ix.get_stat("last_path") == 6, "the streaming kernel must have served this batch"
            assert ix.get_stat("mfma_last_overflow") == 0
            r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries[:b], k)
            check_topk(idx, dist, r_idx, r_dist, r_sc, rows, qn[:b], what=f"stream b={b} k={k} nt={nt}")
    # the batch-1 GEMV answer is the same answer
    i64, d64 = ix.search(queries, 10, 40)
    for q in (0, 31, 63):
        i1, d1 = ix.search(queries[q:q + 1], 10, 40)
        assert ix.get_stat("last_path") == 0
        assert np.array_equal(i1[0], i64[q]) and np.array_equal(d1[0], d64[q])
    ix.close()


def test_stream_kernel_multi_pass_and_selfjoin(hip_lib, oracle_mod):
    """Enough tiles that the protocol runs its sample pass and a final pass that skips the sampled tiles, several
    tiles per wave, and the self-join exclusion (a 64-row block of rows-as-queries)."""
    n, dim = 300_007, 768
    spec = SynthSpec(seed=909, kind=1, n_clusters=547, noise=0.3)
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    qspec = SynthSpec(seed=910, kind=2, noise=0.5, base_kind=1, base_seed=909, base_n=n, base_n_clusters=547, base_noise=0.3)
    queries = oracle_mod.synth_rows(0, 40, dim, qspec)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)
    ix.set_option("mfma_append_cap", 4096)      # smaller candidate buffers ...
    ix.set_option("mfma_one_pass_span", 2)      # ... and no single-pass shortcut: the protocol takes two append passes at this size (1 172 tiles, 64 of them in the dense sample)
    ix.add_synth(0, n, spec)
    idx, dist = ix.search(queries, 10, 40)
    assert ix.get_stat("last_path") == 6 and ix.get_stat("mfma_last_overflow") == 0
    assert ix.get_stat("mfma_last_skipped_tiles") > 0
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries, 10)
    check_topk(idx, dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries), what="stream 300k")
    # same answer with the tiled kernel and with the last pass re-multiplying the sampled tiles
    ix.set_option("mfma_variant", 5)
    i5, d5 = ix.search(queries, 10, 40)
    assert ix.get_stat("last_path") == 4
    assert np.array_equal(i5, idx) and np.array_equal(d5, dist)
    ix.set_option("mfma_variant", 6)
    ix.set_option("mfma_skip_sample", 0)
    i6, d6 = ix.search(queries, 10, 40)
    assert np.array_equal(i6, idx) and np.array_equal(d6, dist)
    ix.set_option("mfma_skip_sample", 1)
    ix.set_option("stream_nt_loads", 0)         # plain loads: same bits
    i7, d7 = ix.search(queries, 10, 40)
    assert ix.get_stat("last_path") == 6 and np.array_equal(i7, idx) and np.array_equal(d7, dist)
    ix.set_option("stream_nt_loads", 1)
    sj_idx, sj_dist = ix.selfjoin(5, exclude_self=True, rerank_kprime=40, q_row0=299_950, q_n=57)
    assert ix.get_stat("last_path") == 6
    assert not np.any(sj_idx == (299_950 + np.arange(57))[:, None])
    for j in (0, 30, 56):
        masked = rows.copy()
        masked[299_950 + j] = 0
        q = rows[299_950 + j:299_951 + j]
        m_idx, m_dist, m_sc = oracle_mod.exact_topk(masked, q, 5)
        check_topk(sj_idx[j:j + 1], sj_dist[j:j + 1], m_idx, m_dist, m_sc, rows, q, what=f"stream selfjoin q{j}")
    ix.close()


def test_stream_kernel_overflow_fallback_and_bad_thresholds(hip_lib, oracle_mod, corpus33k):
    """Tiny candidate buffers -> the device-side overflow flag routes through the exact list kernel; and queries
    whose cosines are all <= 0 (non-positive thresholds take the float compare)."""
    n, dim, spec, rows, queries = corpus33k
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)
    ix.add_synth(0, n, spec)
    base_i, base_d = ix.search(queries[:24], 10, 40)
    ix.set_option("mfma_append_cap", 256)       # the smallest buffers: more append passes, possibly a real overflow
    i2, d2 = ix.search(queries[:24], 10, 40)
    assert np.array_equal(i2, base_i) and np.array_equal(d2, base_d)
    ix.set_option("mfma_append_cap", 16384)
    ix.set_option("mfma_force_fallback", 1)     # the overflow route itself
    i3, d3 = ix.search(queries[:24], 10, 40)
    assert ix.get_stat("mfma_last_overflow") == 1
    assert np.array_equal(i3, base_i) and np.array_equal(d3, base_d)
    ix.set_option("mfma_force_fallback", 0)
    neg = -np.abs(queries[:10])            # rows are centroid + noise around 0: mirror the query so that most cosines go negative
    idx, dist = ix.search(neg, 10, 40)
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, neg, 10)
    check_topk(idx, dist, r_idx, r_dist, r_sc, rows, _qn(oracle_mod, neg), what="stream negative thresholds")
    ix.close()


def test_wide_kernel_against_the_other_families(hip_lib, oracle_mod):
    """The wide tile (256 x 256 on v_mfma_f32_16x16x32_bf16) against the same kernel on 128-wide query tiles: same candidates, so the same ids
    and distances bit for bit — several corpus tiles per block, ragged last corpus tile, ragged query tile, self-join
    exclusion, and the oracle's answer."""
    n, dim, b = 140_003, 768, 450
    spec = SynthSpec(seed=271, kind=1, n_clusters=374, noise=0.3)
    qspec = SynthSpec(seed=272, kind=2, noise=0.5, base_kind=1, base_seed=271, base_n=n, base_n_clusters=374, base_noise=0.3)
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    queries = oracle_mod.synth_rows(0, b, dim, qspec)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)
    ix.add_synth(0, n, spec)
    assert ix.get_stat("rows") == n
    i16, d16 = ix.search(queries, 10, 40)
    assert ix.get_stat("last_path") == 5 and ix.get_stat("mfma_last_overflow") == 0
    sj16 = ix.selfjoin(5, exclude_self=True, rerank_kprime=40, q_row0=n - 300, q_n=300)
    ix.set_option("mfma_variant", 4)
    i4, d4 = ix.search(queries, 10, 40)
    assert ix.get_stat("last_path") == 4
    sj4 = ix.selfjoin(5, exclude_self=True, rerank_kprime=40, q_row0=n - 300, q_n=300)
    assert np.array_equal(i16, i4) and np.array_equal(d16, d4)
    assert np.array_equal(sj16[0], sj4[0]) and np.array_equal(sj16[1], sj4[1])
    assert not np.any(sj16[0] == (n - 300 + np.arange(300))[:, None])
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries[:64], 10)
    check_topk(i16[:64], d16[:64], r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries[:64]), what="wide 16x16x32")
    ix.close()


def test_batched_path_chunks_large_batches(hip_lib, oracle_mod):
    """More queries than one batched launch takes (4096): the engine chunks them; every chunk's answers equal the
    answers of the same queries asked alone."""
    n, dim, b = 20_000, 768, 4096 + 300
    spec = SynthSpec(seed=31, kind=1, n_clusters=141, noise=0.3)
    qspec = SynthSpec(seed=32, kind=2, noise=0.5, base_kind=1, base_seed=31, base_n=n, base_n_clusters=141, base_noise=0.3)
    queries = oracle_mod.synth_rows(0, b, dim, qspec)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)
    ix.add_synth(0, n, spec)
    idx, dist = ix.search(queries, 10, 40)
    for lo, hi in ((0, 8), (4090, 4102), (b - 5, b)):
        i1, d1 = ix.search(queries[lo:hi], 10, 40)
        assert np.array_equal(i1, idx[lo:hi]) and np.array_equal(d1, dist[lo:hi])
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    sel = np.r_[0:4, 4094:4098, b - 4:b]
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries[sel], 10)
    check_topk(idx[sel], dist[sel], r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries[sel]), what="chunked batch")
    ix.close()


@pytest.mark.parametrize("dtype,kp", [("f32", 0), ("bf16", 40)])
@pytest.mark.parametrize("g", [1, 2, 3, 8])
def test_sharded_handle_equals_single_store(hip_lib, oracle_mod, corpus33k, g, dtype, kp):
    """vdb_create_sharded with g shards (all on the one GPU of this box): add through the router, search through
    the router — bit-identical to the single store, and equal to the oracle."""
    n, dim, spec, rows, queries = corpus33k
    single = FlatIndex(dim, dtype, True)
    single.add_synth(0, n, spec)
    sh = FlatIndex(dim, dtype, True, devices=[0] * g)
    with pytest.raises(VdbError):
        sh.add_synth(0, 10, spec)            # the partition is not fixed yet
    sh.reserve(n)
    assert sh.get_stat("n_shards") == g and sh.get_stat("rows_per_shard") % 256 == 0
    sh.add_synth(0, 20_000, spec)            # two appends that straddle shard boundaries
    sh.add_synth(20_000, n - 20_000, spec)
    assert sh.count() == n
    assert sh.get_stat("p2p_direct_shards") == g
    qn = _qn(oracle_mod, queries)
    for b in (1, 4, 40):
        for k in (1, 10, 30):
            if kp and kp < k:
                continue
            i_s, d_s = single.search(queries[:b], k, kp)
            i_g, d_g = sh.search(queries[:b], k, kp)
            assert np.array_equal(i_g, i_s) and np.array_equal(d_g, d_s), f"g={g} b={b} k={k}"
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries[:9], 10)
    i_g, d_g = sh.search(queries[:9], 10, kp)
    check_topk(i_g, d_g, r_idx, r_dist, r_sc, rows, qn[:9], what=f"sharded g={g} {dtype}")
    # fetch routes global indices to their shards
    per = sh.get_stat("rows_per_shard")
    pick = np.array([0, 1, min(per, n) - 1, per % n, n - 1, 12345], np.int64)
    assert np.array_equal(sh.fetch(pick), single.fetch(pick))
    t = sh.last_timing()
    assert t["scan_ms"] > 0 and t["bytes_streamed"] > 0
    single.close()
    sh.close()


def test_sharded_host_rows_k_larger_than_a_shard_and_device_entry(hip_lib, oracle_mod):
    """Rows added from host memory through the router; k larger than a shard's row count (shards pad with -1, the
    merge ignores the padding); the device-pointer entry point with torch buffers."""
    import torch
    rng = np.random.default_rng(11)
    n, dim = 700, 96
    rows = rng.standard_normal((n, dim)).astype(np.float32)
    rows_n = oracle_mod.normalize_rows(rows)
    queries = rows[[5, 300, 650]] + 0.05 * rng.standard_normal((3, dim)).astype(np.float32)
    sh = FlatIndex(dim, "f32", devices=[0, 0, 0])
    sh.reserve(n)                     # per = 256: shards hold 256 / 256 / 188 rows
    sh.add(rows[:300]); sh.add(rows[300:])
    assert sh.count() == n
    for k in (10, 300, 1000):
        idx, dist = sh.search(queries, k)
        r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows_n, queries, k)
        check_topk(idx, dist, r_idx, r_dist, r_sc, rows_n, _qn(oracle_mod, queries), what=f"sharded host rows k={k}")
    q_t = torch.from_numpy(queries).cuda()
    oi = torch.empty((3, 10), dtype=torch.int64, device="cuda")
    od = torch.empty((3, 10), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):                # back-to-back calls reuse the gather buffer behind stream order only
        sh.search_device(q_t.data_ptr(), 3, 10, 0, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    idx, dist = sh.search(queries, 10)
    assert np.array_equal(oi.cpu().numpy(), idx) and np.array_equal(od.cpu().numpy(), dist)
    # rows beyond the reserved total go to the last shard and keep their global indices
    extra = rng.standard_normal((100, dim)).astype(np.float32)
    sh.add(extra)
    all_n = oracle_mod.normalize_rows(np.concatenate([rows, extra]))
    q2 = extra[[3, 77]]
    idx, dist = sh.search(q2, 5)
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(all_n, q2, 5)
    check_topk(idx, dist, r_idx, r_dist, r_sc, all_n, _qn(oracle_mod, q2), what="sharded overflow rows")
    assert idx[0, 0] == n + 3 and idx[1, 0] == n + 77
    # what a sharded handle does not serve says so (round 3: masks, self-join and device appends are served: test_gpu_round3.py)
    with pytest.raises(VdbError) as e:
        sh.set_row_offset(5)
    assert e.value.code == -5
    import torch
    t = torch.zeros((1, dim), device="cuda")
    o_i = torch.zeros((1, 3), dtype=torch.int64, device="cuda"); o_d = torch.zeros((1, 3), device="cuda")
    with pytest.raises(VdbError):
        sh.search_device_noalloc(t.data_ptr(), 1, 3, 0, o_i.data_ptr(), o_d.data_ptr())
    sh.close()


def test_sharded_save_load_roundtrip(hip_lib, oracle_mod, tmp_path, corpus33k):
    """One file format whatever the handle: sharded -> single, single -> sharded, sharded -> sharded with another
    shard count; searches agree bit for bit."""
    n, dim, spec, rows, queries = corpus33k
    a = FlatIndex(dim, "bf16", True, devices=[0, 0, 0])
    a.reserve(n)
    a.add_synth(0, n, spec)
    ia, da = a.search(queries[:5], 10, 40)
    a.save(str(tmp_path / "s3"))
    a.close()
    b = FlatIndex.load(str(tmp_path / "s3"), "bf16", True)
    assert b.count() == n
    ib, db = b.search(queries[:5], 10, 40)
    assert np.array_equal(ia, ib) and np.array_equal(da, db)
    b.save(str(tmp_path / "s1"))
    b.close()
    c = FlatIndex.load(str(tmp_path / "s1"), "bf16", True, devices=[0, 0])
    assert c.count() == n and c.get_stat("n_shards") == 2
    ic, dc = c.search(queries[:5], 10, 40)
    assert np.array_equal(ia, ic) and np.array_equal(da, dc)
    c.close()


def test_store_file_validation(hip_lib, tmp_path):
    """vdb_load rejects a truncated file and headers that lie (negative / huge row count, bad dim, bad dtype);
    vdb_save leaves no temporary file behind."""
    rng = np.random.default_rng(2)
    ix = FlatIndex(32, "f32")
    ix.add(rng.standard_normal((100, 32)).astype(np.float32))
    d = tmp_path / "store"
    ix.save(str(d))
    ix.close()
    assert sorted(os.listdir(d)) == ["vectors.vdb"]
    good = (d / "vectors.vdb").read_bytes()
    assert len(good) == 32 + 100 * 32 * 4
    FlatIndex.load(str(d), "f32").close()

    def corrupt(name, data):
        dd = tmp_path / name
        dd.mkdir()
        (dd / "vectors.vdb").write_bytes(data)
        with pytest.raises(VdbError) as e:
            FlatIndex.load(str(dd), "f32")
        assert e.value.code == -4, name      # VDB_ERR_IO
        with pytest.raises(VdbError):
            FlatIndex.load(str(dd), "f32", devices=[0, 0])

    hdr = lambda dim, dt, n: b"VDBFLAT1" + struct.pack("<iiqq", dim, dt, n, 0)   # noqa: E731
    corrupt("truncated", good[:-40])
    corrupt("longer", good + b"\0" * 8)
    corrupt("negative_n", hdr(32, 0, -5) + good[32:])
    corrupt("huge_n", hdr(32, 0, 1 << 40) + good[32:])
    corrupt("bad_dim", hdr(0, 0, 100) + good[32:])
    corrupt("bad_dtype", hdr(32, 7, 100) + good[32:])
    corrupt("bad_magic", b"VDBFLAT2" + good[8:])


def test_pending_two_phase_scan_blocks_other_searches(hip_lib, oracle_mod):
    """A scan pending in a slot owns the handle's workspaces: other searches, a second scan into the same slot and
    a timing-ring resize say VDB_ERR_STATE instead of silently overwriting it; the finish then returns the right answer."""
    import torch
    n, dim = 5000, 768
    spec = SynthSpec(seed=3, kind=0)
    ix = FlatIndex(dim, "f32")
    ix.add_synth(0, n, spec)
    q = torch.from_numpy(oracle_mod.synth_rows(10, 2, dim, spec)).cuda()
    oi = torch.empty((2, 5), dtype=torch.int64, device="cuda")
    od = torch.empty((2, 5), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    ref_i, ref_d = ix.search(q.cpu().numpy(), 5)
    assert ix.search_scan_device(q.data_ptr(), 2, 5, 0, 0, st)
    for call in (lambda: ix.search(q.cpu().numpy(), 5),
                 lambda: ix.search_device(q.data_ptr(), 2, 5, 0, oi.data_ptr(), od.data_ptr(), st),
                 lambda: ix.selfjoin(3, q_n=4),
                 lambda: ix.set_option("timing_ring", 8),
                 lambda: ix._chk(ix._lib.vdb_search_scan_dev(ix._h, q.data_ptr(), 2, 5, 0, 0, None))):
        with pytest.raises(VdbError) as e:
            call()
        assert e.value.code == -5
    ix.search_finish_device(0, oi.data_ptr(), od.data_ptr(), st)
    torch.cuda.synchronize()
    assert np.array_equal(oi.cpu().numpy(), ref_i) and np.array_equal(od.cpu().numpy(), ref_d)
    ix.search(q.cpu().numpy(), 5)          # and the handle is usable again
    ix.close()


def test_search_fetch_returns_the_hits_rows_in_one_call(hip_lib, oracle_mod, corpus33k):
    """vdb_search_fetch == vdb_search + vdb_fetch: same ids and distances, rows bit-identical to the stored rows; zero
    rows behind -1 padding; fp32 and bf16+rerank stores; a sharded handle (rows come from their shards)."""
    n, dim, spec, rows, queries = corpus33k
    for dtype, kp, devices in (("f32", 0, None), ("bf16", 40, None), ("bf16", 40, [0, 0, 0])):
        ix = FlatIndex(dim, dtype, True, devices=devices)
        if devices:
            ix.reserve(n)
        ix.add_synth(0, n, spec)
        for b, k in ((1, 30), (5, 10)):
            i0, d0 = ix.search(queries[:b], k, kp)
            i1, d1, r1 = ix.search_fetch(queries[:b], k, kp)
            assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
            assert np.array_equal(r1.reshape(b * k, dim), ix.fetch(i1.reshape(-1)))
            assert np.max(np.abs(r1.reshape(b * k, dim) - rows[i1.reshape(-1)])) < 1e-6
        ix.close()
    tiny = FlatIndex(64, "f32")
    rng = np.random.default_rng(0)
    tiny.add(rng.standard_normal((3, 64)).astype(np.float32))
    i, d, r = tiny.search_fetch(rng.standard_normal((2, 64)).astype(np.float32), 5)
    assert np.all(i[:, 3:] == -1) and np.all(r[:, 3:] == 0) and np.all(np.abs(np.linalg.norm(r[:, :3], axis=2) - 1) < 1e-5)
    tiny.close()


def test_near_duplicate_clusters_show_the_limit_of_bf16_candidates(hip_lib, oracle_mod):
    """37 directions, ~220 rows each, 1e-3 of noise apart: in fp32 the rows of a cluster differ by ~1e-5 in score, in bf16
    not at all, so WHICH k' of a cluster the candidate pass keeps is arbitrary (found by tests/fuzz_gpu.py; include/vdb.h
    states the limit).  What must hold: returned rows carry their exact fp32 distances in order without repeats, the k-th
    distance is within bf16 resolution of the true one — on every kernel family — and an fp32 store (or k' = the whole
    cluster) returns the exact answer."""
    rng = np.random.default_rng(99)
    n, dim, b, k = 8191, 768, 65, 10
    base = rng.standard_normal((37, dim)).astype(np.float32)
    rows = base[rng.integers(0, 37, n)] + 1e-3 * rng.standard_normal((n, dim)).astype(np.float32)
    q = rows[rng.integers(0, n, b)] + 0.3 * rng.standard_normal((b, dim)).astype(np.float32)
    rows_n = oracle_mod.normalize_rows(rows)
    qn = _qn(oracle_mod, q)
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows_n, q, k)
    ix = FlatIndex(dim, "bf16", True)
    ix.set_option("mfma_cost_model", 0)
    ix.add(rows)
    for nb, variant in ((b, 4), (b, 5), (33, 6), (1, 6)):          # 128-wide tile, wide tile, streaming kernel, GEMV scan
        ix.set_option("mfma_variant", variant)
        idx, dist = ix.search(q[:nb], k, 40)
        for j in range(nb):
            got = idx[j]
            assert len(set(got.tolist())) == k and got.min() >= 0
            true_d = 1.0 - rows_n[got].astype(np.float64) @ qn[j].astype(np.float64)
            assert np.all(np.abs(true_d - dist[j]) <= 2e-6) and np.all(np.diff(dist[j].astype(np.float64)) >= -2e-6)
            assert dist[j, -1] <= r_dist[j, -1] + 8e-3
    ix.set_option("mfma_variant", 6)
    idx, dist = ix.search(q, k, 1024)                               # k' larger than any cluster: exact again
    check_topk(idx, dist, r_idx, r_dist, r_sc, rows_n, qn, what="near-duplicates k'=1024")
    ix.close()
    fx = FlatIndex(dim, "f32")
    fx.add(rows)
    idx, dist = fx.search(q, k)
    check_topk(idx, dist, r_idx, r_dist, r_sc, rows_n, qn, what="near-duplicates f32 store")
    fx.close()


def test_batched_search_on_a_store_without_the_fp32_copy(hip_lib, oracle_mod, corpus33k):
    """2 bytes per element (no fp32 copy): rerank_kprime re-scores the candidates against the stored bf16 rows with the
    fp32 query, so every kernel family returns the scan's own answer — the exact top-k of <fp32 query, bf16 row> —
    and batches no longer cost ceil(b/4) corpus passes."""
    n, dim, spec, rows, _ = corpus33k
    qspec = SynthSpec(seed=78, kind=2, noise=0.5, base_kind=1, base_seed=4321, base_n=n, base_n_clusters=181, base_noise=0.3)
    queries = oracle_mod.synth_rows(0, 300, dim, qspec)
    src = oracle_mod.round_bf16(rows)
    qn = _qn(oracle_mod, queries)
    r_idx, r_dist, r_sc = oracle_mod.exact_topk(src, queries, 10)
    ix = FlatIndex(dim, "bf16", False)
    ix.set_option("mfma_cost_model", 0)
    ix.add_synth(0, n, spec)
    i0, d0 = ix.search(queries, 10, 0)                       # the scan: 4 queries per corpus pass
    assert ix.get_stat("last_path") == 0
    check_topk(i0, d0, r_idx, r_dist, r_sc, src, qn, what="bf16-only scan")
    for b, variant, path in ((300, 5, 5), (129, 4, 4), (64, 6, 6), (3, 6, 6), (1, 6, 0)):
        ix.set_option("mfma_variant", variant)
        idx, dist = ix.search(queries[:b], 10, 40)
        assert ix.get_stat("last_path") == path and ix.get_stat("mfma_last_overflow") == 0
        check_topk(idx, dist, r_idx[:b], r_dist[:b], r_sc[:b], src, qn[:b], what=f"bf16-only batched b={b} variant {variant}")
        assert np.allclose(dist, d0[:b], atol=1e-6)
    sj_i, sj_d = ix.selfjoin(5, exclude_self=True, rerank_kprime=40, q_row0=100, q_n=70)
    assert not np.any(sj_i == (100 + np.arange(70))[:, None]) and np.all(np.diff(sj_d, axis=1) >= 0)
    # k' > 64: the generic finish (merge_lists_kernel + rerank_kernel) re-scores against the bf16 rows as well
    r30 = oracle_mod.exact_topk(src, queries[:5], 30)
    i30, d30 = ix.search(queries[:5], 30, 200)
    assert ix.get_stat("last_path") == 0
    check_topk(i30, d30, r30[0], r30[1], r30[2], src, qn[:5], what="bf16-only k=30 k'=200")
    ix.close()
    sh = FlatIndex(dim, "bf16", False, devices=[0, 0, 0])
    sh.set_option("mfma_cost_model", 0)
    sh.reserve(n)
    sh.add_synth(0, n, spec)
    idx, dist = sh.search(queries[:64], 10, 40)
    check_topk(idx, dist, r_idx[:64], r_dist[:64], r_sc[:64], src, qn[:64], what="bf16-only sharded batched")
    sh.close()


def test_add_path_kernels_against_the_oracle(hip_lib, oracle_mod):
    """The add path's two normalisation kernels (rows in registers with 16-byte accesses for dims that are a multiple of 4 up
    to 2048 on aligned input; the scalar kernel otherwise): the stored fp32 rows are the oracle's within 1 ulp, the stored
    bf16 rows its rounding of them, whichever kernel ran — host rows, device rows, and a device pointer that is only
    4-byte aligned."""
    import torch
    rng = np.random.default_rng(12)
    for dim in (4, 100, 384, 768, 1024, 2048, 2052, 770, 7):
        n = 1000
        rows = (rng.standard_normal((n, dim)) * rng.uniform(0.01, 100.0, (n, 1))).astype(np.float32)
        rows[17] = 0.0                                             # a zero row stays zero (1 / (0 + 1e-30) * 0)
        ref = oracle_mod.normalize_rows(rows)
        ulp = lambda a, b: np.max(np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)))   # noqa: E731
        for how in ("host", "device", "device_unaligned"):
            ix = FlatIndex(dim, "f32")
            bx = FlatIndex(dim, "bf16", False)
            if how == "host":
                ix.add(rows); bx.add(rows)
            else:
                flat = torch.empty(n * dim + 1, dtype=torch.float32, device="cuda")
                off = 1 if how == "device_unaligned" else 0
                t = flat[off:off + n * dim].view(n, dim)
                t.copy_(torch.from_numpy(rows))
                torch.cuda.synchronize()
                assert (t.data_ptr() % 16 == 0) == (off == 0)
                ix.add_device(t.data_ptr(), n); bx.add_device(t.data_ptr(), n)
            got = ix.fetch(np.arange(n))
            assert ulp(got, ref) <= 1 and np.all(got[17] == 0), (dim, how)
            got_b = bx.fetch(np.arange(n))
            ref_b = oracle_mod.round_bf16(got)                    # bf16 of what the fp32 path stored
            assert np.array_equal(got_b, ref_b), (dim, how)
            ix.close(); bx.close()
