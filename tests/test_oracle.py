"""CPU suite: the C oracle against the committed golden vectors and against an
independent numpy statement; generator determinism; ABI-on-CPU behaviour."""
import os

import numpy as np
import pytest

from vectordb_amd import FlatIndex, SynthSpec

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _spec(g, prefix):
    return SynthSpec(**{f: g[f"{prefix}_{f}"].item() for f, _ in SynthSpec._fields_})


@pytest.mark.parametrize("name", ["iso", "clustered"])
def test_oracle_matches_golden_synth(oracle_mod, name):
    g = np.load(os.path.join(GOLD, f"synth_{name}_n4096.npz"))
    n, dim, b = int(g["n"]), int(g["dim"]), int(g["b"])
    spec, qspec = _spec(g, "spec"), _spec(g, "qspec")
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    queries = oracle_mod.synth_rows(0, b, dim, qspec)
    # the generator reproduces the exact bits the fixture was made from
    assert np.array_equal(rows[0, :8], g["row0_head"]) and np.array_equal(queries[0, :8], g["query0_head"])
    assert rows.astype(np.float64).sum() == g["rows_checksum"]
    for k in (1, 10, 30, 100):
        idx, dist, sc = oracle_mod.exact_topk(rows, queries, k)
        assert np.array_equal(idx, g[f"idx_k{k}"])
        assert np.array_equal(dist, g[f"dist_k{k}"])
        assert np.allclose(sc, g[f"score_k{k}"], rtol=0, atol=1e-14)
    # planted neighbours: a kind-2 query's source row is its nearest neighbour
    for q in range(b):
        assert g["idx_k1"][q, 0] == oracle_mod.synth_query_src(qspec, q)


def test_oracle_matches_golden_adversarial(oracle_mod):
    g = np.load(os.path.join(GOLD, "adversarial_d64.npz"))
    rows_n = oracle_mod.normalize_rows(g["rows"])
    for k in (1, 5, 31, 40):
        idx, dist, _ = oracle_mod.exact_topk(rows_n, g["queries"], k)
        assert np.array_equal(idx, g[f"idx_k{k}"]), k
        assert np.array_equal(dist, g[f"dist_k{k}"]), k
    # exact duplicates tie and resolve to the lower row index
    idx, _, _ = oracle_mod.exact_topk(rows_n, g["rows"][:1], 2)
    assert idx[0].tolist() == [0, 20]


def test_generator_is_counter_based(oracle_mod):
    spec = SynthSpec(seed=77, kind=1, n_clusters=9, noise=0.3)
    whole = oracle_mod.synth_rows(0, 100, 48, spec)
    part = oracle_mod.synth_rows(37, 20, 48, spec)
    assert np.array_equal(whole[37:57], part)
    assert whole.min() >= -1.3 and whole.max() <= 1.3 and abs(float(whole.mean())) < 0.05


def test_normalize_and_bf16(oracle_mod):
    x = np.random.default_rng(0).standard_normal((7, 33)).astype(np.float32)
    x[3] = 0
    y = oracle_mod.normalize_rows(x)
    assert np.allclose(np.linalg.norm(y[[0, 1, 2, 4, 5, 6]].astype(np.float64), axis=1), 1, atol=1e-6)
    assert np.all(y[3] == 0)
    r = oracle_mod.round_bf16(np.array([1.0, 1.00390625, 1.005859375, -3.1415927, np.nan], np.float32))
    assert r[0] == 1.0 and r[1] == 1.0 and r[2] == 1.0078125 and np.isnan(r[4])
    assert (r[:4].view(np.uint32) & 0xFFFF == 0).all()


def test_cpu_abi_store_roundtrip(oracle_lib, oracle_mod, tmp_path):
    """The vdb_* ABI on the CPU oracle: add/search/fetch/save/load/self-join/merge."""
    rng = np.random.default_rng(1)
    rows = rng.standard_normal((300, 40)).astype(np.float32)
    ix = FlatIndex(40, "f32", lib=oracle_lib)
    ix.add(rows[:100]); ix.add(rows[100:])
    assert ix.count() == 300
    q = rows[[5, 17]] + 0.01
    idx, dist = ix.search(q, 7)
    ref_idx, ref_dist, _ = oracle_mod.exact_topk(oracle_mod.normalize_rows(rows), q, 7)
    assert np.array_equal(idx, ref_idx) and np.array_equal(dist, ref_dist)
    assert np.array_equal(ix.fetch([3, 299]), oracle_mod.normalize_rows(rows)[[3, 299]])
    ix.save(str(tmp_path))
    for dtype, kp in (("f32", 0), ("bf16", 28)):
        ix2 = FlatIndex.load(str(tmp_path), dtype, True, lib=oracle_lib)
        i2, d2 = ix2.search(q, 7, kp)
        assert np.array_equal(i2, ref_idx) and np.allclose(d2, ref_dist, atol=1e-6)
        ix2.close()
    sj_idx, sj_dist = ix.selfjoin(3, exclude_self=True)
    assert sj_idx.shape == (300, 3) and not np.any(sj_idx == np.arange(300)[:, None])
    sj2, _ = ix.selfjoin(1, exclude_self=False)
    assert np.array_equal(sj2[:, 0], np.arange(300))
    # where= pre-filter: the masked search is the exact top-k of the eligible rows
    mask = rng.random(300) < 0.4
    allowed = np.flatnonzero(mask)
    mi, md = ix.search_masked(q, 7, mask)
    s_idx, s_dist, _ = oracle_mod.exact_topk(oracle_mod.normalize_rows(rows)[allowed], q, 7)
    assert np.array_equal(mi, allowed[s_idx]) and np.array_equal(md, s_dist)
    few = np.zeros(300, bool); few[[9, 200]] = True
    mi, md = ix.search_masked(q, 7, few)
    assert sorted(mi[0, :2].tolist()) == [9, 200] and np.all(mi[:, 2:] == -1) and np.all(np.isinf(md[:, 2:]))
    with pytest.raises(Exception):
        ix.search_masked(q, 7, np.ones(299, bool))
    # two-phase form (phase 1 scan, phase 2 finish): same answer as the one-call search, per slot
    qc = np.ascontiguousarray(q, np.float32)
    for slot in (0, 1):
        ti = np.empty((2, 7), np.int64); td = np.empty((2, 7), np.float32)
        assert ix.search_scan_device(qc.ctypes.data, 2, 7, 0, slot)
        ix.search_finish_device(slot, ti.ctypes.data, td.ctypes.data)
        assert np.array_equal(ti, idx) and np.array_equal(td, dist)
        with pytest.raises(Exception):
            ix.search_finish_device(slot, ti.ctypes.data, td.ctypes.data)      # nothing pending any more
    assert not ix.search_scan_device(np.zeros((65, 40), np.float32).ctypes.data, 65, 7, 0, 0)   # more than one scan round: does not split
    with pytest.raises(Exception):
        ix.search(q, 0)
    with pytest.raises(Exception):
        ix.search(q, 5, 10)  # rerank on an f32 store
    ix.close()


def test_query_source_row_host_mirror(oracle_mod):
    """vectordb_amd.synth.query_source_row (pure Python, used by bench.py's sanity check) ==
    the generator's own planted-neighbour index."""
    from vectordb_amd import SynthSpec
    from vectordb_amd.synth import query_source_row
    q = SynthSpec(seed=1235, kind=2, noise=0.5, base_kind=1, base_seed=1234, base_n=10_000_000, base_n_clusters=3162, base_noise=0.3)
    for i in list(range(64)) + [10 ** 9 + 7, 2 ** 40 + 3]:
        assert query_source_row(q, i) == oracle_mod.synth_query_src(q, i)
