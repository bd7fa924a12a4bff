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
    for fuse in (2, 0, 2):      # 2 = fused wherever the shape allows (1, the default, fuses 2..4 queries on large stores only); the third call re-uses the arrival counter the first one reset
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
    ix.set_option("batch_mfma_min_b", 1 << 20)   # keep 2..4 queries on the scan (the cost model would send a bf16 + rerank batch to the streaming kernel)
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
    one.set_option("fuse_small_batch", 2)       # two and four queries in one launch too (the default fuses them on large stores only)
    sh.set_option("fuse_small_batch", 2)
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


# ---- the rest of the ABI on a sharded handle (round 3): where= bitmaps, self-join, device-resident appends,
# ---- re-partitioning; and the paths a second physical GPU would take, forced on the one GPU of the test box ----

def _pair(dim, dtype, g, n, spec, reserve=None):
    one = FlatIndex(dim, dtype, True)
    one.add_synth(0, n, spec)
    sh = FlatIndex(dim, dtype, True, devices=[0] * g)
    sh.reserve(n if reserve is None else reserve)
    sh.add_synth(0, n, spec)
    return one, sh


def _same(a, b, what):
    assert np.array_equal(a[0], b[0]), f"{what}: ids differ"
    assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32)), f"{what}: distances differ in their bits"


@pytest.mark.parametrize("g", [2, 3, 8])
@pytest.mark.parametrize("dtype,kp", [("f32", 0), ("bf16", 40)])
def test_sharded_handle_serves_masks_device_appends_and_rebalance(hip_lib, oracle_mod, g, dtype, kp):
    import torch
    dim, n = 768, 30_011
    spec = SynthSpec(seed=41, kind=1, n_clusters=173, noise=0.3)
    qspec = SynthSpec(seed=42, kind=2, noise=0.5, base_kind=1, base_seed=41, base_n=n, base_n_clusters=173, base_noise=0.3)
    queries = oracle_mod.synth_rows(0, 9, dim, qspec)
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    one, sh = _pair(dim, dtype, g, n, spec)
    for remote in (0, 1):                                   # 1: every shard but the first as if on a GPU without peer access
        sh.set_option("shard_force_remote", remote)
        for b in (1, 4, 9):
            _same(sh.search(queries[:b], 10, kp), one.search(queries[:b], 10, kp), f"search g={g} b={b} remote={remote}")
        # where=: a global bitmap, each shard takes its word-aligned slice (host and device forms)
        rng = np.random.default_rng(g)
        for frac in (0.5, 0.01):
            allow = rng.random(n) < frac
            got = sh.search_masked(queries[:5], 10, allow, kp)
            _same(got, one.search_masked(queries[:5], 10, allow, kp), f"masked g={g} {frac} remote={remote}")
            assert np.all(allow[got[0][got[0] >= 0]])
            sub = np.flatnonzero(allow)
            r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows[sub], queries[:5], 10)
            r_idx = np.where(r_idx >= 0, sub[np.maximum(r_idx, 0)], -1)
            check_topk(got[0], got[1], r_idx, r_dist, r_sc, rows, _qn(oracle_mod, queries[:5]), what=f"masked g={g} {frac}")
            bits = np.zeros(((n + 31) // 32) * 4, np.uint8)
            pk = np.packbits(allow, bitorder="little")
            bits[: pk.size] = pk
            d_bits = torch.from_numpy(bits.view(np.int32).copy()).cuda()
            q_t = torch.from_numpy(queries[:5]).cuda()
            oi = torch.empty((5, 10), dtype=torch.int64, device="cuda")
            od = torch.empty((5, 10), dtype=torch.float32, device="cuda")
            sh.search_masked_device(q_t.data_ptr(), 5, 10, kp, d_bits.data_ptr(), n, oi.data_ptr(), od.data_ptr(), torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            _same((oi.cpu().numpy(), od.cpu().numpy()), got, f"masked_dev g={g} {frac} remote={remote}")
    # device-resident appends routed by span (rows on shard 0's device), straddling shard borders and the reserved total
    extra = np.random.default_rng(9).standard_normal((2_000, dim)).astype(np.float32)
    t = torch.from_numpy(extra).cuda()
    torch.cuda.synchronize()
    one.add_device(t.data_ptr(), 2_000)
    sh.add_device(t.data_ptr(), 2_000)
    assert sh.count() == one.count() == n + 2_000
    q2 = extra[[5, 1_999]]
    _same(sh.search(q2, 7, kp), one.search(q2, 7, kp), f"after add_dev g={g}")
    assert sh.search(q2, 1, kp)[0][:, 0].tolist() == [n + 5, n + 1_999]
    # re-partition: the rows beyond the reserved total sat in the last shard; now the shards are even again
    per0 = sh.get_stat("rows_per_shard")
    sh.rebalance(2 * (n + 2_000))
    assert sh.get_stat("rows_per_shard") > per0 and sh.count() == n + 2_000
    _same(sh.search(queries, 10, kp), one.search(queries, 10, kp), f"after rebalance g={g}")
    per1 = sh.get_stat("rows_per_shard")
    probe = np.array([0, per0 - 1, per0, min(per1, n) - 1, min(per1, n), n, n + 1_999])     # rows either side of the old and new shard borders
    assert np.array_equal(sh.fetch(probe), one.fetch(probe))
    sh.add_device(t.data_ptr(), 100)                        # and it keeps growing
    one.add_device(t.data_ptr(), 100)
    _same(sh.search(queries[:3], 10, kp), one.search(queries[:3], 10, kp), f"append after rebalance g={g}")
    sh.rebalance(1000)                                      # a smaller total than what is stored: partitioned for the stored rows
    _same(sh.search(queries[:3], 10, kp), one.search(queries[:3], 10, kp), f"rebalance down g={g}")
    one.close(); sh.close()


@pytest.mark.parametrize("g", [2, 5])
def test_sharded_selfjoin_equals_single_store(hip_lib, oracle_mod, g):
    """vdb_selfjoin on a sharded handle: blocks of rows-as-queries cut at shard borders, every shard scanned, merged —
    the single store's answer bit for bit (fp32 store and bf16 + rerank), on a sub-range too."""
    dim, n, k = 768, 23_456, 10
    spec = SynthSpec(seed=61, kind=1, n_clusters=153, noise=0.3)
    for dtype, kp in (("f32", 0), ("bf16", 40)):
        one, sh = _pair(dim, dtype, g, n, spec)
        for remote in (0, 1):
            sh.set_option("shard_force_remote", remote)
            a = sh.selfjoin(k, exclude_self=True, rerank_kprime=kp)
            b = one.selfjoin(k, exclude_self=True, rerank_kprime=kp)
            assert not np.any(a[0] == np.arange(n)[:, None])
            _same(a, b, f"selfjoin g={g} {dtype} remote={remote}")
        _same(sh.selfjoin(3, exclude_self=False, rerank_kprime=kp, q_row0=4_000, q_n=9_000),
              one.selfjoin(3, exclude_self=False, rerank_kprime=kp, q_row0=4_000, q_n=9_000), f"selfjoin range g={g} {dtype}")
        one.close(); sh.close()


def test_warm_presizes_every_workspace(hip_lib, oracle_mod):
    dim, n = 768, 60_000
    spec = SynthSpec(seed=81, kind=1, n_clusters=244, noise=0.3)
    queries = oracle_mod.synth_rows(0, 300, dim, SynthSpec(seed=82, kind=0))
    for devices in (None, [0, 0, 0]):
        ix = FlatIndex(dim, "bf16", True, devices=devices)
        ix.reserve(n)
        ix.add_synth(0, n, spec)
        ix.warm(300, 10, 40)
        before = ix.get_stat("workspace_allocs")
        import torch
        q_t = torch.from_numpy(queries).cuda()
        oi = torch.empty((300, 10), dtype=torch.int64, device="cuda")
        od = torch.empty((300, 10), dtype=torch.float32, device="cuda")
        for b in (1, 3, 4, 17, 64, 100, 128, 255, 300):
            ix.search_device(q_t.data_ptr(), b, 10, 40, oi.data_ptr(), od.data_ptr(), torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert ix.get_stat("workspace_allocs") == before, "a search after vdb_warm grew a workspace"
        ix.close()


def test_collection_on_a_sharded_segment(hip_lib, oracle_mod, tmp_path):
    """PersistentClient(path, devices=[...]): the reference's calls (add in batches, query with embeddings, where=, get,
    reopen: src/vectordb.py:149,156-160,215-219,230,259-264) on a row-sharded segment give the single-GPU collection's answers."""
    from vectordb_amd.chroma_compat import PersistentClient
    dim, n = 768, 12_345
    spec = SynthSpec(seed=91, kind=1, n_clusters=111, noise=0.3)
    raw = oracle_mod.synth_rows(0, n, dim, spec)
    queries = oracle_mod.synth_rows(0, 6, dim, SynthSpec(seed=92, kind=2, noise=0.4, base_kind=1, base_seed=91, base_n=n, base_n_clusters=111, base_noise=0.3))
    metas = [{"product_code": str(i), "usda_code": "A" if i % 5 == 0 else "B"} for i in range(n)]
    ids = [f"item_{i}" for i in range(n)]
    cols = {}
    for name, kw in (("one", {}), ("sharded", {"devices": [0, 0, 0, 0]})):
        client = PersistentClient(path=str(tmp_path / name), dtype="bf16", **kw)
        col = client.get_or_create_collection(name="actual_products", metadata={"hnsw:space": "cosine"})
        for lo in range(0, n, 5000):
            col.add(embeddings=raw[lo:lo + 5000], metadatas=metas[lo:lo + 5000], ids=ids[lo:lo + 5000])
        cols[name] = (client, col)
    assert cols["sharded"][1]._index.get_stat("n_shards") == 4 and cols["sharded"][1].count() == n

    def answers(col):
        out = []
        for q in range(6):
            r = col.query(query_embeddings=[queries[q].tolist()], n_results=30, include=["metadatas", "distances", "embeddings"])
            out.append((r["ids"][0], r["distances"][0], np.asarray(r["embeddings"][0])))
        r = col.query(query_embeddings=queries[:3], n_results=5, where={"usda_code": "A"}, include=["distances"])
        out.append((r["ids"], r["distances"]))
        g = col.get(ids=["item_7", "item_12344", "nope"], include=["embeddings", "metadatas"])
        out.append((g["ids"], np.asarray(g["embeddings"]), g["metadatas"]))
        return out

    def plain(x):
        if isinstance(x, np.ndarray):
            return x.tolist()
        if isinstance(x, (list, tuple)):
            return [plain(y) for y in x]
        return x

    a, b = answers(cols["one"][1]), answers(cols["sharded"][1])
    assert plain(a) == plain(b)
    assert all(i.endswith(("0", "5")) for row in a[6][0] for i in row)
    cols["sharded"][0].close()                                  # writes vectors.vdb; reopen row-sharded over another shard count
    again = PersistentClient(path=str(tmp_path / "sharded"), dtype="bf16", devices=[0, 0]).get_collection("actual_products")
    assert again.count() == n and again._index.get_stat("n_shards") == 2
    assert plain(answers(again)) == plain(a)
    again.add(embeddings=raw[:10] * 2.0, ids=[f"more_{i}" for i in range(10)])     # grows past the reopened partition
    assert again.count() == n + 10 and again.query(query_embeddings=[raw[3].tolist()], n_results=2)["ids"][0] == ["item_3", "more_3"]
    cols["one"][0].close()


def test_rccl_exchange_behind_the_c_abi(hip_lib, oracle_mod):
    """Option shard_exchange_rccl: the shards' results are exchanged by ONE RCCL all-gather (loaded on demand) and merged
    on shard 0 — the exchange north_star names — instead of the P2P stores.  One rank per GPU, so on the one-GPU box it
    runs with a single shard (the collective degenerates to a copy; the packed layout, the strides of the merge and the
    on-demand loading are what is exercised), and a handle with two shards on one device refuses it.  tests/test_zy_physical_gpus.py
    compares the two exchanges over the physical GPUs of a box that has several."""
    from vectordb_amd._capi import VdbError
    dim, n = 768, 30_000
    spec = SynthSpec(seed=95, kind=1, n_clusters=173, noise=0.3)
    queries = oracle_mod.synth_rows(0, 70, dim, SynthSpec(seed=96, kind=2, noise=0.5, base_kind=1, base_seed=95, base_n=n, base_n_clusters=173, base_noise=0.3))
    one, sh = _pair(dim, "bf16", 1, n, spec)
    assert sh.get_stat("shard_exchange_rccl") == 0
    sh.set_option("shard_exchange_rccl", 1)
    assert sh.get_stat("shard_exchange_rccl") == 1
    for b, k in ((1, 10), (1, 3), (4, 10), (33, 5), (70, 10)):          # odd b * k: the packed block is padded to 8 bytes
        _same(sh.search(queries[:b], k, 40), one.search(queries[:b], k, 40), f"rccl exchange b={b} k={k}")
    allow = np.arange(n) % 5 == 0
    _same(sh.search_masked(queries[:3], 7, allow, 40), one.search_masked(queries[:3], 7, allow, 40), "rccl exchange, masked")
    sh.set_option("shard_exchange_rccl", 0)
    _same(sh.search(queries[:4], 10, 40), one.search(queries[:4], 10, 40), "back on the P2P exchange")
    one.close(); sh.close()
    two = FlatIndex(dim, "bf16", True, devices=[0, 0])
    with pytest.raises(VdbError) as e:
        two.set_option("shard_exchange_rccl", 1)
    assert e.value.code == -5 and "one rank per GPU" in str(e.value)
    two.close()


@pytest.mark.parametrize("dtype,dim,kp", [("f32", 768, 0), ("bf16", 768, 0), ("bf16", 384, 40), ("f32", 384, 0)])
def test_group_launch_equals_one_launch_per_group(hip_lib, oracle_mod, dtype, dim, kp):
    """A GEMV-path batch sends its whole groups of four queries as ONE 2-D launch (option scan_group_launch, the default):
    same lists in the same places as one launch per group, so the same bits; checked against the oracle too, with a
    where= bitmap, with the self-join's own-row exclusion, and across the 64-query chunk boundary."""
    n = 23_457
    spec = SynthSpec(seed=61, kind=1, n_clusters=151, noise=0.3)
    qspec = SynthSpec(seed=62, kind=2, noise=0.5, base_kind=1, base_seed=61, base_n=n, base_n_clusters=151, base_noise=0.3)
    rows = oracle_mod.normalize_rows(oracle_mod.synth_rows(0, n, dim, spec))
    queries = oracle_mod.synth_rows(0, 135, dim, qspec)
    qn = _qn(oracle_mod, queries)
    ix = FlatIndex(dim, dtype, True)
    ix.add_synth(0, n, spec)
    ix.set_option("batch_mfma_min_b", 1 << 20)      # the scan, whatever the batch
    rng = np.random.default_rng(7)
    allow = rng.random(n) < 0.3
    for b in (5, 8, 11, 30, 64, 67, 135):
        for k in (10, 100):                          # 100: two list entries per lane
            got = {}
            for mode in (1, 0):
                ix.set_option("scan_group_launch", mode)
                got[mode] = (ix.search(queries[:b], k, kp if kp >= k else 0),
                             ix.search_masked(queries[:b], k, allow, kp if kp >= k else 0))
                assert ix.get_stat("last_corpus_passes") == (b // 4) + (1 if b % 4 >= 2 else 0) + (b % 2)
            for (i1, d1), (i0, d0) in zip(got[1], got[0]):
                assert np.array_equal(i1, i0) and np.array_equal(d1.view(np.uint32), d0.view(np.uint32)), f"{dtype} b={b} k={k}"
            if dtype == "f32" or kp >= 4 * k:
                r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows, queries[:b], k)
                check_topk(got[1][0][0], got[1][0][1], r_idx, r_dist, r_sc, rows, qn[:b], what=f"group launch {dtype} {dim} b={b} k={k}")
                sub = np.flatnonzero(allow)
                m_idx, m_dist, m_sc = oracle_mod.exact_topk(rows[sub], queries[:b], k)
                check_topk(got[1][1][0], got[1][1][1], sub[m_idx], m_dist, m_sc, rows, qn[:b], what=f"group launch masked {dtype} {dim} b={b} k={k}")
    # the self-join hands each query of a group its own row to skip
    out = {}
    for mode in (1, 0):
        ix.set_option("scan_group_launch", mode)
        out[mode] = ix.selfjoin(5, True, 0, 0, 1000)
    assert np.array_equal(out[1][0], out[0][0]) and np.array_equal(out[1][1].view(np.uint32), out[0][1].view(np.uint32))
    assert not (out[1][0] == np.arange(1000)[:, None]).any()
    ix.set_option("scan_group_launch", 1)
    ix.close()


def test_allocation_order_is_only_a_placement(hip_lib, oracle_mod):
    """alloc_f32_first swaps the order in which a bf16 store's two arrays are allocated (the scanned rows go first by
    default: heap placement changes a streaming read's rate, profiles/r03_placement.txt) — growth, answers and the
    stored rows are the same either way."""
    n, dim = 9_000, 768
    spec = SynthSpec(seed=71, kind=1, n_clusters=90, noise=0.3)
    rows = oracle_mod.synth_rows(0, n, dim, spec)
    queries = rows[:5] + 0.05
    got = []
    for first in (0, 1):
        ix = FlatIndex(dim, "bf16", True)
        ix.set_option("alloc_f32_first", first)
        for lo in range(0, n, 2_777):                 # several growth steps (copy into the new arrays, free the old)
            ix.add(rows[lo:lo + 2_777])
        got.append((ix.search(queries, 10, 40), ix.search(queries[:1], 10, 40), ix.fetch(np.arange(0, n, 997))))
        ix.close()
    for a, b in zip(got[0], got[1]):
        if isinstance(a, tuple):
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        else:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    r_idx, _, _ = oracle_mod.exact_topk(oracle_mod.normalize_rows(rows), queries, 10)
    assert np.array_equal(got[0][0][0][:, 0], r_idx[:, 0])


@pytest.mark.parametrize("dtype,dim", [("bf16", 768), ("f32", 768), ("bf16", 384)])
def test_super_round_partition_covers_every_tile_once(hip_lib, oracle_mod, dtype, dim):
    """The fused scan's block groups take slices of super-rounds (TileWalk, scan_gemv.h).  Whatever the size of a round
    and the grid — whole rounds, the split remainder, a store shorter than one round — every tile is scored exactly
    once: same bits as one range per group (scan_super_rounds = 0), as the unfused kernels, and the oracle's answers.
    A planted copy of the query in EVERY 997th row must come back complete: a tile left out or read twice shows."""
    n = 70_001
    spec = SynthSpec(seed=81, kind=1, n_clusters=263, noise=0.3)
    rows = oracle_mod.synth_rows(0, n, dim, spec)
    rng = np.random.default_rng(5)
    q = rng.standard_normal((4, dim)).astype(np.float32)
    planted = np.arange(3, n, 997)
    rows[planted] = q[0] * 3.0                           # 71 exact copies of query 0 (up to scale), spread over the store
    rows_n = oracle_mod.normalize_rows(rows)
    ix = FlatIndex(dim, dtype, True)
    ix.add(rows)
    ix.set_option("batch_mfma_min_b", 1 << 20)
    ix.set_option("fuse_small_batch", 2)
    kp = 64 if dtype == "bf16" else 0
    ref = None
    for bpc in (2, 1, 3):
        ix.set_option("scan_blocks_per_cu", bpc)
        for c in (0, 2, 3, 8, 64):
            ix.set_option("scan_super_rounds", c)
            for rep in range(3):                          # the shares tune themselves between launches: three partitions per setting
                got = [ix.search(q[:b], 64, kp) for b in (1, 2, 4)]
                if ref is None:
                    ref = got
                for (gi, gd), (ri, rd) in zip(got, ref):
                    assert np.array_equal(gi, ri) and np.array_equal(gd.view(np.uint32), rd.view(np.uint32)), f"{dtype} bpc={bpc} c={c} rep={rep}"
    assert ref[0][0][0].tolist() == planted[:64].tolist()      # 71 equal scores: the 64 lowest row indices, in order
    ix.set_option("fuse_small_batch", 0)
    un = ix.search(q, 64, kp)
    assert np.array_equal(un[0], ref[2][0]) and np.array_equal(un[1].view(np.uint32), ref[2][1].view(np.uint32))
    if dtype == "f32":                                         # (bf16 candidates with k' = k have no head-room for a strict check)
        r_idx, r_dist, r_sc = oracle_mod.exact_topk(rows_n, q, 64)
        check_topk(un[0][1:], un[1][1:], r_idx[1:], r_dist[1:], r_sc[1:], rows_n, oracle_mod.normalize_rows(q)[1:], what=f"super rounds {dtype} {dim}")
    ix.close()
