"""CPU suite, round 2: incremental persistence of the Chroma face, store-file validation, the sharded handle's
calling rules (on the oracle's ABI twin — no GPU here), the (None, -1.0) mapping of empty matches and bench.py's
self-launch plumbing."""
import json
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from vectordb_amd import FlatIndex
from vectordb_amd._capi import VdbError
from vectordb_amd.chroma_compat import PersistentClient

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _col(client, name="c"):
    return client.get_or_create_collection(name=name, metadata={"hnsw:space": "cosine"})


def test_adds_append_instead_of_rewriting(oracle_lib, tmp_path, monkeypatch):
    """N single-row adds: records and embeddings files only ever grow by the rows added, and the engine's store file
    is not written per add at all (it is a reopen cache: persist()/close() write it once)."""
    saves = []
    real_save = FlatIndex.save
    monkeypatch.setattr(FlatIndex, "save", lambda self, d: (saves.append(self.count()), real_save(self, d))[1])
    client = PersistentClient(path=str(tmp_path), lib=oracle_lib)
    col = _col(client)
    rng = np.random.default_rng(0)
    emb = rng.standard_normal((40, 16)).astype(np.float32)
    d = tmp_path / "c"
    sizes = []
    for i in range(40):
        col.add(embeddings=emb[i:i + 1], metadatas=[{"i": i}], ids=[f"item_{i}"])
        sizes.append(((d / "records.jsonl").stat().st_size, (d / "embeddings.f32").stat().st_size))
    assert saves == []                                                       # no O(N) store rewrite per add
    assert [e for _, e in sizes] == [(i + 1) * 16 * 4 for i in range(40)]      # embeddings: exactly the rows added
    growth = np.diff([r for r, _ in sizes])
    assert growth.min() > 0 and growth.max() < 2 * growth.min()               # records: one line per add
    assert json.load(open(d / "collection.json"))["count"] == 40
    client.close()
    assert saves == [40]                                                     # written once, on close
    # reopen from the cache file, and again without it (rebuilt from embeddings.f32): same answers
    q = emb[7] + 0.01
    a = _col(PersistentClient(path=str(tmp_path), lib=oracle_lib)).query(query_embeddings=[q], n_results=5, include=["distances", "metadatas"])
    os.remove(d / "vectors.vdb")
    b = _col(PersistentClient(path=str(tmp_path), lib=oracle_lib)).query(query_embeddings=[q], n_results=5, include=["distances", "metadatas"])
    assert a["ids"] == b["ids"] and a["ids"][0][0] == "item_7" and a["distances"] == b["distances"]
    assert a["metadatas"][0][0] == {"i": 7}


def test_torn_append_is_ignored_and_repaired(oracle_lib, tmp_path):
    """A crash between the appends and the manifest: the committed count wins on reopen, and the next add cuts the
    torn tail off before appending."""
    client = PersistentClient(path=str(tmp_path), lib=oracle_lib)
    col = _col(client)
    rng = np.random.default_rng(1)
    emb = rng.standard_normal((10, 8)).astype(np.float32)
    col.add(embeddings=emb[:6], ids=[f"a{i}" for i in range(6)])
    client.close()
    d = tmp_path / "c"
    with open(d / "records.jsonl", "a") as f:                # torn: a record and half of another, embeddings for one
        f.write(json.dumps({"id": "ghost", "metadata": None, "document": None}) + "\n{\"id\": \"gho")
    with open(d / "embeddings.f32", "ab") as f:
        f.write(emb[9].tobytes())
    os.remove(d / "vectors.vdb")                             # and no cache: the rebuild must stop at the committed count
    client = PersistentClient(path=str(tmp_path), lib=oracle_lib)
    col = _col(client)
    assert col.count() == 6 and col.get(ids=["ghost"])["ids"] == []
    col.add(embeddings=emb[6:8], ids=["a6", "a7"])
    client.close()
    col = _col(PersistentClient(path=str(tmp_path), lib=oracle_lib))
    assert col.count() == 8 and col.get()["ids"] == [f"a{i}" for i in range(8)]
    res = col.query(query_embeddings=[emb[7]], n_results=1)
    assert res["ids"] == [["a7"]]
    assert (d / "embeddings.f32").stat().st_size == 8 * 8 * 4


def test_store_file_validation_on_the_oracle_twin(oracle_lib, tmp_path):
    rng = np.random.default_rng(2)
    ix = FlatIndex(32, "f32", lib=oracle_lib)
    ix.add(rng.standard_normal((100, 32)).astype(np.float32))
    d = tmp_path / "store"
    ix.save(str(d))
    ix.close()
    assert sorted(os.listdir(d)) == ["vectors.vdb"]
    good = (d / "vectors.vdb").read_bytes()
    FlatIndex.load(str(d), "f32", lib=oracle_lib).close()
    hdr = lambda dim, dt, n: b"VDBFLAT1" + struct.pack("<iiqq", dim, dt, n, 0)   # noqa: E731
    for name, data in {"truncated": good[:-40], "negative_n": hdr(32, 0, -5) + good[32:], "huge_n": hdr(32, 0, 1 << 40) + good[32:],
                       "bad_dim": hdr(0, 0, 100) + good[32:], "bad_dtype": hdr(32, 7, 100) + good[32:]}.items():
        dd = tmp_path / name
        dd.mkdir()
        (dd / "vectors.vdb").write_bytes(data)
        with pytest.raises(VdbError) as e:
            FlatIndex.load(str(dd), "f32", lib=oracle_lib)
        assert e.value.code == -4, name


def test_sharded_handle_rules_on_the_oracle_twin(oracle_lib, oracle_mod, tmp_path):
    """The Python face of vdb_create_sharded without a GPU: reserve-before-add, same answers as the plain store,
    one file format."""
    rng = np.random.default_rng(3)
    rows = rng.standard_normal((500, 24)).astype(np.float32)
    q = rows[[3, 400]] + 0.01
    plain = FlatIndex(24, "f32", lib=oracle_lib)
    plain.add(rows)
    sh = FlatIndex(24, "f32", lib=oracle_lib, devices=[0, 1, 2])
    assert sh.get_stat("n_shards") == 3
    with pytest.raises(VdbError) as e:
        sh.add(rows)
    assert e.value.code == -5
    sh.reserve(500)
    sh.add(rows[:123]); sh.add(rows[123:])
    i0, d0 = plain.search(q, 7)
    i1, d1 = sh.search(q, 7)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    sh.save(str(tmp_path / "s"))
    back = FlatIndex.load(str(tmp_path / "s"), "f32", lib=oracle_lib, devices=[0, 1])
    i2, d2 = back.search(q, 7)
    assert np.array_equal(i0, i2) and np.array_equal(d0, d2) and back.get_stat("n_shards") == 2
    with pytest.raises(VdbError):
        FlatIndex(24, "f32", lib=oracle_lib, devices=[])
    for ix in (plain, sh, back):
        ix.close()


def test_search_fetch_on_the_oracle_twin_and_through_the_collection(oracle_lib, tmp_path):
    rng = np.random.default_rng(5)
    rows = rng.standard_normal((50, 16)).astype(np.float32)
    ix = FlatIndex(16, "f32", lib=oracle_lib)
    ix.add(rows)
    q = rows[[4, 40]] + 0.01
    i0, d0 = ix.search(q, 60)
    i1, d1, r1 = ix.search_fetch(q, 60)
    assert np.array_equal(i0, i1) and np.array_equal(d0, d1)
    assert np.array_equal(r1[:, :50].reshape(-1, 16), ix.fetch(i1[:, :50].reshape(-1))) and np.all(r1[:, 50:] == 0)
    ix.close()
    col = _col(PersistentClient(path=str(tmp_path), lib=oracle_lib))
    col.add(embeddings=rows, ids=[f"i{j}" for j in range(50)])
    res = col.query(query_embeddings=q, n_results=3, include=["embeddings", "distances"])
    assert [r[0] for r in res["ids"]] == ["i4", "i40"]
    for j, src in enumerate((4, 40)):
        assert np.allclose(res["embeddings"][j][0], rows[src] / np.linalg.norm(rows[src]), atol=1e-6)


def test_rerank_on_a_store_without_the_fp32_copy_on_the_oracle_twin(oracle_lib, oracle_mod):
    """A bf16 store created WITHOUT the fp32 copy accepts rerank_kprime: the k' candidates are re-scored against the
    stored bf16 rows, so the answer is rerank_kprime = 0's (the exact top-k of <fp32 query, bf16 row>); an fp32 store
    still refuses a rerank."""
    rng = np.random.default_rng(6)
    rows = rng.standard_normal((300, 48)).astype(np.float32)
    q = rows[[5, 250]] + 0.05 * rng.standard_normal((2, 48)).astype(np.float32)
    ix = FlatIndex(48, "bf16", False, lib=oracle_lib)
    ix.add(rows)
    i0, d0 = ix.search(q, 10, 0)
    i1, d1 = ix.search(q, 10, 40)
    assert np.array_equal(i0, i1) and np.allclose(d0, d1, atol=1e-7)
    src = oracle_mod.round_bf16(oracle_mod.normalize_rows(rows))
    r_idx, r_dist, _ = oracle_mod.exact_topk(src, q, 10)
    assert np.array_equal(i1, r_idx) and np.allclose(d1, r_dist, atol=1e-6)
    ix.close()
    fx = FlatIndex(48, "f32", lib=oracle_lib)
    fx.add(rows)
    with pytest.raises(VdbError) as e:
        fx.search(q, 10, 40)
    assert e.value.code == -5
    fx.close()


def test_collection_without_the_fp32_copy(oracle_lib, oracle_mod, tmp_path):
    """PersistentClient(dtype="bf16", fp32_copy=False): 2 bytes per element; queries (reranked against the bf16 rows) and
    include=["embeddings"] work from the stored bf16 rows; reopening WITH the copy rebuilds from the embeddings as added."""
    rng = np.random.default_rng(8)
    rows = rng.standard_normal((200, 32)).astype(np.float32)
    client = PersistentClient(path=str(tmp_path), dtype="bf16", fp32_copy=False, lib=oracle_lib)
    col = _col(client)
    col.add(embeddings=rows, ids=[f"i{j}" for j in range(200)])
    q = rows[[3, 150]] + 0.01
    res = col.query(query_embeddings=q, n_results=5, include=["embeddings", "distances"])
    assert [r[0] for r in res["ids"]] == ["i3", "i150"]
    want = oracle_mod.round_bf16(oracle_mod.normalize_rows(rows[[3]]))[0]
    assert np.array_equal(np.asarray(res["embeddings"][0][0], np.float32), want)
    client.close()
    hdr = open(tmp_path / "c" / "vectors.vdb", "rb").read(16)
    assert struct.unpack("<ii", hdr[8:16]) == (32, 1)                      # the cache holds bf16 rows
    col2 = _col(PersistentClient(path=str(tmp_path), dtype="bf16", lib=oracle_lib))      # with the copy: rebuilt, fp32 rows again
    res2 = col2.query(query_embeddings=q, n_results=5, include=["embeddings", "distances"])
    assert res2["ids"] == res["ids"]
    assert np.allclose(res2["embeddings"][0][0], rows[3] / np.linalg.norm(rows[3]), atol=1e-6)


def test_batched_best_match_maps_an_empty_answer_to_none(oracle_lib):
    """A product whose embedding is not finite gets no row back from the engine (idx -1 / +inf): it must read
    (None, -1.0) like the reference's loop (generate_best_usda_matches.py:116-143), not the last USDA code."""
    from vectordb_amd.matching import best_usda_matches_batched, find_best_usda_match
    rng = np.random.default_rng(4)
    usda = {f"U{i}": rng.standard_normal(12).astype(np.float32) for i in range(5)}
    prods = rng.standard_normal((3, 12)).astype(np.float32)
    prods[1, 4] = np.nan
    codes, sims = best_usda_matches_batched(prods, usda, lib=oracle_lib)
    assert codes[1] is None and sims[1] == -1.0
    ref = [find_best_usda_match(p, usda) for p in prods]
    assert ref[1][0] is None and ref[1][1] == -1.0
    for j in (0, 2):
        assert codes[j] == ref[j][0] and abs(sims[j] - ref[j][1]) < 1e-6


def test_bench_self_launch_refuses_more_gpus_than_the_node_has():
    """`python bench.py --gpus N` without a launcher starts its own ranks before any GPU call; on this GPU-less
    container the only observable part is that it checks the node's GPU count first and says so."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert p.returncode != 0 and "--gpus 2 but this node shows" in (p.stderr + p.stdout)


def test_collection_with_devices_on_the_oracle_twin(oracle_lib, tmp_path):
    """PersistentClient(path, devices=[...]) (round 3): the Chroma-shaped face on a row-sharded segment — adds that
    outgrow the partition (vdb_rebalance at every doubling), query with embeddings, where=, get, reopen under another
    shard count — answers what the single-store collection answers.  (On the twin a sharded handle IS the plain
    store; what is checked here is the host logic around it.)"""
    rng = np.random.default_rng(17)
    rows = rng.standard_normal((700, 16)).astype(np.float32)
    q = rows[[3, 650]] + 0.01
    metas = [{"usda_code": "A" if i % 4 == 0 else "B"} for i in range(700)]
    out = {}
    for name, kw in (("one", {}), ("sharded", {"devices": [0, 1, 2]})):
        col = _col(PersistentClient(path=str(tmp_path / name), lib=oracle_lib, **kw))
        for lo in range(0, 700, 250):
            col.add(embeddings=rows[lo:lo + 250], metadatas=metas[lo:lo + 250], ids=[f"i{j}" for j in range(lo, min(700, lo + 250))])
        r = col.query(query_embeddings=q, n_results=5, include=["embeddings", "distances", "metadatas"])
        w = col.query(query_embeddings=q, n_results=5, where={"usda_code": "A"}, include=["distances"])
        g = col.get(ids=["i7", "i699"], include=["embeddings"])
        out[name] = (r["ids"], r["distances"], [np.asarray(e).tolist() for e in r["embeddings"]], w["ids"], w["distances"], np.asarray(g["embeddings"]).tolist())
        if kw:
            assert col._index.get_stat("n_shards") == 3 and col._partition_rows >= 700
            col.persist()
    assert out["one"] == out["sharded"]
    again = PersistentClient(path=str(tmp_path / "sharded"), lib=oracle_lib, devices=[0, 1]).get_collection("c")
    assert again.count() == 700 and again._index.get_stat("n_shards") == 2
    assert again.query(query_embeddings=q, n_results=5, include=["distances"])["ids"] == out["one"][0]
    again.add(embeddings=rows[:3] * 3.0, ids=["x0", "x1", "x2"])
    assert again.count() == 703 and again._partition_rows >= 703
    ix = again._index
    ix.warm(8, 5)
    ix.rebalance(10)
    assert ix.count() == 703


def test_bench_two_ranks_self_launch_reaches_the_gpu_check_and_a_failing_inproc_child_is_an_error_string():
    """Round 3: `bench.py --gpus 2 --device-map 0,0 --backend gloo` (the diagnostic two-ranks-on-one-GPU form) passes the
    node's GPU-count check, starts its ranks through torch.distributed.run and — on this GPU-less container — each rank
    stops at "needs a GPU"; and the one-process sharded block, which runs in a child process, comes back as an error
    string when that child fails instead of taking the JSON line down with it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--device-map", "0,0", "--backend", "gloo", "--rows", "1000",
                        "--steps", "1", "--warmup", "0", "--no-traffic-pass", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode != 0 and "bench.py needs a GPU" in (p.stderr + p.stdout) and "this node shows" not in (p.stderr + p.stdout)
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    old_argv = sys.argv
    sys.argv = ["bench.py", "--gpus", "2", "--rows", "1000", "--steps", "1", "--warmup", "0"]
    try:
        args = bench.parse_args()
    finally:
        sys.argv = old_argv
    assert bench.device_map(args) == [] and args.backend == "nccl"
    blk = bench.run_inproc_child(args, 2)
    assert set(blk) == {"error"} and "exited with" in blk["error"]
