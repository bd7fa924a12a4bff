"""CPU suite for the embed_fn seam: the MPNet embedder from a
LOCAL checkpoint directory (a tiny randomly initialised MPNet written by the test — no model is
ever fetched), its pooling/normalisation contract, and its resolution through the reference's
`SentenceTransformerEmbeddingFunction(model_name=...)` call shape."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
transformers = pytest.importorskip("transformers")

VOCAB = ["<s>", "<pad>", "</s>", "<unk>", "<mask>"] + "beef pork ribeye boneless bone in ground chuck roll steak lamb 80 20 ##s".split()


@pytest.fixture(scope="module")
def tiny_mpnet(tmp_path_factory):
    return _write_tiny_mpnet(str(tmp_path_factory.mktemp("tiny_mpnet")), hidden=32)


def _write_tiny_mpnet(d, hidden):
    from transformers import MPNetConfig, MPNetModel, MPNetTokenizer
    with open(os.path.join(d, "vocab.txt"), "w") as f:
        f.write("\n".join(VOCAB) + "\n")
    MPNetTokenizer(os.path.join(d, "vocab.txt")).save_pretrained(d)
    torch.manual_seed(0)
    cfg = MPNetConfig(vocab_size=len(VOCAB), hidden_size=hidden, num_hidden_layers=2, num_attention_heads=4,
                      intermediate_size=64, max_position_embeddings=64)
    MPNetModel(cfg).save_pretrained(d)
    return d


def test_mpnet_embedder_is_masked_mean_pool_then_l2(tiny_mpnet):
    from transformers import AutoModel, AutoTokenizer
    from vectordb_amd.embedders import MPNetEmbedder
    texts = ["beef ribeye boneless", "ground chuck roll 80 20", "pork", "lamb steaks bone in"]
    emb = MPNetEmbedder(tiny_mpnet, device="cpu", batch_size=3)      # two batches, different padding
    out = np.stack(emb(texts))
    assert out.shape == (4, 32) and out.dtype == np.float32
    assert np.allclose(np.linalg.norm(out, axis=1), 1.0, atol=1e-6)
    # the contract of sentence-transformers' all-mpnet-base-v2 pipeline: attention-masked mean of the last
    # hidden states, then L2 normalisation — recomputed here one text at a time (no padding at all)
    tok = AutoTokenizer.from_pretrained(tiny_mpnet, local_files_only=True)
    model = AutoModel.from_pretrained(tiny_mpnet, local_files_only=True).eval()
    for i, t in enumerate(texts):
        enc = tok([t], return_tensors="pt")
        with torch.no_grad():
            h = model(**enc).last_hidden_state[0]
        v = h.mean(0).numpy()
        assert np.allclose(out[i], v / np.linalg.norm(v), atol=2e-6), t
    assert np.array_equal(np.stack(emb(texts[:1]))[0], np.stack(emb(texts[:1]))[0])       # deterministic
    dev = emb.embed_device(texts)                                                          # the tensor form of the same call
    assert tuple(dev.shape) == (4, 32) and dev.dtype == torch.float32 and np.array_equal(dev.numpy(), out)
    with pytest.raises(ValueError):
        MPNetEmbedder(tiny_mpnet, device="cpu", autocast="int8")


def test_embedder_resolution_is_offline_only(tiny_mpnet, monkeypatch, tmp_path):
    from vectordb_amd import embedders
    from vectordb_amd.embedders import HashEmbedder, MPNetEmbedder, SentenceTransformerEmbeddingFunction, resolve_embedder
    monkeypatch.delenv("VECTORDB_MODEL_DIR", raising=False)
    with pytest.raises(RuntimeError, match="not available offline"):
        resolve_embedder("all-mpnet-base-v2")                       # a model NAME is never fetched
    with pytest.raises(FileNotFoundError):
        MPNetEmbedder(str(tmp_path / "nope"))
    assert isinstance(resolve_embedder("hash:64"), HashEmbedder) and resolve_embedder("hash:64").dim == 64
    fn = lambda texts: [np.ones(4, np.float32) for _ in texts]     # noqa: E731
    assert resolve_embedder("whatever", fn) is fn
    assert isinstance(resolve_embedder(tiny_mpnet), MPNetEmbedder)  # a local directory as the model name
    monkeypatch.setenv("VECTORDB_MODEL_DIR", tiny_mpnet)
    ef = SentenceTransformerEmbeddingFunction(model_name="all-mpnet-base-v2")   # the reference's call, src/vectordb.py:17
    assert isinstance(ef, MPNetEmbedder) and len(ef(["beef"])[0]) == 32
    assert embedders.SentenceTransformerEmbeddingFunction is SentenceTransformerEmbeddingFunction


def test_product_vector_db_with_the_local_model(tiny_mpnet, monkeypatch, tmp_path, oracle_lib):
    """The whole add/query path with a real (tiny) transformer behind the embedder seam."""
    import pandas as pd
    from vectordb_amd.vectordb import ProductVectorDB
    monkeypatch.setenv("VECTORDB_MODEL_DIR", tiny_mpnet)
    db = ProductVectorDB(persist_directory=str(tmp_path / "s"), usda_lookup_map={}, lib=oracle_lib)     # default model NAME, resolved locally
    df = pd.DataFrame({"product_description": ["beef ribeye boneless", "ground chuck roll", "pork steak bone in", "lamb steaks"],
                       "product_code": ["1", "2", "3", "4"]})
    db.add_products_to_db(df, recreate=True)
    assert db.collection.count() == 4 and db.collection._dim == 32
    res = db.get_similar_products("beef ribeye boneless", n_results=2)
    assert res.iloc[0]["product_description"] == "beef ribeye boneless" and res.iloc[0]["distance"] < 1e-6
    assert abs(res.iloc[0]["backward_similarity"] - 1.0) < 1e-6


@pytest.mark.gpu
def test_device_resident_embeddings_feed_the_store_in_place(tmp_path, hip_lib):
    """the embedder's output stays in HBM and goes into the engine through
    vdb_add_dev (no tolist()/host round trip, src/vectordb.py:216).  Same store, same answers as
    the host path; bf16 autocast stays within 1e-2 of the fp32 forward."""
    import pandas as pd
    from vectordb_amd.embedders import MPNetEmbedder
    from vectordb_amd.vectordb import ProductVectorDB
    d = _write_tiny_mpnet(str(tmp_path), hidden=64)
    emb = MPNetEmbedder(d, device="cuda:0", batch_size=16)
    words = VOCAB[5:17]
    texts = sorted({f"{a} {b} {c}" for a in words for b in words[:5] for c in words[7:]})[:200]
    df = pd.DataFrame({"product_description": texts, "product_code": [str(1000 + i) for i in range(len(texts))]})
    dev_db = ProductVectorDB(persist_directory=str(tmp_path / "dev"), usda_lookup_map={}, embed_fn=emb, dtype="bf16")
    assert dev_db.embedder.on_device
    dev_db.add_products_to_db(df.copy(), recreate=True, batch_size=64)
    host_fn = lambda t: emb(t)                                                     # noqa: E731  (hides embed_device)
    host_db = ProductVectorDB(persist_directory=str(tmp_path / "host"), usda_lookup_map={}, embed_fn=host_fn, dtype="bf16")
    assert not host_db.embedder.on_device
    host_db.add_products_to_db(df.copy(), recreate=True, batch_size=64)
    assert dev_db.collection.count() == host_db.collection.count() == len(texts)
    a = dev_db.collection._index.fetch(np.arange(len(texts)))
    b = host_db.collection._index.fetch(np.arange(len(texts)))
    assert np.array_equal(a, b)                                                    # the same normalised rows in the store
    got = dev_db.collection.get(ids=["item_7"], include=["embeddings"])
    assert np.allclose(np.asarray(got["embeddings"][0]), emb([texts[7]])[0], atol=1e-6)
    for q in texts[::40]:
        ra, rb = dev_db.get_similar_products(q, n_results=5), host_db.get_similar_products(q, n_results=5)
        assert ra["id"].tolist() == rb["id"].tolist() and ra.iloc[0]["product_description"] == q
    amp = MPNetEmbedder(d, device="cuda:0", autocast="bf16").embed_device(texts[:32])
    ref = emb.embed_device(texts[:32])
    assert amp.dtype == torch.float32 and float((amp - ref).abs().max()) < 1e-2
    assert np.allclose(amp.norm(dim=1).cpu().numpy(), 1.0, atol=1e-5)
