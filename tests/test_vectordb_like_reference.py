"""The checks of the reference's own tests/test_vectordb.py (embedding shape and norm :30-58,
database creation count :82-108, similarity search finds the expected product type :110-138,
a product's own description comes back first :140-185, recall of the description clusters of
the mapping file :187-286), written against this build's mirror API.  The embedding model
cannot be fetched, so the deterministic hash embedder stands in; the engine is the CPU oracle
library here and the HIP library in the `gpu` variants."""
import os

import numpy as np
import pandas as pd
import pytest

from vectordb_amd.vectordb import ProductEmbedder, ProductVectorDB, find_similar_products

CATALOGUE = [
    "almond milk unsweetened 32oz", "almond milk vanilla", "oat milk barista", "whole milk gallon",
    "beef frozen ground 80/20", "beef frozen patties", "premium beef ribeye fresh", "pork shoulder frozen",
    "apple gala organic", "apple fuji fresh", "apple juice 64oz", "orange juice pulp free", "orange juice concentrate frozen",
    "fresh vegetables mixed", "organic bananas", "chicken breast boneless", "turkey breast smoked", "lamb rack frenched",
]
REF = os.environ.get("VECTORDB_REFERENCE", "/root/reference")


def _build(tmp_path, lib, dtype):
    db = ProductVectorDB(persist_directory=str(tmp_path / "store"), collection_name="products", embedding_model_name="hash",
                         usda_lookup_map={}, dtype=dtype, lib=lib)
    df = pd.DataFrame({"product_description": CATALOGUE, "product_code": [str(5000 + i) for i in range(len(CATALOGUE))]})
    db.add_products_to_db(df, recreate=True)
    return db, df


def _checks(tmp_path, lib, dtype):
    # test_embedding_model
    embedder = ProductEmbedder("hash")
    for text in ("almond milk", "frozen beef", "fresh vegetables", "organic bananas"):
        e = embedder.embed_query(text)
        assert e.shape == (768,) and abs(np.linalg.norm(e) - 1.0) < 1e-5
    # test_vector_db_creation
    db, df = _build(tmp_path, lib, dtype)
    assert db.client.get_collection("products").count() == len(CATALOGUE) > 0
    # test_similarity_search
    for query, expected in (("almond milk unsweetened", "almond milk"), ("premium frozen beef", "beef frozen"),
                            ("fresh organic apples", "apple"), ("orange juice", "juice")):
        res = find_similar_products(query, n_results=3, vector_db=db)
        assert len(res) == 3 and any(expected in d.lower() for d in res["product_description"]), (query, list(res["product_description"]))
        assert list(res["bi_directional_similarity"]) == sorted(res["bi_directional_similarity"], reverse=True)
    # test_performance_metrics: every product's own description finds itself first
    for desc in df["product_description"]:
        top = db.get_similar_products(desc, n_results=3).iloc[0]
        assert top["product_description"].lower() == desc.lower() and top["distance"] < 1e-5
    # thresholded search returns only rows at or above the threshold
    res = find_similar_products("almond milk", n_results=20, similarity_threshold=0.25, vector_db=db)
    assert len(res) >= 2 and (res["bi_directional_similarity"] >= 0.25).all()
    return db


def test_reference_checks_on_the_oracle_engine(tmp_path, oracle_lib):
    _checks(tmp_path, oracle_lib, "f32")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["f32", "bf16"])
def test_reference_checks_on_the_hip_engine(tmp_path, hip_lib, dtype):
    _checks(tmp_path, hip_lib, dtype)


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "data", "CorrectMapping", "product_mapping_semantic.xlsx")),
                    reason="reference data files not present")
def test_mapping_recall_on_the_reference_clusters(tmp_path, oracle_lib):
    """test_mapping_recall (:187-286) on the reference's own cluster file and its semantic transaction set:
    query = standardized name, found = possible descriptions among the thresholded top 20.  With the hash
    embedder the figure itself is not comparable; the bookkeeping is exercised end to end and a cluster
    whose standardized name shares words with its variations must be recalled."""
    from vectordb_amd.data_processing import read_excel
    mapping = read_excel(os.path.join(REF, "data", "CorrectMapping", "product_mapping_semantic.xlsx"))
    tx = read_excel(os.path.join(REF, "data", "Transactions", "product_transactions_semantic.xlsx"))
    assert list(mapping.columns) == ["Standardized Product Name", "Product Codes", "Possible Descriptions"] and len(mapping) == 20
    descs = sorted({str(d).strip().lower() for d in tx["Product Description"].dropna()})
    db = ProductVectorDB(persist_directory=str(tmp_path / "s"), collection_name="products", embedding_model_name="hash",
                         usda_lookup_map={}, lib=oracle_lib)
    db.add_products_to_db(pd.DataFrame({"product_description": descs, "product_code": [str(i) for i in range(len(descs))]}), recreate=True)
    total_possible = total_found = 0
    for _, row in mapping.iterrows():
        possible = [d.strip() for d in row["Possible Descriptions"].split(";")]
        res = find_similar_products(row["Standardized Product Name"], n_results=20, similarity_threshold=0.25, vector_db=db)
        found = {p for p in possible if any(p.lower() == r.lower() for r in res.get("product_description", []))}
        total_possible += len(possible)
        total_found += len(found)
    assert total_possible > 20 and 0 < total_found <= total_possible
