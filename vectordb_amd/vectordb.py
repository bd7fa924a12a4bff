"""Host-side mirror of the reference's src/vectordb.py public surface, backed by
the MI355X engine instead of chromadb's HNSW.

Same names, arguments and return shapes as the reference so its analysis scripts
run unchanged against this module:

  ProductEmbedder            src/vectordb.py:13-69
  normalize_mapping_id       src/vectordb.py:72-81
  build_usda_lookup          src/vectordb.py:84-140
  ProductVectorDB            src/vectordb.py:142-337
  create_product_vector_db   src/vectordb.py:340-379
  find_similar_products      src/vectordb.py:382-422

The search itself (collection.query) runs on the GPU through the C ABI; the
bi-directional rerank of src/vectordb.py:277-335 is the small host epilogue
implemented in `bidirectional_rerank` below.
"""
from __future__ import annotations

import re
from typing import Any, Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import pandas as pd

from . import config
from .chroma_compat import PersistentClient
from .embedders import resolve_embedder

RESULT_COLUMNS = ["id", "forward_similarity", "backward_similarity", "bi_directional_similarity",
                  "distance", "product_description", "product_code", "usda_code"]
_METADATA_COLUMNS = ["product_description", "product_code", "usda_code"]


class ProductEmbedder:
    """Text -> 768-d vectors.  `self.model` is a callable(list[str]) -> list of
    vectors, as chromadb's SentenceTransformerEmbeddingFunction is in the
    reference (src/vectordb.py:17); see embedders.py for what can back it."""

    def __init__(self, model_name: str = config.EMBEDDING_MODEL, embed_fn: Optional[Callable] = None):
        self.model_name = model_name
        self.model = resolve_embedder(model_name, embed_fn)

    def _prepare_metadata(self, df: pd.DataFrame) -> List[Dict[str, Any]]:
        missing = [c for c in _METADATA_COLUMNS if c not in df.columns]
        if missing:
            raise ValueError(f"Metadata preparation error: Column '{missing[0]}' not found in DataFrame.")
        return df[_METADATA_COLUMNS].fillna("N/A").astype(str).to_dict("records")

    def embed_products(self, df: pd.DataFrame) -> Tuple[List[str], List[np.ndarray], List[Dict[str, Any]]]:
        if "product_description" not in df.columns:
            raise ValueError("DataFrame must contain 'product_description' column for embedding.")
        embeddings = [np.asarray(e, dtype=np.float32) for e in self.model(df["product_description"].tolist())]
        ids = [f"item_{i}" for i in range(len(df))]          # positional ids, src/vectordb.py:57
        return ids, embeddings, self._prepare_metadata(df)

    def embed_products_device(self, df: pd.DataFrame):
        """embed_products with the embeddings left on the GPU as one [n, dim] tensor (embedders that
        offer embed_device on a CUDA device only) — the ingest then skips the host round trip of
        src/vectordb.py:216."""
        if "product_description" not in df.columns:
            raise ValueError("DataFrame must contain 'product_description' column for embedding.")
        emb = self.model.embed_device(df["product_description"].tolist())
        ids = [f"item_{i}" for i in range(len(df))]
        return ids, emb, self._prepare_metadata(df)

    @property
    def on_device(self) -> bool:
        return hasattr(self.model, "embed_device") and str(getattr(self.model, "device", "cpu")).startswith("cuda")

    def embed_query(self, query: str) -> np.ndarray:
        return np.array(self.model([query])[0])


def normalize_mapping_id(code):
    """'1234-5' -> '1234'; a leading company digit on an all-digit id is dropped
    ('51040948' -> '1040948'); non-strings are stringified and stripped."""
    if not isinstance(code, str):
        return str(code).strip()
    code = re.sub(r"-\d+$", "", code).strip()
    return re.sub(r"^\d(\d+)$", r"\1", code)


def build_usda_lookup(mapping_file=None, sheet_name=None, id_cols=None, usda_col=None,
                      mapping_df: Optional[pd.DataFrame] = None) -> Dict[str, str]:
    """{normalised mapping id -> USDA code}; later rows win on conflicts.  Load
    failures yield {} (the reference prints and returns {}, src/vectordb.py:93-98)."""
    mapping_file = config.GROUND_TRUTH_FILE if mapping_file is None else mapping_file
    sheet_name = config.GROUND_TRUTH_SHEET_NAME if sheet_name is None else sheet_name
    id_cols = list(config.GROUND_TRUTH_ID_COLS if id_cols is None else id_cols)
    usda_col = config.GROUND_TRUTH_USDA_COL if usda_col is None else usda_col
    if mapping_df is None:
        try:
            from .data_processing import read_excel
            mapping_df = read_excel(mapping_file, sheet_name=sheet_name)
        except Exception as exc:  # missing file, unreadable workbook
            print(f"Error loading mapping data: {exc}")
            return {}
    if any(c not in mapping_df.columns for c in id_cols + [usda_col]):
        print(f"Error: Missing one or more required columns in mapping file. Need: {id_cols + [usda_col]}. "
              f"Found: {mapping_df.columns.tolist()}")
        return {}
    lookup: Dict[str, str] = {}
    for _, row in mapping_df.iterrows():
        usda = str(row[usda_col]) if pd.notna(row[usda_col]) else None
        if not usda:
            continue
        for col in id_cols:
            if pd.notna(row[col]):
                key = normalize_mapping_id(row[col])
                if key:
                    lookup[key] = usda
    return lookup


def bidirectional_rerank(ids, distances, metadatas, embeddings, query_embedding, n_results: int,
                         similarity_threshold: Optional[float]) -> pd.DataFrame:
    """Host epilogue of get_similar_products (reference src/vectordb.py:277-335).

    forward = 1 - distance; backward = cos(item, query) recomputed from the
    returned embeddings (0 when either norm is 0); score = their mean; rows
    sorted by the mean (descending), thresholded with >=, cut to n_results;
    the 8 expected columns first, in fixed order.  Arithmetic is float64."""
    q = np.asarray(query_embedding, dtype=np.float64).reshape(-1)
    items = np.asarray(embeddings, dtype=np.float64).reshape(len(ids), -1)
    fwd = np.array([1 - d for d in distances], dtype=np.float64)
    denom = np.linalg.norm(items, axis=1) * np.linalg.norm(q)
    dots = items @ q
    bwd = np.divide(dots, denom, out=np.zeros_like(dots), where=denom != 0)
    records = []
    for i, item_id in enumerate(ids):
        records.append({"id": item_id, "forward_similarity": float(fwd[i]), "backward_similarity": float(bwd[i]),
                        "bi_directional_similarity": float((fwd[i] + bwd[i]) / 2), "distance": distances[i],
                        **(metadatas[i] or {})})
    df = pd.DataFrame(records).sort_values("bi_directional_similarity", ascending=False)
    if similarity_threshold is not None:
        df = df[df["bi_directional_similarity"] >= similarity_threshold]
    df = df.head(n_results)
    for col in RESULT_COLUMNS:
        if col not in df.columns:
            df[col] = "N/A"
    return df[RESULT_COLUMNS + [c for c in df.columns if c not in RESULT_COLUMNS]]


class ProductVectorDB:
    """Same constructor, attributes (`client`, `embedder`, `collection`,
    `collection_name`, `usda_lookup_map`) and methods as the reference class
    (src/vectordb.py:142-337).  Extra keyword arguments configure the engine."""

    def __init__(self, persist_directory: str = None, collection_name: str = config.COLLECTION_NAME,
                 embedding_model_name: str = config.EMBEDDING_MODEL, *, embed_fn: Optional[Callable] = None,
                 usda_lookup_map: Optional[Dict[str, str]] = None, dtype: str = None, device: int = None,
                 lib=None, client: Optional[PersistentClient] = None, devices: Optional[Sequence[int]] = None,
                 expected_rows: Optional[int] = None):
        self.persist_directory = str(config.CHROMA_DB_PATH) if persist_directory is None else persist_directory
        self.collection_name = collection_name
        self.client = client if client is not None else PersistentClient(
            path=self.persist_directory, dtype=dtype or config.STORE_DTYPE,
            device=config.DEVICE if device is None else device, lib=lib, devices=devices, expected_rows=expected_rows)   # devices=[...]: row-sharded over those GPUs
        self.embedder = ProductEmbedder(model_name=embedding_model_name, embed_fn=embed_fn)
        self.usda_lookup_map = build_usda_lookup() if usda_lookup_map is None else dict(usda_lookup_map)
        self.collection = self._open_collection()

    def _open_collection(self):
        return self.client.get_or_create_collection(name=self.collection_name, embedding_function=self.embedder.model,
                                                    metadata={"hnsw:space": "cosine"})

    def _get_usda_code(self, product_code: str) -> str:
        return self.usda_lookup_map.get(normalize_mapping_id(product_code), "NOT_FOUND")

    def add_products_to_db(self, unique_products_df: pd.DataFrame, recreate: bool = False, batch_size: int = 5000):
        if recreate:
            try:
                self.client.delete_collection(name=self.collection_name)
            except Exception as exc:
                print(f"Warning: Could not delete collection {self.collection_name} (may not exist): {exc}")
            self.collection = self._open_collection()
        if not {"product_description", "product_code"} <= set(unique_products_df.columns):
            raise ValueError("Input DataFrame must contain 'product_description' and 'product_code' columns.")
        unique_products_df["usda_code"] = unique_products_df["product_code"].apply(self._get_usda_code)
        if getattr(self.embedder, "on_device", False) and getattr(self.collection, "_lib", 0) is None:
            ids, dev_emb, metadatas = self.embedder.embed_products_device(unique_products_df)
            for lo in range(0, len(ids), batch_size):
                self.collection.add(embeddings=dev_emb[lo:lo + batch_size], metadatas=metadatas[lo:lo + batch_size],
                                    ids=ids[lo:lo + batch_size])
            return
        ids, embeddings, metadatas = self.embedder.embed_products(unique_products_df)
        for lo in range(0, len(ids), batch_size):
            hi = lo + batch_size
            # one contiguous fp32 block per batch instead of the reference's list-of-lists (src/vectordb.py:216)
            self.collection.add(embeddings=np.stack(embeddings[lo:hi]) if embeddings[lo:hi] else [],
                                metadatas=metadatas[lo:hi], ids=ids[lo:hi])

    def get_similar_products(self, query: str, n_results: int = 5, similarity_threshold: Optional[float] = None,
                             where_filter: Optional[Dict[str, Any]] = None,
                             initial_results: int = config.N_RESULTS_INITIAL_SEARCH) -> pd.DataFrame:
        query_embedding = self.embedder.embed_query(query)
        res = self.collection.query(query_embeddings=[np.asarray(query_embedding).tolist()], n_results=initial_results,
                                    include=["metadatas", "distances", "embeddings"], where=where_filter)
        ids = res.get("ids", [[]])[0]
        if not ids:
            return pd.DataFrame()
        return bidirectional_rerank(ids, res["distances"][0], res["metadatas"][0], res["embeddings"][0],
                                    query_embedding, n_results, similarity_threshold)


def create_product_vector_db(recreate: bool = False, **db_kwargs) -> Tuple[Optional[ProductVectorDB], Optional[pd.DataFrame]]:
    """Load + clean the transaction report, open the store, add every unique
    product.  Returns (db, unique_products_df) or (None, None) when the data
    cannot be processed (reference src/vectordb.py:340-379)."""
    from .data_processing import load_transaction_data, process_transaction_data
    unique_products_df = process_transaction_data(load_transaction_data())
    if unique_products_df is None or unique_products_df.empty:
        print("Error: Failed to process transaction data. Cannot create vector DB.")
        return None, None
    db_kwargs.setdefault("persist_directory", str(config.CHROMA_DB_PATH))
    db_kwargs.setdefault("collection_name", config.COLLECTION_NAME)
    db_kwargs.setdefault("embedding_model_name", config.EMBEDDING_MODEL)
    vector_db = ProductVectorDB(**db_kwargs)
    # recreate=False reopens the persisted store; rows whose ids already exist are
    # skipped by add() instead of being embedded into the index a second time.
    vector_db.add_products_to_db(unique_products_df, recreate=recreate)
    return vector_db, unique_products_df


def find_similar_products(query: str, n_results: int = 5, similarity_threshold: Optional[float] = None,
                          vector_db: Optional[ProductVectorDB] = None,
                          initial_results: int = config.N_RESULTS_INITIAL_SEARCH) -> pd.DataFrame:
    if vector_db is None:
        vector_db, _ = create_product_vector_db(recreate=False)
        if vector_db is None:
            print("Failed to load or create vector database.")
            return pd.DataFrame()
    return vector_db.get_similar_products(query, n_results=n_results, similarity_threshold=similarity_threshold,
                                          initial_results=initial_results)
