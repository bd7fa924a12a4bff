"""The embed_fn seam.  The reference embeds text with
chromadb's SentenceTransformerEmbeddingFunction('all-mpnet-base-v2')
(src/vectordb.py:17, src/config.py:36): callable(list[str]) -> list of 768-d
unit vectors.  That model is loaded by NAME from the network, which is not
possible offline, so the seam accepts:

  * any callable(list[str]) -> array-like [n, dim];
  * MPNetEmbedder(local_dir): transformers MPNet + mean pooling + L2 normalise
    on PyTorch-ROCm, from a user-supplied local checkpoint directory;
  * HashEmbedder: deterministic hashed character-n-gram embedding used by tests
    and demos (NOT semantically equivalent to MPNet).
"""
from __future__ import annotations

import hashlib
import os
from typing import Callable, List, Sequence

import numpy as np


class HashEmbedder:
    """Deterministic text -> unit vector.  Signed feature hashing of word unigrams
    and character trigrams, so that strings sharing words land near each other."""

    def __init__(self, dim: int = 768):
        self.dim = dim

    def _features(self, text: str):
        t = " ".join(str(text).lower().split())
        for w in t.split(" "):
            if w:
                yield "w:" + w, 2.0
        padded = f"  {t}  "
        for i in range(len(padded) - 2):
            yield "c:" + padded[i:i + 3], 1.0

    def __call__(self, texts: Sequence[str]) -> List[np.ndarray]:
        out = []
        for text in texts:
            v = np.zeros(self.dim, np.float64)
            for feat, w in self._features(text):
                h = int.from_bytes(hashlib.blake2b(feat.encode(), digest_size=8).digest(), "little")
                v[h % self.dim] += w if (h >> 63) & 1 else -w
            n = np.linalg.norm(v)
            out.append((v / n if n > 0 else v).astype(np.float32))
        return out


class MPNetEmbedder:
    """sentence-transformers/all-mpnet-base-v2 semantics from a LOCAL directory:
    tokenizer + MPNetModel -> attention-masked mean pooling -> L2 normalise."""

    def __init__(self, local_dir: str, device: str = None, batch_size: int = 64, max_length: int = 384,
                 autocast: str = None):
        """autocast: None (fp32 forward, the reference's precision), "bf16" or "fp16" (GPU only; the
        pooled vector is still accumulated and normalised in fp32)."""
        import torch
        from transformers import AutoModel, AutoTokenizer
        if not os.path.isdir(local_dir):
            raise FileNotFoundError(f"{local_dir} is not a local model directory (models are never downloaded by name)")
        self._torch = torch
        self.device = device or ("cuda" if torch.cuda.is_available() else "cpu")
        self.tok = AutoTokenizer.from_pretrained(local_dir, local_files_only=True)
        self.model = AutoModel.from_pretrained(local_dir, local_files_only=True).to(self.device).eval()
        self.batch_size = batch_size
        self.max_length = max_length
        if autocast not in (None, "bf16", "fp16"):
            raise ValueError("autocast must be None, 'bf16' or 'fp16'")
        self._amp = {None: None, "bf16": torch.bfloat16, "fp16": torch.float16}[autocast]
        self.dim = int(self.model.config.hidden_size)

    def embed_device(self, texts: Sequence[str]):
        """-> torch.Tensor [n, dim] fp32 on self.device, rows unit length.  On a GPU the result can go
        straight into the store (Collection.add accepts it; vdb_add_dev) without the host round trip
        of the reference's `embeddings.tolist()` (src/vectordb.py:216)."""
        torch = self._torch
        texts = list(texts)
        out = torch.empty((len(texts), self.dim), dtype=torch.float32, device=self.device)
        dev_type = "cuda" if str(self.device).startswith("cuda") else "cpu"
        with torch.no_grad():
            for i in range(0, len(texts), self.batch_size):
                enc = self.tok(texts[i:i + self.batch_size], padding=True, truncation=True,
                               max_length=self.max_length, return_tensors="pt").to(self.device)
                with torch.autocast(dev_type, dtype=self._amp, enabled=self._amp is not None and dev_type == "cuda"):
                    hid = self.model(**enc).last_hidden_state
                mask = enc["attention_mask"].unsqueeze(-1).float()
                pooled = (hid.float() * mask).sum(1) / mask.sum(1).clamp(min=1e-9)
                out[i:i + self.batch_size] = torch.nn.functional.normalize(pooled, p=2, dim=1)
        return out

    def __call__(self, texts: Sequence[str]) -> List[np.ndarray]:
        return list(self.embed_device(texts).cpu().numpy())


def SentenceTransformerEmbeddingFunction(model_name: str = "all-mpnet-base-v2", **_ignored):
    """Same name and call shape as chromadb.utils.embedding_functions' class, so that
    `from vectordb_amd import embedders as embedding_functions` leaves the reference's
    `src/vectordb.py:17` unchanged.  Resolves offline only (see resolve_embedder)."""
    return resolve_embedder(model_name)


def resolve_embedder(model_name: str, embed_fn: Callable = None):
    """embed_fn wins; a local directory -> MPNetEmbedder; 'hash'/'hash:<dim>' -> HashEmbedder."""
    if embed_fn is not None:
        return embed_fn
    if isinstance(model_name, str) and model_name.startswith("hash"):
        dim = int(model_name.split(":")[1]) if ":" in model_name else 768
        return HashEmbedder(dim)
    local = os.environ.get("VECTORDB_MODEL_DIR", model_name)
    if isinstance(local, str) and os.path.isdir(local):
        return MPNetEmbedder(local)
    raise RuntimeError(
        f"embedding model {model_name!r} is not available offline: pass embed_fn=..., set VECTORDB_MODEL_DIR to a local "
        "checkpoint directory, or use embedding_model_name='hash' for the deterministic test embedder")
