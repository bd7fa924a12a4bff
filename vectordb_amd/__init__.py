"""vectordb_amd — MI355X-native exact cosine top-k search behind the
billenewman4/VectorDB ProductVectorDB / Chroma-style query API.

Importing the package does not touch the GPU; the HIP engine
(vectordb_amd/lib/libvdb_hip.so) is loaded on first use and there is no CPU
fallback."""
from .index import FlatIndex, SynthSpec, VdbError  # noqa: F401
from .chroma_compat import PersistentClient, EphemeralClient, Collection  # noqa: F401

__all__ = ["FlatIndex", "SynthSpec", "VdbError", "PersistentClient", "EphemeralClient", "Collection"]
