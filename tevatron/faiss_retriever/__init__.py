"""Same import surface as the reference package (src/tevatron/faiss_retriever/__init__.py:1)."""
from mural_amd.faiss_retriever.retriever import BaseFaissIPRetriever, FaissRetriever  # noqa: F401
