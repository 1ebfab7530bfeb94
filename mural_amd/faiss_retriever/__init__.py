"""MI355X-native drop-in for tevatron.faiss_retriever
(/root/reference/src/tevatron/faiss_retriever/__init__.py:1 exports the retriever classes)."""
from .retriever import BaseFaissIPRetriever, FaissRetriever, FlatIPIndex  # noqa: F401
