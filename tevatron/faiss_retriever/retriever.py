from mural_amd.faiss_retriever.retriever import *  # noqa: F401,F403
from mural_amd.faiss_retriever.retriever import BaseFaissIPRetriever, FaissRetriever, FlatIPIndex  # noqa: F401
