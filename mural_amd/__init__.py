"""mural_amd — MI355X (gfx950) exact inner-product retrieval for the tevatron.faiss_retriever
hot path of MURAL. Product code: csrc/ (HIP kernels + C ABI, built into libipx.so),
faiss_retriever/ (host-side mirror of the reference interface), sharded.py (multi-GPU)."""
__version__ = "0.1"
