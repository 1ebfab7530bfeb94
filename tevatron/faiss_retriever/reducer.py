"""`python -m tevatron.faiss_retriever.reducer` — same CLI as the reference (reducer.py:23-39)."""
from mural_amd.faiss_retriever.reducer import combine_faiss_results, main  # noqa: F401

if __name__ == '__main__':
    main()
