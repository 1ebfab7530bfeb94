"""`python -m tevatron.faiss_retriever` — same CLI as the reference (__main__.py:56-96)."""
from mural_amd.faiss_retriever.__main__ import (  # noqa: F401
    main, pickle_load, pickle_save, search_queries, write_ranking)

if __name__ == '__main__':
    main()
