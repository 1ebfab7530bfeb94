"""Command-line helpers around the retrieval path (mirrors of the reference's scripts/)."""
