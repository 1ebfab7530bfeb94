"""Same command line as the reference's scripts/reduce_results.py, on the MI355X merge kernel."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mural_amd.tools.reduce_results import main  # noqa: E402

if __name__ == '__main__':
    main()
