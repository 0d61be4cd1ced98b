"""`python -m retrieval.quantize_index ...` -- alias of dhr_amd.retrieval.quantize_index (reference module path)."""
from dhr_amd.retrieval.quantize_index import *  # noqa: F401,F403
from dhr_amd.retrieval.quantize_index import main

if __name__ == "__main__":
    main()
