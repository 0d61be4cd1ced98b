"""`python retrieval/merge.result.py --total_shrad N` -- alias of dhr_amd.retrieval.trec.merge_main
(the reference's script name; it is run as a file, the dot keeps it from being imported)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dhr_amd.retrieval.trec import merge_main as main  # noqa: E402

if __name__ == "__main__":
    main()
