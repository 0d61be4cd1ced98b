"""`python -m retrieval.rcap_eval ...` -- alias of dhr_amd.retrieval.trec.rcap_main (reference module path)."""
from dhr_amd.retrieval.trec import recall_cap, rcap_main as main  # noqa: F401

if __name__ == "__main__":
    main()
