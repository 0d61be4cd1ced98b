"""Drop-in module path of the reference (`python -m retrieval.gip_retrieval`, `python -m retrieval.index`):
thin aliases of dhr_amd.retrieval."""
