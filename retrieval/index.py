from dhr_amd.retrieval.index import *  # noqa: F401,F403
from dhr_amd.retrieval.index import main

if __name__ == "__main__":
    main()
