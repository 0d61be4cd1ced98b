from dhr_amd.retrieval.gip_retrieval import *  # noqa: F401,F403
from dhr_amd.retrieval.gip_retrieval import main

if __name__ == "__main__":
    main()
