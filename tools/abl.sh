for a in ${ABLS:-0 1 2 3 4 5}; do echo "ABL $a"; DHR_GEMM_ABLATE=$a timeout 200 python tools/gemm_bench.py --dlr 768 --k 1536 --idx-buckets 2 2>&1 | tail -1; done
