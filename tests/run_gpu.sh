#!/bin/bash
# helper for gpurun: run the GPU test-suite, keep the log under gpurun_out/
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q "$@" 2>&1 | tee gpurun_out/pytest_gpu.log | tail -60
