#!/bin/bash
# A/B builds of the library: tools/ab_build.sh name "-DFLAG=1 ..." -> dhr_amd/csrc/_ab/libdhr_hip_<name>.so (git-ignored, travels with gpurun);
# run with DHR_HIP_LIB=<path> (dhr_amd/_lib.py).  Same sources and flags as dhr_amd/_build.py otherwise.
set -e
cd "$(dirname "$0")/../dhr_amd/csrc"
mkdir -p _ab
SRCS=$(python3 -c "import sys; sys.path.insert(0, '../..'); from dhr_amd import _build; print(' '.join(_build.SOURCES))")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-unused-result $2 -o _ab/libdhr_hip_$1.so \
  $SRCS -L/opt/rocm/lib -lrccl
echo built _ab/libdhr_hip_$1.so
