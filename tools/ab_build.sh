#!/bin/bash
# A/B builds of the library: tools/ab_build.sh name "-DFLAG=1 ..." -> dhr_amd/csrc/_ab/libdhr_hip_<name>.so (git-ignored, travels with gpurun);
# run with DHR_HIP_LIB=<path> (dhr_amd/_lib.py).  Same sources and flags as dhr_amd/_build.py otherwise.
# DHR_AB_VARIANTS=1 tools/ab_build.sh g8p ""  also links the retired persistent-workgroup form of the integer bound GEMM (tools/ab/gemm_g8p.hip;
# DHR_PARAM_GEMM_VARIANT = 6 / DHR_G8_PERSIST=1): the measurement behind DESIGN.md section 4b, not part of the shipped library.
set -e
cd "$(dirname "$0")/../dhr_amd/csrc"
mkdir -p _ab
SRCS=$(python3 -c "import sys; sys.path.insert(0, '../..'); from dhr_amd import _build; print(' '.join(_build.SOURCES))")
AB=""
[ "$DHR_AB_VARIANTS" = 1 ] && AB="-DDHR_AB_VARIANTS -I."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-unused-result $AB $2 -Wl,--version-script=libdhr.map -o _ab/libdhr_hip_$1.so \
  $SRCS -L/opt/rocm/lib -lrccl
echo built _ab/libdhr_hip_$1.so
