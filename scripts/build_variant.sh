#!/bin/bash
# build_variant.sh NAME "-DFOO=1 ..." : a copy of libleccr_hip.so with extra compile flags for ONE source (default gemm256w4.hip),
# written to gpurun_variants/libleccr_NAME.so (diagnostic builds for scripts/gemm_bench.py via LECCR_LIB=...).
set -e
cd "$(dirname "$0")/.."
name=$1; flags=$2; src=${3:-gemm256w4}
mkdir -p gpurun_variants/obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags $([ "$src" = attention_mfma ] && echo -fno-slp-vectorize) -c leccr_amd/csrc/$src.hip -o gpurun_variants/obj/${src}_$name.o
objs=$(ls leccr_amd/csrc/build/*.o | grep -v "/$src.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_variants/libleccr_$name.so $objs gpurun_variants/obj/${src}_$name.o
echo gpurun_variants/libleccr_$name.so
