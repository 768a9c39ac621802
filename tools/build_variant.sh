#!/bin/bash
# A/B builds of the native library (never shipped): tools/build_variant.sh <name> [git-rev|-] [extra hipcc flags...]
#   -> gpurun_in/libs/<name>.so, loaded with ALFALFA_AMD_LIB=...   (gpurun_in/ is git-ignored and travels to the GPU box)
set -eu
cd "$(dirname "$0")/.."
name=$1; rev=${2:--}; shift; shift || true
mkdir -p gpurun_in/libs
src=alfalfa_amd/csrc; inc=include
if [ "$rev" != "-" ]; then
  tmp=$(mktemp -d); git archive "$rev" alfalfa_amd/csrc include | tar -x -C "$tmp"; src=$tmp/alfalfa_amd/csrc; inc=$tmp/include
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function "$@" $src/parser.cpp $src/runtime.cpp $src/kernels.hip $src/parse_kernels.hip -o gpurun_in/libs/$name.so
ls -la gpurun_in/libs/$name.so
