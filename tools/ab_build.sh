#!/bin/bash
# A/B builds of the native library: tools/ab_build.sh <tag> <hipcc defines...>  ->  tools/ab/lib_<tag>.so (travels to the GPU box;
# load it with SPS_LIB=tools/ab/lib_<tag>.so).  Same flags as sps_amd/_build.py.
set -e
tag=$1; shift
cd "$(dirname "$0")/.."
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" -o tools/ab/lib_$tag.so sps_amd/csrc/sps_hip.hip
