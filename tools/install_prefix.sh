#!/bin/sh
# Copy the built drop-in into the reference's install layout (src/CMakeLists.txt:63-88):
#   <prefix>/include/nm/*.h + NiftyMatchConfig.cmake,  <prefix>/lib/nm/lib{gpuutils,kernels,sift}.a + libnm_hip.so
set -e
PREFIX=${1:?usage: install_prefix.sh <prefix>}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$PREFIX/include/nm" "$PREFIX/lib/nm"
cp "$ROOT"/niftymatch_amd/nm/*.h "$PREFIX/include/nm/"
cp "$ROOT"/include/nm_abi.h "$ROOT"/include/nm_client.h "$PREFIX/include/nm/"
cp "$ROOT"/niftymatch_amd/cmake/NiftyMatchConfig.cmake "$PREFIX/include/nm/"
cp "$ROOT"/niftymatch_amd/lib/nm/*.a "$PREFIX/lib/nm/"
cp "$ROOT"/niftymatch_amd/lib/libnm_hip.so "$PREFIX/lib/nm/"
echo "installed to $PREFIX"
