#!/bin/sh
# Build the bounds-checked debug library (-DYS_BOUNDS, brotli_amd/csrc/yeast_bounds.h) next to the product one:
# brotli_amd/libyeast_bounds.so (git-ignored).  Use: YEAST_LIB=$PWD/brotli_amd/libyeast_bounds.so python -m pytest tests -m gpu
set -e
cd "$(dirname "$0")/.."
YEAST_EXTRA_FLAGS=-DYS_BOUNDS brotli_amd/csrc/build.sh
mv brotli_amd/libyeast_mi355x.so brotli_amd/libyeast_bounds.so
brotli_amd/csrc/build.sh
