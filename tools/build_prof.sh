#!/bin/sh
# Build the instrumented library (-DYS_PROFILE) next to the product one: brotli_amd/libyeast_prof.so
set -e
cd "$(dirname "$0")/.."
cp brotli_amd/libyeast_mi355x.so /tmp/libyeast_keep.so
YEAST_EXTRA_FLAGS=-DYS_PROFILE brotli_amd/csrc/build.sh
mv brotli_amd/libyeast_mi355x.so brotli_amd/libyeast_prof.so
brotli_amd/csrc/build.sh
