#!/bin/sh
# Build an experimental variant of the library next to the product one:
#   tools/build_variant.sh <name> <extra hipcc flags>   ->  brotli_amd/libyeast_<name>.so   (git-ignored)
# The product library is rebuilt afterwards so the objects in csrc/ match it again.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
YEAST_EXTRA_FLAGS="$*" brotli_amd/csrc/build.sh
mv brotli_amd/libyeast_mi355x.so "brotli_amd/libyeast_$NAME.so"
brotli_amd/csrc/build.sh
