#!/bin/bash
# Same-box A/B: event-timed attention launches with the library named by $1 (default: the in-tree build), twice.
set -o pipefail
for rep in 1 2; do
for lib in "$@"; do
  export SDAG_HIP_LIB=$PWD/$lib
  echo "== $lib"
  timeout -k 10 120 python3 tools/attn_microbench.py one || exit 1
  timeout -k 10 120 python3 tools/attn_microbench.py cfg3b || exit 1
done
done
