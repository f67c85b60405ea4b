#!/bin/bash
# Same-box comparison of several builds of the library on the search microbenchmark.
# usage: tools/ab_dense.sh <rounds> <microbench args in quotes> <lib>...   ("tree" = the in-tree build)
set -o pipefail
rounds=$1; args=$2; shift 2
for i in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ $lib = tree ]; then unset SDAG_HIP_LIB; else export SDAG_HIP_LIB=$PWD/$lib; fi
    echo "== $(basename $lib .so) $i"
    timeout -k 10 200 python3 tools/dense_microbench.py $args 2>/dev/null | sed 's/TB.s-equivalent scan, //' || exit 1
  done
done
