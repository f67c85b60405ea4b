#!/bin/bash
# Same-box comparison of several builds of the library (paths given as arguments; "tree" = the in-tree build).
# usage: tools/ab_multi.sh <tag> <rounds> <lib>...
set -o pipefail
tag=$1; rounds=$2; shift 2
out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $rounds); do
  for lib in "$@"; do
    if [ $lib = tree ]; then unset SDAG_HIP_LIB; else export SDAG_HIP_LIB=$PWD/$lib; fi
    name=$(basename $lib .so)
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/${name}_$i.json 2> $out/${name}_$i.err || { tail -5 $out/${name}_$i.err; exit 1; }
    python - <<PY
import json
j = json.load(open("$out/${name}_$i.json"))
print("%-14s %d value %.1f q/s  launch %.1f us  per-query %.2f us  cfg4 %.1f us/q-layer" % ("$name", $i, j["value"], j["roofline"]["us_per_launch"], j["stage"]["sdag_per_query_call_us"], j["stage"]["cfg4_us_per_query_layer"]))
PY
  done
done
