#!/bin/bash
# Same-box A/B of two builds of the library: tools/_base.so against the in-tree one, alternating, default bench.
# usage: tools/ab_bench.sh <tag> [rounds]
set -o pipefail
tag=${1:-ab}; rounds=${2:-3}
out=gpurun_out/$tag; mkdir -p $out
for i in $(seq 1 $rounds); do
  for arm in base new; do
    if [ $arm = base ]; then export SDAG_HIP_LIB=$PWD/tools/_base.so; else unset SDAG_HIP_LIB; fi
    timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/${arm}$i.json 2> $out/${arm}$i.err || { tail -5 $out/${arm}$i.err; exit 1; }
    python - <<PY
import json
j = json.load(open("$out/${arm}$i.json"))
print("$arm$i value %.1f q/s  launch %.1f us  per-query %.2f us  cfg4 %.1f us/q-layer  retrieval %.3f ms" % (j["value"], j["roofline"]["us_per_launch"], j["stage"]["sdag_per_query_call_us"], j["stage"]["cfg4_us_per_query_layer"], j["stage"]["retrieval_ms_per_batch"]))
PY
  done
done
