#!/bin/bash
# A/B helper: attention parity tests + the default bench without CPU baseline; prints value, launch time, probes.
# usage: tools/ab_quick.sh <tag>
set -o pipefail
tag=${1:-ab}
out=gpurun_out/$tag
mkdir -p $out
timeout -k 10 400 python -u -m pytest tests/test_gpu_first.py tests/test_gpu_more.py tests/test_gpu_e2e.py -x -q -k "attention or generate or rag" --timeout 240 --timeout-method thread > $out/pytest.log 2>&1 || { tail -20 $out/pytest.log; exit 1; }
tail -2 $out/pytest.log
for i in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $out/bench$i.json 2> $out/bench$i.err || { tail -5 $out/bench$i.err; exit 1; }
python - <<PY
import json
j = json.load(open("$out/bench$i.json"))
print("value %.1f q/s  launch %.1f us  per-query %.2f us  cfg4 %.1f us/q-layer  retrieval %.3f ms" % (j["value"], j["roofline"]["us_per_launch"], j["stage"]["sdag_per_query_call_us"], j["stage"]["cfg4_us_per_query_layer"], j["stage"]["retrieval_ms_per_batch"]))
PY
done
