#!/bin/bash
# Short A/B run: GPU tests of the attention path, event-timed single launches, bench without the CPU baseline.
set -o pipefail
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "attention or mask or generate" 2>&1 | tail -2 || exit 1
timeout -k 10 120 python3 tools/attn_microbench.py one || exit 1
timeout -k 10 120 python3 tools/attn_microbench.py cfg4 || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.1f q/s  sdag %.3f ms/query  retr %.3f ms/batch  pq %.1f us  frac %.3f'%(d['value'],d['stage']['sdag_ms_per_query'],d['stage']['retrieval_ms_per_batch'],d['stage']['sdag_per_query_call_us'] or 0,d['roofline']['frac']))" || exit 1
