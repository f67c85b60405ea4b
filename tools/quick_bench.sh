#!/bin/bash
# Short A/B run: GPU tests of the attention path, bench without the CPU baseline, and the read-side HBM counter.
set -o pipefail
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread -k "attention or mask or generate" 2>&1 | tail -2 || exit 1
timeout -k 10 300 python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value %.1f q/s  sdag %.3f ms/query  retr %.3f ms/batch  pq %.1f us  frac %.3f'%(d['value'],d['stage']['sdag_ms_per_query'],d['stage']['retrieval_ms_per_batch'],d['stage']['sdag_per_query_call_us'] or 0,d['roofline']['frac']))" || exit 1
rm -rf gpurun_out/qb_fetch
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/qb_fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-probe > /dev/null 2> gpurun_out/qb_fetch.err || exit 1
python3 - <<'PY'
import csv,glob,re,collections
tot=collections.defaultdict(float);n=collections.Counter()
for p in glob.glob('gpurun_out/qb_fetch/*/*counter_collection.csv'):
    for r in csv.DictReader(open(p)):
        m=re.search(r'(sdag_attn_main|dense_scan)',r['Kernel_Name'])
        if m and r['Counter_Name']=='FETCH_SIZE': tot[m.group(1)]+=float(r['Counter_Value']);n[m.group(1)]+=1
for k in tot: print(k,'read MB/launch (corrected x2): %.1f'%(2*tot[k]/n[k]*1024/1e6))
PY
