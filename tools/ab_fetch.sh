#!/bin/bash
# Same-box A/B of the batched cfg-3 attention launch: event-timed, then one FETCH_SIZE counter pass per library.
set -o pipefail
for rep in 1 2; do for lib in "$@"; do
  export SDAG_HIP_LIB=$PWD/$lib; echo "== $lib"
  timeout -k 10 120 python3 tools/attn_microbench.py vbatch quick || exit 1
done; done
for lib in "$@"; do
  export SDAG_HIP_LIB=$PWD/$lib
  out=gpurun_out/fetch_$(basename $lib .so); rm -rf $out
  timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out -- python3 tools/attn_microbench.py vbatch quick > $out.log 2>&1 || { echo "pass failed"; tail -5 $out.log; exit 1; }
  python3 - $out $lib <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
tot = n = 0
for r in csv.DictReader(open(f)):
    if "sdag_attn_main" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        tot += float(r["Counter_Value"]); n += 1
print(f"{sys.argv[2]}: reads per launch {tot / n * 1024 * 2 / 1e9:.3f} GB (FETCH_SIZE x 2)")
PY
done
