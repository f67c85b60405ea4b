#!/bin/bash
# Everything the profiles/ summaries of one round come from: plain bench, rocprofv3 kernel trace of both launch
# modes, and the two HBM-traffic counter passes.  usage: tools/profile_round.sh <tag>   (writes gpurun_out/<tag>_*)
set -o pipefail
tag=${1:-rXX}
out=gpurun_out
mkdir -p $out
timeout -k 10 400 python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { echo "bench failed"; tail -5 $out/${tag}_bench.err; exit 1; }
for mode in batched per_query; do
  rm -rf $out/${tag}_trace_$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace_$mode -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-probe --mode $mode > $out/${tag}_trace_${mode}_bench.json 2> $out/${tag}_trace_$mode.err || { echo "trace $mode failed"; tail -5 $out/${tag}_trace_$mode.err; exit 1; }
  python3 tools/summarize_rocprof.py $(ls $out/${tag}_trace_$mode/*/*kernel_stats.csv | head -1) > $out/${tag}_${mode}_kernel_stats.txt || exit 1
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $out/${tag}_pmc_${mode}_$ctr
    timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d $out/${tag}_pmc_${mode}_$ctr -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-probe --mode $mode > /dev/null 2> $out/${tag}_pmc_${mode}_$ctr.err || { echo "pmc $mode $ctr failed"; tail -5 $out/${tag}_pmc_${mode}_$ctr.err; exit 1; }
  done
  python3 tools/collect_traffic.py $(ls $out/${tag}_pmc_${mode}_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $out/${tag}_pmc_${mode}_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/${tag}_traffic.json $mode > /dev/null || exit 1
done
cat $out/${tag}_bench.json
cat $out/${tag}_batched_kernel_stats.txt
