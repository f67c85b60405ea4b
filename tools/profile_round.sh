#!/bin/bash
# Everything the profiles/ summaries of one round come from, in one GPU call: the plain bench line, rocprofv3 kernel
# traces of the default workload (cfg3) and of cfg4, the two HBM-traffic counter passes of the default workload
# (each --pmc pass is a run of its own), and SQ counters of the attention kernel.
# usage: tools/profile_round.sh <tag>   (writes gpurun_out/<tag>_*)
set -o pipefail
tag=${1:-rXX}
root=$PWD
out=$root/gpurun_out
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $root/bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err || { echo "bench failed"; tail -5 $out/${tag}_bench.err; exit 1; }
for cfg in cfg3 cfg4; do
  rm -rf $out/${tag}_trace_$cfg
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace_$cfg -- python3 $root/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-probe > $out/${tag}_trace_${cfg}_bench.json 2> $out/${tag}_trace_$cfg.err || { echo "trace $cfg failed"; tail -5 $out/${tag}_trace_$cfg.err; exit 1; }
  python3 $root/tools/summarize_rocprof.py $(ls $out/${tag}_trace_$cfg/*/*kernel_stats.csv | head -1) > $out/${tag}_${cfg}_kernel_stats.txt || exit 1
done
# probes (per-query calls, decode steps) under the kernel trace
rm -rf $out/${tag}_trace_probe
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace_probe -- python3 $root/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $out/${tag}_trace_probe_bench.json 2> $out/${tag}_trace_probe.err || { echo "trace probe failed"; tail -5 $out/${tag}_trace_probe.err; exit 1; }
python3 $root/tools/summarize_rocprof.py $(ls $out/${tag}_trace_probe/*/*kernel_stats.csv | head -1) 16 > $out/${tag}_probe_kernel_stats.txt || exit 1
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/${tag}_pmc_$ctr
  timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d $out/${tag}_pmc_$ctr -- python3 $root/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-probe > /dev/null 2> $out/${tag}_pmc_$ctr.err || { echo "pmc $ctr failed"; tail -5 $out/${tag}_pmc_$ctr.err; exit 1; }
done
python3 $root/tools/collect_traffic.py $(ls $out/${tag}_pmc_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $out/${tag}_pmc_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/${tag}_traffic.json cfg3 "$tag" > /dev/null || exit 1
# the same two passes for the long-context workload (its roofline.traffic)
for ctr in FETCH_SIZE WRITE_SIZE; do
  rm -rf $out/${tag}_pmc4_$ctr
  timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d $out/${tag}_pmc4_$ctr -- python3 $root/bench.py --config cfg4 --steps 1 --warmup 0 --no-cpu-baseline --no-probe > /dev/null 2> $out/${tag}_pmc4_$ctr.err || { echo "pmc cfg4 $ctr failed"; tail -5 $out/${tag}_pmc4_$ctr.err; exit 1; }
done
python3 $root/tools/collect_traffic.py $(ls $out/${tag}_pmc4_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls $out/${tag}_pmc4_WRITE_SIZE/*/*counter_collection.csv | head -1) $out/${tag}_traffic.json cfg4 "$tag" > /dev/null || exit 1
# the search with 32 .. 1024 queries per call under the kernel trace
rm -rf $out/${tag}_trace_dense
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/${tag}_trace_dense -- python3 $root/tools/dense_microbench.py 2626916 1024 3 10 wide > $out/${tag}_dense_wide.txt 2> $out/${tag}_trace_dense.err || { echo "trace dense failed"; tail -5 $out/${tag}_trace_dense.err; exit 1; }
python3 $root/tools/summarize_rocprof.py $(ls $out/${tag}_trace_dense/*/*kernel_stats.csv | head -1) 12 > $out/${tag}_dense_wide_kernel_stats.txt || exit 1
# decode attention over batch sizes
timeout -k 10 200 python3 $root/tools/decode_sweep.py 3100 > $out/${tag}_decode_sweep.txt 2>&1 || { echo "decode sweep failed"; exit 1; }
# the sharded-corpus workload with the whole 21,015,324-row corpus on this one GPU
timeout -k 10 600 python3 $root/bench.py --config cfg5 --steps 3 --warmup 1 --no-cpu-baseline --no-probe > $out/${tag}_bench_cfg5_21m.json 2> $out/${tag}_bench_cfg5.err || { echo "cfg5 failed"; tail -5 $out/${tag}_bench_cfg5.err; exit 1; }
for cfg in cfg3 cfg4; do
  rm -rf $out/${tag}_sq_$cfg
  timeout -s KILL 240 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $out/${tag}_sq_$cfg -- python3 $root/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-probe > /dev/null 2> $out/${tag}_sq_$cfg.err || { echo "sq $cfg failed"; tail -5 $out/${tag}_sq_$cfg.err; exit 1; }
  rm -rf $out/${tag}_sq2_$cfg
  timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $out/${tag}_sq2_$cfg -- python3 $root/bench.py --config $cfg --steps 1 --warmup 0 --no-cpu-baseline --no-probe > /dev/null 2> $out/${tag}_sq2_$cfg.err || { echo "sq2 $cfg failed"; tail -5 $out/${tag}_sq2_$cfg.err; exit 1; }
  python3 $root/tools/pmc_summary.py $out/${tag}_sq_$cfg $out/${tag}_sq2_$cfg > $out/${tag}_${cfg}_sq_counters.txt || exit 1
done
cat $out/${tag}_bench.json
cat $out/${tag}_cfg3_kernel_stats.txt
