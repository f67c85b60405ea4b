#!/bin/bash
# Kernel times of the 256-query search under timing-only builds of the scan (tools/_exp_<bits>.so, SDAG_SCAN_EXP bits:
# 1 no filtering, 2 no MFMAs, 4 no query pieces, 8 no corpus pieces) next to the in-tree build.
# usage: tools/exp_scan.sh <tag> <lib>...   ("tree" = the in-tree build)
set -o pipefail
tag=$1; shift
out=$PWD/gpurun_out/$tag; mkdir -p $out
root=$PWD
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  if [ $lib = tree ]; then unset SDAG_HIP_LIB; else export SDAG_HIP_LIB=$root/$lib; fi
  name=$(basename $lib .so)
  rm -rf $out/$name
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -- python3 $root/tools/dense_microbench.py 2626916 1024 3 10 only256 > $out/$name.txt 2> $out/$name.err || { tail -5 $out/$name.err; exit 1; }
  echo "== $name"; python3 $root/tools/summarize_rocprof.py $(ls $out/$name/*/*kernel_stats.csv | head -1) 12 | grep -E "dense_scan|rescore" 
done
