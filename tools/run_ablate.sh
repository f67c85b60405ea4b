#!/bin/bash
# Times the ablated diagnostic builds of the attention kernel (tools/_abl_N.so) against the shipped library.
for a in base 1 2 4 8 6 7; do
  if [ $a = base ]; then unset SDAG_HIP_LIB; else export SDAG_HIP_LIB=$PWD/tools/_abl_$a.so; fi
  echo "== ablate $a"
  timeout -k 10 60 python3 tools/attn_microbench.py vbatch quick 2>&1 | grep -v amdgpu.ids || exit 1
  timeout -k 10 60 python3 tools/attn_microbench.py one 2>&1 | grep -v amdgpu.ids || exit 1
done
