#!/bin/bash
# Same-box A/B with interleaved repetitions: the batched cfg-3 attention launch, best and median of $REPS runs per library.
REPS=${REPS:-5}
for lib in "$@"; do : > /tmp/ab_$(basename $lib).txt; done
for rep in $(seq $REPS); do for lib in "$@"; do
  SDAG_HIP_LIB=$PWD/$lib timeout -k 10 120 python3 tools/attn_microbench.py vbatch quick | sed -n 's/.*: \([0-9.]*\) us\/launch.*/\1/p' >> /tmp/ab_$(basename $lib).txt || exit 1
done; done
for lib in "$@"; do python3 -c "
import sys; v=sorted(float(x) for x in open('/tmp/ab_$(basename $lib).txt')); print('$lib', 'min %.1f median %.1f max %.1f us/launch' % (v[0], v[len(v)//2], v[-1]), v)"; done
