#!/bin/bash
# Same-box A/B on bench.py itself (attention time per query, averaged over steps x 32 launches), interleaved.
REPS=${REPS:-3}
for rep in $(seq $REPS); do for lib in "$@"; do
  SDAG_HIP_LIB=$PWD/$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probe --steps 4 --warmup 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'q/s %.1f  attention us/launch %.1f' % (d['value'], d['roofline']['us_per_launch']))" || exit 1
done; done
