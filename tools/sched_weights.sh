#!/bin/bash
# Sweep of the kind-scheduled walk's body weights (PT_SCHED_W=inner,triangle,switch) on bench configs, full render.
# Usage: [WEIGHTS="6,9,5 4,9,5"] tools/sched_weights.sh <out> <config>...
out=$1; shift; : > $out
for cfg in "$@"; do
  for w in ${WEIGHTS:-6,9,5 4,9,5 3,9,5 6,12,5 6,9,7 6,9,3 8,12,5 5,9,5}; do
    PT_SCHED_W=$w timeout -k 10 300 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /tmp/sw.json 2>/tmp/sw.err || { echo "failed: $cfg $w" >> $out; tail -3 /tmp/sw.err >> $out; exit 1; }
    python3 -c "import json;d=json.load(open('/tmp/sw.json'));print('$cfg weights $w :', d['value'], 'Mrays/s', d['ms_per_step'], 'ms per render')" >> $out
  done
done
