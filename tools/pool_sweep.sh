#!/bin/bash
# Sweep of pool count x pool size x hardware queues per process on one bench config.  Usage: tools/pool_sweep.sh <config> <out-file>
# GPU_MAX_HW_QUEUES is HIP's own knob (default 4): every pool drives its own stream, and streams beyond the hardware queues share one.
cfg=$1; out=$2; : > $out
for hwq in 4 8; do
  for shape in "1 2000000" "2 1250000" "3 750000" "4 560000" "4 750000" "6 375000" "6 500000" "8 280000"; do
    set -- $shape
    GPU_MAX_HW_QUEUES=$hwq timeout -k 10 200 python3 bench.py --config $cfg --steps 3 --warmup 1 --no-extras --no-cpu-baseline --pools $1 --paths $2 > /tmp/sweep.json 2>/tmp/sweep.err || { echo "failed: $hwq $shape" >> $out; tail -3 /tmp/sweep.err >> $out; exit 1; }
    python3 -c "import json;d=json.load(open('/tmp/sweep.json'));print('$cfg hwq $hwq pools $1 x $2 :', d['value'], 'Mrays/s', d['ms_per_step'], 'ms')" >> $out
  done
done
