#!/bin/bash
# Full-render A/B of extend variants (PT_EXTEND_VARIANT) on bench configs.  Usage: VARIANTS="4 5" tools/variant_sweep.sh <out> <config>...
out=$1; shift; : > $out
for cfg in "$@"; do
  for v in ${VARIANTS:-3 4}; do
    PT_EXTEND_VARIANT=$v timeout -k 10 300 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /tmp/vs.json 2>/tmp/vs.err || { echo "failed: $cfg $v" >> $out; tail -3 /tmp/vs.err >> $out; exit 1; }
    python3 -c "import json;d=json.load(open('/tmp/vs.json'));print('$cfg variant $v :', d['value'], 'Mrays/s', d['ms_per_step'], 'ms per render')" >> $out
  done
done
