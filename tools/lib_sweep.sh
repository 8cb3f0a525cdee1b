#!/bin/bash
# Full-render A/B of several builds of the library (PT_AMD_LIB) on bench configs.
# Usage: tools/lib_sweep.sh <out> "<config> ..." <lib.so>...
out=$1; cfgs=$2; shift 2; : > $out
for cfg in $cfgs; do
  for lib in "$@"; do
    PT_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /tmp/ls.json 2>/tmp/ls.err || { echo "failed: $cfg $lib" >> $out; tail -3 /tmp/ls.err >> $out; exit 1; }
    python3 -c "import json;d=json.load(open('/tmp/ls.json'));print('$cfg', '$(basename $lib)', ':', d['value'], 'Mrays/s', d['ms_per_step'], 'ms per render')" >> $out
  done
done
