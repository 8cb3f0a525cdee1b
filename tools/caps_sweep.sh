#!/bin/bash
# Sweep of the extend stage's trip caps (PT_EXTEND_CAPS=first,second; 0 = one launch walks every ray to its end) on bench configs.
# PT_EXTEND_CAPS only exists with profiles/r02_trip_cap_experiment/trip_capped_extend.patch applied (the experiment was not adopted).
# Usage: [CAPS="0 16 16,32"] tools/caps_sweep.sh <out-file> <config>... 
out=$1; shift; : > $out
for cfg in "$@"; do
  for caps in ${CAPS:-0 8 16 32 "8,16" "8,24" "12,24" "16,32" "16,48" "24,48" "32,64"}; do
    PT_EXTEND_CAPS=$caps timeout -k 10 300 python3 bench.py --config $cfg --steps 2 --warmup 1 --no-extras --no-cpu-baseline > /tmp/caps.json 2>/tmp/caps.err || { echo "failed: $cfg $caps" >> $out; tail -3 /tmp/caps.err >> $out; exit 1; }
    python3 -c "import json;d=json.load(open('/tmp/caps.json'));print('$cfg caps $caps :', d['value'], 'Mrays/s', d['ms_per_step'], 'ms per render')" >> $out
  done
done
