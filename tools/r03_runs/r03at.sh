set -e
mkdir -p gpurun_out/r03at
out=gpurun_out/r03at/modes.txt; : > $out
for cfg in sphere cornell instanced floor; do for mode in "" "--fast --check-period 0" "--fast"; do
  timeout -k 10 240 python bench.py --config $cfg $mode --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg', '[$mode]', ':', d['value'])" >> $out
done; done
