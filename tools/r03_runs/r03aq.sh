set -e
mkdir -p gpurun_out/r03aq
out=gpurun_out/r03aq/compact_shape.txt; : > $out
for cfg in sphere "grid --fast" instanced; do for shape in 5 8 2 7; do
  PT_COMPACT_SHAPE=$shape timeout -k 10 240 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg', 'PT_COMPACT_SHAPE=$shape', ':', d['value'])" >> $out
done; done
