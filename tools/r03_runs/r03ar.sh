set -e
mkdir -p gpurun_out/r03ar
out=gpurun_out/r03ar/ab.txt; : > $out
for rep in 1 2 3; do for shape in 5 8; do
  PT_COMPACT_SHAPE=$shape timeout -k 10 240 python bench.py --config sphere --no-extras --no-cpu-baseline --steps 4 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('sphere', 'PT_COMPACT_SHAPE=$shape', ':', d['value'])" >> $out
done; done
for shape in 5 8; do
  PT_COMPACT_SHAPE=$shape timeout -k 10 240 python bench.py --config grid --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('grid', 'PT_COMPACT_SHAPE=$shape', ':', d['value'])" >> $out
done
