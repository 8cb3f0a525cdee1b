set -e
mkdir -p gpurun_out/r03af
out=gpurun_out/r03af/weights.txt; : > $out
for cfg in sphere instanced; do
for w in 6,9,5 5,9,5 7,9,5 8,9,5 6,8,5 6,10,5 6,9,4 6,9,6; do
  PT_SCHED_W=$w timeout -k 10 240 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg weights inner,triangle,switch = $w :', d['value'])" >> $out
done; done
