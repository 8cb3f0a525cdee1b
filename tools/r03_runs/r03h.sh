set -e
mkdir -p gpurun_out/r03h
out=gpurun_out/r03h/weights.txt; : > $out
for w in 3,2,5,3 3,3,5,3 3,4,5,3 3,6,5,3 3,9,5,3 3,14,5,3 3,6,5,2 3,9,5,2; do
  PT_SCHED_W=$w timeout -k 10 240 python bench.py --config grid --fast --no-extras --no-cpu-baseline --steps 2 --warmup 1 --check-period 0 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('grid wide weights inner,triangle,switch,wide = $w :', d['value'])" >> $out
done
for shape in "2 1250000" "3 1000000" "4 600000" "2 1000000"; do set -- $shape
  timeout -k 10 240 python bench.py --config grid --fast --no-extras --no-cpu-baseline --steps 2 --warmup 1 --check-period 0 --pools $1 --paths $2 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('grid wide pools $1 x $2 :', d['value'])" >> $out
done
