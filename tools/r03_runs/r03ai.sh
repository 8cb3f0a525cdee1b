set -e
mkdir -p gpurun_out/r03ai
timeout -k 10 1100 python -u -m pytest tests -x -q -m gpu --timeout 400 --timeout-method thread > gpurun_out/r03ai/tests_gpu.txt 2>&1
out=gpurun_out/r03ai/bench.txt; : > $out
for cfg in sphere cornell instanced grid "grid --fast"; do
  timeout -k 10 240 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg', ':', d['value'])" >> $out
done
