set -e
mkdir -p gpurun_out/r03e
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "fast_mode" --timeout 120 --timeout-method thread > gpurun_out/r03e/tests_fast.txt 2>&1
out=gpurun_out/r03e/check_cost.txt; : > $out
for cfg in grid instanced; do
for p in 0 64 16 1; do
  timeout -k 10 240 python bench.py --config $cfg --fast --no-extras --no-cpu-baseline --steps 2 --warmup 1 --check-period $p > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg', 'period $p', ':', d['value'], d['tolerance_check'])" >> $out
done; done
