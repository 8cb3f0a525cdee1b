set -e
mkdir -p gpurun_out/r03r
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu -k "tolerance" --timeout 400 --timeout-method thread > gpurun_out/r03r/tests.txt 2>&1
out=gpurun_out/r03r/bench.txt; : > $out
for cfg in "grid --fast --check-period 0" "grid --fast" "instanced --fast --check-period 0"; do
  timeout -k 10 240 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg', ':', d['value'])" >> $out
done
