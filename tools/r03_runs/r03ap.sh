set -e
mkdir -p gpurun_out/r03ap
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q -m gpu -k "axis_aligned or visit_counts or n1 or closest_hit or 10m or tolerance or floor" --timeout 300 --timeout-method thread > gpurun_out/r03ap/tests.txt 2>&1
out=gpurun_out/r03ap/bench.txt; : > $out
for cfg in grid "grid --fast" "grid --fast --check-period 0"; do
  timeout -k 10 240 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg', ':', d['value'])" >> $out
done
