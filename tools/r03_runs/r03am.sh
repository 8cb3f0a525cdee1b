set -e
mkdir -p gpurun_out/r03am
echo "GPU_MAX_HW_QUEUES on the box: ${GPU_MAX_HW_QUEUES:-unset}" > gpurun_out/r03am/env.txt
out=gpurun_out/r03am/bench.txt; : > $out
for cfg in sphere grid "grid --fast" instanced instanced_ext cornell; do
  timeout -k 10 240 python bench.py --config $cfg --no-extras --no-cpu-baseline --steps 3 --warmup 1 > /tmp/o.json 2> /tmp/o.err
  python3 -c "import json;d=json.load(open('/tmp/o.json'));print('$cfg', ':', d['value'], d['config']['workload'][60:140], d['hip_hardware_queues'])" >> $out
done
timeout -k 10 900 python -u -m pytest tests/test_gpu_full_size.py tests/test_js_host.py tests/test_cli_host.py -x -q -m gpu --timeout 400 --timeout-method thread > gpurun_out/r03am/tests.txt 2>&1
