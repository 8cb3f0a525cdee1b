set -e
mkdir -p gpurun_out/r03l
timeout -k 10 1100 python -u -m pytest tests -x -q -m gpu --timeout 400 --timeout-method thread > gpurun_out/r03l/tests_gpu.txt 2>&1
for cfg in sphere cornell; do timeout -k 10 300 python bench.py --config $cfg --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r03l/bench_$cfg.json 2> gpurun_out/r03l/err; done
timeout -k 10 300 python bench.py --config grid --fast --check-period 0 --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r03l/bench_grid_fast.json 2> gpurun_out/r03l/err
timeout -k 10 300 python bench.py --config grid --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r03l/bench_grid.json 2> gpurun_out/r03l/err
timeout -k 10 300 python bench.py --config instanced --steps 3 --warmup 1 --no-extras --no-cpu-baseline > gpurun_out/r03l/bench_instanced.json 2> gpurun_out/r03l/err
