set -e
mkdir -p gpurun_out/r03k
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -v -m gpu -k "tolerance" --timeout 400 --timeout-method thread > gpurun_out/r03k/tests2.txt 2>&1
timeout -k 10 400 python bench.py --config grid --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03k/bench_grid.json 2> gpurun_out/r03k/bench_grid.err
timeout -k 10 400 python bench.py --config instanced_ext --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r03k/bench_instanced_ext.json 2> gpurun_out/r03k/bench_instanced_ext.err
