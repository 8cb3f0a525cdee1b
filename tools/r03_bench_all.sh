# The round's bench lines: the headline (default) and the other BASELINE configs, each with its extras.
set -e
mkdir -p gpurun_out/r03_bench
timeout -k 10 600 python bench.py > gpurun_out/r03_bench/sphere.json 2> gpurun_out/r03_bench/sphere.err
timeout -k 10 600 python bench.py --config grid > gpurun_out/r03_bench/grid.json 2> gpurun_out/r03_bench/grid.err
timeout -k 10 600 python bench.py --config grid --fast > gpurun_out/r03_bench/grid_tolerance.json 2> gpurun_out/r03_bench/grid_tolerance.err
timeout -k 10 600 python bench.py --config instanced_ext > gpurun_out/r03_bench/instanced_ext.json 2> gpurun_out/r03_bench/instanced_ext.err
timeout -k 10 600 python bench.py --config instanced > gpurun_out/r03_bench/instanced.json 2> gpurun_out/r03_bench/instanced.err
timeout -k 10 600 python bench.py --config cornell > gpurun_out/r03_bench/cornell.json 2> gpurun_out/r03_bench/cornell.err
timeout -k 10 600 python bench.py --config floor --no-cpu-baseline > gpurun_out/r03_bench/floor.json 2> gpurun_out/r03_bench/floor.err
