set -e
mkdir -p gpurun_out/r03f
timeout -k 10 900 python -u -m pytest tests -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r03f/tests_gpu.txt 2>&1
bash tools/pmc_passes.sh r3_grid_wide --config grid --fast --check-period 0 --pools 1 --paths 2000000 --steps 1 --warmup 1
