set -e
mkdir -p gpurun_out/r03c
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "fast_mode" --timeout 120 --timeout-method thread > gpurun_out/r03c/tests_fast.txt 2>&1
B="python bench.py --config grid --fast --no-extras --no-cpu-baseline --steps 2 --warmup 1"
for w in 3,9,5,1 3,9,5,2 3,9,5,3 3,9,5,4; do PT_SCHED_W=$w timeout -k 10 240 $B --check-period 0 > gpurun_out/r03c/grid_wide_p0_w$w.json 2> gpurun_out/r03c/err; done
PT_SCHED_W=3,9,5,3 timeout -k 10 240 $B > gpurun_out/r03c/grid_wide_p64.json 2> gpurun_out/r03c/err
PT_SCHED_W=3,9,5,3 timeout -k 10 240 $B --check-period 16 > gpurun_out/r03c/grid_wide_p16.json 2> gpurun_out/r03c/err
PT_SCHED_W=3,9,5,3 timeout -k 10 240 $B --check-period 1 > gpurun_out/r03c/grid_wide_p1.json 2> gpurun_out/r03c/err
timeout -k 10 240 python bench.py --config instanced --fast --no-extras --no-cpu-baseline --steps 2 --warmup 1 --check-period 1 > gpurun_out/r03c/inst_wide_p1.json 2> gpurun_out/r03c/err
for w in 6,9,5,3 6,9,5,6 6,9,5,2; do PT_SCHED_W=$w timeout -k 10 240 python bench.py --config instanced --fast --no-extras --no-cpu-baseline --steps 2 --warmup 1 --check-period 0 > gpurun_out/r03c/inst_wide_p0_w$w.json 2> gpurun_out/r03c/err; done
