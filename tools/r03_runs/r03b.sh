set -e
mkdir -p gpurun_out/r03b
cd /root/repo
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -x -q -k "fast_mode" --timeout 120 --timeout-method thread > gpurun_out/r03b/tests_fast.txt 2>&1
B="python bench.py --config grid --fast --no-extras --no-cpu-baseline --steps 2 --warmup 1"
timeout -k 10 240 $B > gpurun_out/r03b/grid_wide_p64.json 2> gpurun_out/r03b/err1
timeout -k 10 240 $B --check-period 0 > gpurun_out/r03b/grid_wide_p0.json 2> gpurun_out/r03b/err2
timeout -k 10 240 $B --check-period 1 > gpurun_out/r03b/grid_wide_p1.json 2> gpurun_out/r03b/err3
PT_EXTEND_VARIANT=2 timeout -k 10 240 $B --check-period 0 > gpurun_out/r03b/grid_wide_perlane.json 2> gpurun_out/r03b/err4
for w in 3,9,5,3 3,9,5,6 3,9,5,12 6,9,5,12 3,9,5,24; do PT_SCHED_W=$w timeout -k 10 240 $B --check-period 0 > gpurun_out/r03b/grid_wide_w$w.json 2> gpurun_out/r03b/err5; done
timeout -k 10 240 python bench.py --config instanced --fast --no-extras --no-cpu-baseline --steps 2 --warmup 1 --check-period 1 > gpurun_out/r03b/inst_wide_p1.json 2> gpurun_out/r03b/err6
timeout -k 10 240 python bench.py --config instanced --fast --no-extras --no-cpu-baseline --steps 2 --warmup 1 --check-period 0 > gpurun_out/r03b/inst_wide_p0.json 2> gpurun_out/r03b/err7
