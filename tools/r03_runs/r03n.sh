set -e
mkdir -p gpurun_out/r03n
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "reduce or multi_gpu or pools_on" --timeout 300 --timeout-method thread > gpurun_out/r03n/tests.txt 2>&1
PT_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --config cornell --steps 2 --warmup 1 --no-extras > gpurun_out/r03n/gloo2_weak.json 2> gpurun_out/r03n/gloo2_weak.err
PT_BENCH_BACKEND=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 2 --config cornell --strong --steps 2 --warmup 1 --no-extras > gpurun_out/r03n/gloo2_strong.json 2> gpurun_out/r03n/gloo2_strong.err
