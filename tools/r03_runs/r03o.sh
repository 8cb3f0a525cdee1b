set -e
mkdir -p gpurun_out/r03o
timeout -k 10 900 python -u -m pytest tests/test_js_host.py tests/test_cli_host.py -x -q -m gpu --timeout 300 --timeout-method thread > gpurun_out/r03o/tests.txt 2>&1
